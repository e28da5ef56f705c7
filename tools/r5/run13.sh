mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py tests/test_gpu_cascade.py -q -x 2>&1 | tail -3
timeout 900 python tools/fuzz.py 200 5105 2>&1 | tail -3
python bench.py --classes snp1,repeats,repeats_indel --class-timeout 400 > gpurun_out/classes_c4_f.json 2> gpurun_out/classes_c4_f.err
bash tools/r5/prof_class.sh p4_rep3 repeats 250000000 > /dev/null
