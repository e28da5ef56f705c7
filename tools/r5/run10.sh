mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py -q -x 2>&1 | tail -4
timeout 900 python tools/fuzz.py 150 5104 2>&1 | tail -3
python bench.py --classes snp0.1,snp1,identical --class-timeout 400 > gpurun_out/classes_c4_d.json 2> gpurun_out/classes_c4_d.err
bash tools/r5/prof_class.sh p4_snp01b snp0.1 250000000 > /dev/null
