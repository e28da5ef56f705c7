mkdir -p gpurun_out
python -m pytest tests/test_gpu_cascade.py tests/test_gpu_graphrem.py -q -x 2>&1 | tail -4
timeout 900 python tools/fuzz.py 120 5107 2>&1 | tail -3
python bench.py --classes snp1,repeats --class-timeout 400 > gpurun_out/classes_c4_h.json 2> gpurun_out/classes_c4_h.err
bash tools/r5/prof_class.sh p4_rep6 repeats 250000000
