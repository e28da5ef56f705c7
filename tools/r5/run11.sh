mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py -q -x 2>&1 | tail -3
python bench.py --classes snp0.1,snp1,identical --class-timeout 400 > gpurun_out/classes_c4_e.json 2> gpurun_out/classes_c4_e.err
