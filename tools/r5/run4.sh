mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py -q -x 2>&1 | tail -15
python bench.py --classes snp0.1,snp1,repeats,repeats_indel,identical --L 50000000 --class-timeout 300 > gpurun_out/classes_50M_b.json 2> gpurun_out/classes_50M_b.err
