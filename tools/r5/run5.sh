mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py -q -x 2>&1 | tail -15
python bench.py --classes snp0.1,snp1,repeats,identical --L 50000000 --class-timeout 300 > gpurun_out/classes_50M_c.json 2> gpurun_out/classes_50M_c.err
python bench.py --L 50000000 --snp 0.001 --steps 3 > gpurun_out/b50_snp01.json 2>/dev/null
python bench.py --L 50000000 --repeats 0.02 --nruns 5 --steps 3 > gpurun_out/b50_rep.json 2>/dev/null
python bench.py --L 50000000 --steps 3 --no-cpu --no-extra > gpurun_out/b50_snp1.json 2>/dev/null
