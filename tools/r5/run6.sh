mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py -q -x 2>&1 | tail -8
timeout 600 python tools/fuzz.py 240 5101 2>&1 | tail -4
python bench.py --classes snp0.1,snp1,repeats,identical --L 50000000 --class-timeout 300 > gpurun_out/classes_50M_d.json 2> gpurun_out/classes_50M_d.err
bash tools/r5/prof_class.sh p_snp01 snp0.1 50000000 > /dev/null
bash tools/r5/prof_class.sh p_rep repeats 50000000 > /dev/null
bash tools/r5/prof_class.sh p_ident identical 50000000 > /dev/null
