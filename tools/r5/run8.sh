mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py tests/test_gpu_preselect.py -q -x 2>&1 | tail -4
bash tools/r5/prof_class.sh p4_rep repeats 250000000 > /dev/null
bash tools/r5/prof_class.sh p4_ident identical 250000000 > /dev/null
bash tools/r5/prof_class.sh p4_snp01 snp0.1 250000000 > /dev/null
python bench.py --classes indel,identical,contigs50 --class-timeout 400 > gpurun_out/classes_c4_b.json 2> gpurun_out/classes_c4_b.err
