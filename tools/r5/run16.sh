mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py tests/test_gpu_cascade.py tests/test_gpu_graphrem.py tests/test_gpu_preselect.py -q -x 2>&1 | tail -6
timeout 900 python tools/fuzz.py 200 5106 2>&1 | tail -3
python bench.py --classes snp1,repeats,repeats_indel --class-timeout 400 > gpurun_out/classes_c4_g.json 2> gpurun_out/classes_c4_g.err
python tools/mem_probe.py 250000000 2>&1 | tail -1 > gpurun_out/mem_probe_c4.txt; cat gpurun_out/mem_probe_c4.txt
bash tools/r5/prof_class.sh p4_rep5 repeats 250000000
