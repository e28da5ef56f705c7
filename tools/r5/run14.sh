mkdir -p gpurun_out
python bench.py --cpu-full --no-extra --no-allcores > gpurun_out/bench_cpu_full.json 2> gpurun_out/bench_cpu_full.err
