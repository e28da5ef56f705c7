mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python bench.py > gpurun_out/bench_default_a.json 2> gpurun_out/bench_default_a.err; tail -2 gpurun_out/bench_default_a.err
