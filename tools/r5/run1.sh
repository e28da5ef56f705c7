set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py -q -x -k "reset" 2>&1 | tail -5
python bench.py --config stream --pairs 6 --steps 1 --warmup 1 > gpurun_out/stream6.json 2> gpurun_out/stream6.err; tail -3 gpurun_out/stream6.err; cat gpurun_out/stream6.json
python bench.py --config stream --pairs 20 --steps 1 --warmup 1 > gpurun_out/stream20.json 2> gpurun_out/stream20.err; tail -3 gpurun_out/stream20.err; cat gpurun_out/stream20.json
RV_NO_CASCADE=1 python tools/level_log.py 2 250000000 2> gpurun_out/level_log_c4.txt; tail -60 gpurun_out/level_log_c4.txt
nproc; free -g | head -2
