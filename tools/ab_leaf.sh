mkdir -p gpurun_out/r2m
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'])" >> gpurun_out/r2m/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), d['breakdown_ms_per_step'])" >> gpurun_out/r2m/ab.txt
  done
}
run leaf2k FOO=1
run leaf4k RV_LIB_DIR=$PWD/gpurun_ab/leaf4k
python -m pytest tests/test_gpu_align.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r2m/pytest_leaf2k.log
RV_LIB_DIR=$PWD/gpurun_ab/leaf4k python -m pytest tests/test_gpu_align.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r2m/pytest_leaf4k.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2m/prof_c2 -- python $R/bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check > $R/gpurun_out/r2m/prof_c2.log 2>&1
