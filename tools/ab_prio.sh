mkdir -p gpurun_out/r2u
python -m pytest tests/test_gpu_align.py tests/test_gpu_handoff.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2u/pytest.log
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'])" >> gpurun_out/r2u/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), d['breakdown_ms_per_step'])" >> gpurun_out/r2u/ab.txt
  done
}
run prio FOO=1
run noprio RV_LEAF_PRIO=0 RV_MAIN_PRIO_DEFAULT=1
run leaflow_only RV_MAIN_PRIO_DEFAULT=1
run prio FOO=1
run noprio RV_LEAF_PRIO=0 RV_MAIN_PRIO_DEFAULT=1
