mkdir -p gpurun_out/r2s
python -m pytest tests/test_gpu_construct.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2s/pytest.log
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'])" >> gpurun_out/r2s/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), d['breakdown_ms_per_step'])" >> gpurun_out/r2s/ab.txt
  done
  env "$@" python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'])" >> gpurun_out/r2s/ab.txt
}
run packed FOO=1
run bytes RV_NO_PACKED_TEXT=1
