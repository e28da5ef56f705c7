mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_prims.py tests/test_gpu_construct.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3a/pytest.log
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['sa_build'],1))" >> gpurun_out/r3a/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), round(d['breakdown_ms_per_step']['sa_build'],2))" >> gpurun_out/r3a/ab.txt
  done
  env "$@" python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['sa_build'],1))" >> gpurun_out/r3a/ab.txt
}
run rs512 FOO=1
run rs256 RV_LIB_DIR=$PWD/gpurun_ab/rs256
run rs512 FOO=1
