mkdir -p gpurun_out/r2r
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'])" >> gpurun_out/r2r/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), d['breakdown_ms_per_step'])" >> gpurun_out/r2r/ab.txt
  done
}
run new FOO=1
run pre RV_LIB_DIR=$PWD/gpurun_ab/pre_tmin
run new FOO=1
run pre RV_LIB_DIR=$PWD/gpurun_ab/pre_tmin
