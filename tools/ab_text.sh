mkdir -p gpurun_out/r2j
run() { # label env...
  lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['sa_build'],1))" >> gpurun_out/r2j/ab.txt
  done
  env "$@" python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['sa_build'],1))" >> gpurun_out/r2j/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), round(d['breakdown_ms_per_step']['sa_build'],2))" >> gpurun_out/r2j/ab.txt
}
run m0w4h RV_TEXT_MODE=0 RV_TEXT_W=4
run m1w4h RV_TEXT_MODE=1 RV_TEXT_W=4
run m1w2h RV_TEXT_MODE=1 RV_TEXT_W=2
run m2w4h RV_TEXT_MODE=2 RV_TEXT_W=4
run m1w4f0 RV_TEXT_MODE=1 RV_TEXT_W=4 RV_TEXT_FROM0=1
run m1w8h RV_TEXT_MODE=1 RV_TEXT_W=8
