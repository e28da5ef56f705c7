mkdir -p gpurun_out/r2z
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['bubble'],1))" >> gpurun_out/r2z/ab.txt
  done
  env "$@" python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['bubble'],1))" >> gpurun_out/r2z/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), round(d['breakdown_ms_per_step']['bubble'],2))" >> gpurun_out/r2z/ab.txt
}
run default FOO=1
run par128k RV_BUBBLE_PAR_MIN=131072
run par256k RV_BUBBLE_PAR_MIN=262144
run par384k RV_BUBBLE_PAR_MIN=393216
run par512k RV_BUBBLE_PAR_MIN=524288
run default FOO=1
