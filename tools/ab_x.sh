OUT=gpurun_out/$1; mkdir -p $OUT
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 $lab', round(d['ms_per_step'],2), round(b['bubble'],2))" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C3 $lab', round(d['ms_per_step'],1), round(b['bubble'],1))" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 $lab', round(d['ms_per_step'],1), round(b['bubble'],1))" >> $OUT/ab.txt
}
for rep in 1 2; do
run cap2048 FOO=1
run cap512 RV_X_SEARCH_CAP=512
run cap128 RV_X_SEARCH_CAP=128
run cap32 RV_X_SEARCH_CAP=32
done
