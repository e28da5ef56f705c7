OUT=gpurun_out/$1; mkdir -p $OUT
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C3 $lab', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --genomes 5 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5job $lab', round(d['ms_per_step'],1))" >> $OUT/ab.txt
}
for rep in 1 2; do
run pm256k FOO=1
run pm128k RV_BUBBLE_PAR_MIN=131072
run pm64k RV_BUBBLE_PAR_MIN=65536
run pm512k RV_BUBBLE_PAR_MIN=524288
done
