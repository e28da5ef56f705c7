OUT=gpurun_out/$1; mkdir -p $OUT
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 $lab', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 $lab', round(d['ms_per_step'],2), {k: round(v,2) for k,v in b.items()})" >> $OUT/ab.txt
}
for rep in 1 2; do
run big16k FOO=1
run big64k RV_LIB_DIR=$PWD/gpurun_ab/big64k
run big256k RV_LIB_DIR=$PWD/gpurun_ab/big256k
run big8k RV_LIB_DIR=$PWD/gpurun_ab/big8k
done
