OUT=gpurun_out/$1; mkdir -p $OUT
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 $lab', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
}
for rep in 1 2 3; do
run new FOO=1
run nostreams RV_X_NOSTREAMS=1
run lowerfirst RV_X_LOWERFIRST=1
run both RV_X_NOSTREAMS=1 RV_X_LOWERFIRST=1
run prev RV_LIB_DIR=$PWD/gpurun_ab/prev
done
