OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_single_steps.py tests/test_gpu_callbacks.py -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest.log
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 $lab', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 $lab', round(d['ms_per_step'],2), {k: round(v,2) for k,v in b.items()})" >> $OUT/ab.txt
}
for rep in 1 2 3; do
run new FOO=1
run prev RV_LIB_DIR=$PWD/gpurun_ab/prev
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_c4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-check > $R/$OUT/prof_c4.log 2>&1 < /dev/null
