OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
python -m pytest tests/test_gpu_prims.py tests/test_gpu_construct.py -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/radix -- python $R/tools/ubench/radix_probe.py 27 > $OUT/radix.log 2>&1 < /dev/null
cd $R
for rep in 1 2; do
python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 new', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
RV_LIB_DIR=$PWD/gpurun_ab/prev python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 prev', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 new', round(d['ms_per_step'],2), {k: round(v,2) for k,v in b.items()})" >> $OUT/ab.txt
RV_LIB_DIR=$PWD/gpurun_ab/prev python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 prev', round(d['ms_per_step'],2), {k: round(v,2) for k,v in b.items()})" >> $OUT/ab.txt
done
