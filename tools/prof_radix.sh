OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
timeout 300 python -m pytest tests/test_gpu_prims.py tests/test_gpu_construct.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/big -- python $R/tools/ubench/radix_probe.py 27 > $OUT/big.log 2>&1 < /dev/null
cd $R
for rep in 1 2 3; do
python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 big', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
RV_RS_SMALL_TILES=1 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 small', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 big', round(d['ms_per_step'],2), round(b['sa_build'],2))" >> $OUT/ab.txt
RV_RS_SMALL_TILES=1 python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 small', round(d['ms_per_step'],2), round(b['sa_build'],2))" >> $OUT/ab.txt
done
