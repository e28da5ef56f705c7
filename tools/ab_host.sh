OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_construct.py tests/test_gpu_align.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_properties.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest.log
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 $lab', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C3 $lab', round(d['ms_per_step'],1), round(b['sa_build'],1))" >> $OUT/ab.txt
}
for rep in 1 2 3; do
run new FOO=1
run prev RV_LIB_DIR=$PWD/gpurun_ab/prev
done
