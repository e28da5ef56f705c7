mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_single_steps.py tests/test_gpu_handoff.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r4d/pytest.log
RV_LEAF_PROF=1 RV_LIB_DIR=$PWD/gpurun_ab/leafprof timeout 200 python bench.py --steps 1 --warmup 0 --no-cpu --no-check 2>&1 | grep -E "^leaf" > gpurun_out/r4d/prof.txt
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 $lab', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> gpurun_out/r4d/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 $lab', round(d['ms_per_step'],2), {k: round(v,2) for k,v in b.items()})" >> gpurun_out/r4d/ab.txt
}
for rep in 1 2; do
run wave FOO=1
run pad2 RV_LIB_DIR=$PWD/gpurun_ab/pad2
run leafblk RV_LIB_DIR=$PWD/gpurun_ab/leafblk
done
