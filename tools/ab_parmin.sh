mkdir -p gpurun_out/r2l
python -m pytest tests/test_gpu_align.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_handoff.py tests/test_gpu_single_steps.py tests/test_gpu_preselect.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2l/pytest.log
run() { lab=$1; shift
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['bubble'],1))" >> gpurun_out/r2l/ab.txt
  env "$@" python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 $lab', round(d['ms_per_step'],1), round(d['breakdown_ms_per_step']['bubble'],1))" >> gpurun_out/r2l/ab.txt
  env "$@" python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $lab', round(d['ms_per_step'],2), round(d['breakdown_ms_per_step']['bubble'],2))" >> gpurun_out/r2l/ab.txt
}
run default FOO=1
run refresh RV_PB_REFRESH_TMIN=1
run par16k RV_BUBBLE_PAR_MIN=16384
run par32k RV_BUBBLE_PAR_MIN=32768
run par64k RV_BUBBLE_PAR_MIN=65536
run par128k RV_BUBBLE_PAR_MIN=131072
run par256k RV_BUBBLE_PAR_MIN=262144
run par512k RV_BUBBLE_PAR_MIN=524288
