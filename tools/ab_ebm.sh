mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_align.py tests/test_gpu_fuzz.py tests/test_gpu_handoff.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3d/pytest.log
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --steps 3 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'], d['properties_full_size']['all'])" >> gpurun_out/r3d/ab.txt
  env "$@" python bench.py --L 50000000 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2x50M $lab', round(d['ms_per_step'],2), d['breakdown_ms_per_step'], d['properties_full_size']['all'])" >> gpurun_out/r3d/ab.txt
  done
}
run many FOO=1
run nomany RV_NO_EARLY_BUBBLE_MANY=1
run many FOO=1
