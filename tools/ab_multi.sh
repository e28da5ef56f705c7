mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_align.py tests/test_gpu_fuzz.py tests/test_gpu_handoff.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3b/pytest.log
run() { lab=$1; shift
  for rep in 1 2; do
  env "$@" python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'], d['properties_full_size']['all'], d['recursion']['anchors'])" >> gpurun_out/r3b/ab.txt
  env "$@" python bench.py --L 5000000 --genomes 5 --steps 3 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5job $lab', round(d['ms_per_step'],1), d['breakdown_ms_per_step'], d['properties_full_size']['all'], d['recursion']['anchors'])" >> gpurun_out/r3b/ab.txt
  done
}
run early FOO=1
run noearlybubble RV_NO_EARLY_BUBBLE=1
run noearly RV_NO_EARLY_SPLIT=1
