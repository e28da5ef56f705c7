set -x
O=gpurun_out/r4_run1; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_prims.py tests/test_gpu_construct.py tests/test_gpu_fullsize_golden.py -x -q > $O/pytest1.log 2>&1; tail -5 $O/pytest1.log
python bench.py --L 20000000 > $O/bench_s20.json 2> $O/bench_s20.err; tail -c 800 $O/bench_s20.err
python -c "import json; d=json.load(open('$O/bench_s20.json')); print(d['value'], d['ms_per_step'], d['parity'], d.get('level_pipeline'), d.get('indel'))"
python bench.py --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 800 $O/bench_c4.err
python -c "import json; d=json.load(open('$O/bench_c4.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['breakdown_ms_per_step'], d['parity'], d.get('level_pipeline'), d.get('indel'))"
