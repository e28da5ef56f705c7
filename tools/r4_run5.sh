set -x
O=gpurun_out/r4_run5; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 400 $O/bench_c4.err
python -c "import json; d=json.load(open('$O/bench_c4.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['breakdown_ms_per_step'], d['parity'], d['cpu_baseline']['value'], d['level_pipeline']['value'], d['indel']['value'], d['indel']['golden'])"
