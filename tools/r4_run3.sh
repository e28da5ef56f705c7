set -x
O=gpurun_out/r4_run3; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python tools/fuzz.py 40 77 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
python bench.py --indelfrac 0.2 --no-cpu --no-extra > $O/bench_c4_indel.json 2> $O/bench_c4_indel.err; tail -c 600 $O/bench_c4_indel.err
python -c "import json; d=json.load(open('$O/bench_c4_indel.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms_per_step'], d['sa_build'], d['parity'], d['properties_full_size']['all'])"
