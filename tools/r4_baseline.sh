set -x
O=gpurun_out/r4_base; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 1500 $O/bench_c4.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-check > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1) > $O/kernel_stats_c4.txt
rm -rf $O/prof
head -40 $O/kernel_stats_c4.txt
python -c "import json; d=json.load(open('$O/bench_c4.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['breakdown_ms_per_step'])"
