# kernel statistics of one bench.py invocation: bash tools/prof_bench.sh OUTNAME <bench args...>
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/bench.json 2> $O/bench.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1) > $O/kernel_stats.txt
[ -n "$KEEP_DB" ] || rm -rf $O/prof
head -36 $O/kernel_stats.txt
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms_per_step'], d['sa_build'], d['cascade'])"
