cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_lp; mkdir -p $O
RV_NO_CASCADE=1 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-check > $O/bench.json 2> $O/bench.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1) > $O/kernel_stats.txt
rm -rf $O/prof
head -40 $O/kernel_stats.txt
