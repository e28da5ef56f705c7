# rocprofv3 kernel stats of any bench invocation (run on the GPU box): bash tools/prof_any.sh OUTNAME <bench args...>
OUT=$PWD/gpurun_out/$1; shift; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o x -- python $R/bench.py --no-cpu --no-check "$@" > $OUT/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $OUT/prof/*/x_results.db $OUT/prof/x_results.db 2>/dev/null | head -1) > $OUT/kernel_stats.txt
rm -rf $OUT/prof
head -30 $OUT/kernel_stats.txt
