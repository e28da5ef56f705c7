# rocprofv3 kernel stats of the default bench (run on the GPU box): bash tools/prof_c4.sh OUTNAME [bench args]
OUT=$PWD/gpurun_out/$1; shift; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o x -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-check "$@" > $OUT/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $OUT/prof/*/x_results.db $OUT/prof/x_results.db 2>/dev/null | head -1) > $OUT/kernel_stats.txt
rm -rf $OUT/prof
head -40 $OUT/kernel_stats.txt
