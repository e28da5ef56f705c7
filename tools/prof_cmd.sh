# rocprofv3 kernel stats of one bench command: bash tools/prof_cmd.sh OUTNAME <bench.py arguments>
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; shift
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o x -- python $R/bench.py "$@" > $OUT/prof.log 2>&1
cd $R
DB=$(ls $OUT/prof/*/x_results.db $OUT/prof/x_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt
python tools/rocpd_step.py $DB > $OUT/step.txt
rm -rf $OUT/prof
