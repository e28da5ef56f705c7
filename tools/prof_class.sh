# rocprofv3 kernel stats of one input class: bash tools/prof_class.sh OUTNAME CLASS L
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o x -- python $R/bench.py --class-one $2 --L $3 --no-check > $OUT/prof.log 2>&1
cd $R
DB=$(ls $OUT/prof/*/x_results.db $OUT/prof/x_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats.txt
python tools/rocpd_step.py $DB > $OUT/step.txt
rm -rf $OUT/prof
