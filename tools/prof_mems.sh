# rocprofv3 kernel stats of getmultimems on 10 x 5 Mbp: bash tools/prof_mems.sh OUTNAME
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof -o x -- python $R/tools/time_mems.py 5000000 10 20 --no-tuples > $OUT/prof.log 2>&1
cd $R
DB=$(ls $OUT/prof/*/x_results.db $OUT/prof/x_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB | grep -i "mems\|^#\|^kernel" > $OUT/kernel_stats_mems.txt
rm -rf $OUT/prof
