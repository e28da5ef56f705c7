# kernel statistics of construct() alone: bash tools/prof_construct.sh OUTNAME L GENOMES [REPS] [grep pattern]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p -o x -- python $R/tools/sa_probe.py $2 $3 ${4:-4} > $O/run.log 2>&1
grep "construct ms" $O/run.log | cut -c1-100
python $R/tools/rocpd_stats.py $(ls $O/p/*/x_results.db $O/p/x_results.db 2>/dev/null | head -1) > $O/kernel_stats.txt
rm -rf $O/p
grep "${5:-round_text\|heads_publish \|total GPU}" $O/kernel_stats.txt | cut -c1-120
