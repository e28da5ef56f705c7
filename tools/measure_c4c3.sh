set -x
O=gpurun_out/final_r1d; mkdir -p $O
export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o x -- python bench.py --L 250000000 --steps 3 --warmup 1 --no-cpu > $O/prof_bench_c4.log 2>&1
python tools/rocpd_stats.py $(ls $O/prof_c4/*/x_results.db $O/prof_c4/x_results.db 2>/dev/null | head -1) > $O/kernel_stats_c4.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o x -- python bench.py --genomes 10 --steps 3 --warmup 1 --no-cpu > $O/prof_bench_c3.log 2>&1
python tools/rocpd_stats.py $(ls $O/prof_c3/*/x_results.db $O/prof_c3/x_results.db 2>/dev/null | head -1) > $O/kernel_stats_c3.txt
rm -rf $O/prof_c4 $O/prof_c3
tail -1 $O/prof_bench_c4.log | cut -c1-900; head -30 $O/kernel_stats_c4.txt; head -14 $O/kernel_stats_c3.txt
