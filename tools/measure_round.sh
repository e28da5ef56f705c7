set -x
O=gpurun_out/final_r1c; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --genomes 10 --steps 3 --warmup 1 --no-cpu > $O/bench_c3.json 2>/dev/null
python bench.py --genomes 5 --steps 3 --warmup 1 --no-cpu > $O/bench_c5job.json 2>/dev/null
python bench.py --L 250000000 --steps 3 --warmup 1 --no-cpu > $O/bench_c4.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 5 --warmup 2 --no-cpu > $O/prof_bench.log 2>&1
python tools/rocpd_stats.py $(ls $O/prof/*/x_results.db $O/prof/x_results.db 2>/dev/null | head -1) > $O/kernel_stats_c2.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write k_scan_pair "2x5000000-32" > $O/pmc_scan.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch8 -o f -- python tools/scan_probe.py 50000000 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write8 -o w -- python tools/scan_probe.py 50000000 > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch8 $O/pmc_write8 k_scan_pair "tools/scan_probe.py 50000000 (1e8 ranks)" > $O/pmc_scan_1e8.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_fetch8 $O/pmc_write8
tail -c 600 $O/bench_c2.json; cat $O/pmc_scan.json $O/pmc_scan_1e8.json; head -20 $O/kernel_stats_c2.txt
for f in c3 c5job c4; do python -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline'])"; done
