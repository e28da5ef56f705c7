# final collection, part 2: other configs, profiles, counters
mkdir -p gpurun_out/final
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 2 > gpurun_out/final/bench_c3.json 2>/dev/null
python bench.py --L 5000000 --steps 20 --warmup 2 > gpurun_out/final/bench_c2.json 2>/dev/null
python bench.py --config c5 --steps 2 --warmup 1 > gpurun_out/final/bench_c5_level0.json 2>/dev/null
python bench.py --config stream --pairs 20 --steps 2 --warmup 1 > gpurun_out/final/bench_stream.json 2>/dev/null
python tools/mem_probe.py 250000000 2>&1 | tail -1 > gpurun_out/final/mem_probe.txt
python tools/mem_probe.py 1100000000 --sa64 2>&1 | tail -1 >> gpurun_out/final/mem_probe.txt
R=$PWD; OUT=$PWD/gpurun_out/final
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o x -- python $R/bench.py --no-cpu --no-extra --no-check --steps 8 --warmup 2 > $OUT/prof.log 2>&1
cd $R
DB=$(ls $OUT/prof/*/x_results.db $OUT/prof/x_results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats_c4.txt; python tools/rocpd_step.py $DB > $OUT/step_c4.txt; rm -rf $OUT/prof
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-check > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-check > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write k_scan_pair "2x250000000-32" > $OUT/pmc_scan_2x250000000-32.json; rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pmc_scan_2x250000000-32.json | head -12
