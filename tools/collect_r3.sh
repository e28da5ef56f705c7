# what profiles/ holds for round 3 (run on the GPU box): bash tools/collect_r3.sh OUTNAME
set -x
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
python bench.py --steps 10 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --L 5000000 --steps 20 --warmup 5 --no-allcores > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --L 5000000 --genomes 10 --steps 5 --warmup 2 --no-allcores > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --L 5000000 --genomes 5 --steps 5 --warmup 2 --no-cpu > $OUT/bench_c5job.json 2> $OUT/bench_c5job.err
RV_NO_CASCADE=1 python bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_c4_nocascade.json 2> $OUT/bench_c4_nocascade.err
cd /tmp; export TMPDIR=/tmp
for w in "c4" "c2 --L 5000000" "c3 --L 5000000 --genomes 10"; do
  set -- $w; name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o x -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-check "$@" > $OUT/prof_$name.log 2>&1
  python $R/tools/rocpd_stats.py $(ls $OUT/prof_$name/*/x_results.db $OUT/prof_$name/x_results.db 2>/dev/null | head -1) > $OUT/kernel_stats_$name.txt
  rm -rf $OUT/prof_$name
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-check > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-check > $OUT/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write k_scan_pair "2x250000000-32" > $OUT/pmc_scan_c4.json
python tools/pmc_all.py $OUT/pmc_fetch $OUT/pmc_write 40 > $OUT/pmc_traffic_c4.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
python bench.py --sa64 --L 1100000000 --steps 3 --warmup 1 --no-cpu > $OUT/bench_sa64_2x1100M.json 2> $OUT/bench_sa64_2x1100M.err
RV_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --L 60000000 --steps 2 --warmup 1 --no-cpu > $OUT/bench_2ranks_shared_gpu.json 2> $OUT/bench_2ranks_shared_gpu.err
python tools/realistic_probe.py > $OUT/realistic_probe.txt 2> $OUT/realistic_probe.err
python tools/latency_probe.py > $OUT/latency_small.txt 2> $OUT/latency_small.err
