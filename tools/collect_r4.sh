# what profiles/ holds for round 4 (run on the GPU box): bash tools/collect_r4.sh OUTNAME
set -x
NAME=$1; OUT=$PWD/gpurun_out/$NAME; mkdir -p $OUT
R=$PWD
export TMPDIR=/tmp
python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --indelfrac 0.2 --no-cpu --no-extra > $OUT/bench_c4_indel.json 2> $OUT/bench_c4_indel.err
python bench.py --L 5000000 --steps 20 --warmup 5 --no-allcores > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 3 --no-allcores > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --config c5 --steps 2 --warmup 1 > $OUT/bench_c5_level0.json 2> $OUT/bench_c5_level0.err
python bench.py --contigs 2 --steps 3 > $OUT/bench_c4_contigs2.json 2> $OUT/bench_c4_contigs2.err
cd /tmp
for w in "c4" "c4_indel --indelfrac 0.2" "c3 --L 5000000 --genomes 10"; do
  set -- $w; name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o x -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra --no-check "$@" > $OUT/prof_$name.log 2>&1
  python $R/tools/rocpd_stats.py $(ls $OUT/prof_$name/*/x_results.db $OUT/prof_$name/x_results.db 2>/dev/null | head -1) > $OUT/kernel_stats_$name.txt
  rm -rf $OUT/prof_$name
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --no-check > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --no-check > $OUT/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write k_scan_pair "2x250000000-32" > $OUT/pmc_scan_c4.json
python tools/pmc_all.py $OUT/pmc_fetch $OUT/pmc_write 40 > $OUT/pmc_traffic_c4.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
bash tools/pmc_wrreq.sh $NAME/wr $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --no-check > /dev/null 2>&1
python tools/ubench/radix_time.py > $OUT/radix_variants.txt 2>&1
python tools/mem_probe.py 250e6 > $OUT/mem_probe.txt 2>&1
bash tools/pmc_insts.sh $NAME/insts $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --no-check > /dev/null 2>&1
bash tools/pmc_sq.sh $NAME/sq $R/bench.py --steps 1 --warmup 1 --no-cpu --no-extra --no-check > /dev/null 2>&1
python bench.py --sa64 --L 1100000000 --steps 3 --warmup 1 --no-cpu --no-extra > $OUT/bench_sa64_2x1100M.json 2> $OUT/bench_sa64_2x1100M.err
