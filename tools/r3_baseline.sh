# round-3 baseline: bench C4 + PMC traffic of every kernel (run on the GPU box)
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
python bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_c4.json 2> $OUT/bench_c4.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-check > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-check > $OUT/pmc_write.log 2>&1
cd $R
python tools/pmc_all.py $OUT/pmc_fetch $OUT/pmc_write 40 > $OUT/pmc_traffic_c4.txt 2> $OUT/pmc_all.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pmc_traffic_c4.txt; tail -c 600 $OUT/bench_c4.json
