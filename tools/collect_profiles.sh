# collects what profiles/ holds for a round: bench lines, rocprof kernel stats, PMC traffic of the judged kernel (run on the GPU box)
set -x
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
python bench.py --steps 10 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --L 5000000 --steps 20 --warmup 5 --no-allcores > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --L 5000000 --genomes 10 --steps 5 --warmup 2 --no-allcores > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --L 5000000 --genomes 5 --steps 5 --warmup 2 --no-cpu > $OUT/bench_c5job.json 2> $OUT/bench_c5job.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-check > $OUT/prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -- python $R/bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check > $OUT/prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_c3 -- python $R/bench.py --L 5000000 --genomes 10 --steps 5 --warmup 1 --no-cpu --no-check > $OUT/prof_c3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-check > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-check > $OUT/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write k_scan_pair "2x250000000-32" > $OUT/pmc_scan_c4.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
