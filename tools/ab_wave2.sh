OUT=$PWD/gpurun_out/r4b; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/wave -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-check > $OUT/wave.log 2>&1
RV_LIB_DIR=$R/gpurun_ab/leafblk rocprofv3 --kernel-trace --stats -d $OUT/blk -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-check > $OUT/blk.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/wave2 -- python $R/bench.py --L 5000000 --steps 5 --warmup 1 --no-cpu --no-check > $OUT/wave2.log 2>&1
RV_LIB_DIR=$R/gpurun_ab/leafblk rocprofv3 --kernel-trace --stats -d $OUT/blk2 -- python $R/bench.py --L 5000000 --steps 5 --warmup 1 --no-cpu --no-check > $OUT/blk2.log 2>&1
cd $R
for d in wave blk wave2 blk2; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo == $d; head -12 $f | cut -c1-150; grep -i leaf $f | cut -c1-150; done > $OUT/summary.txt
find $OUT -name "*.db" -delete; find $OUT -name "*trace.csv" -delete
