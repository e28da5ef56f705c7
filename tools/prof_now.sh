OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-check > $OUT/prof_c4.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -- python $R/bench.py --L 5000000 --steps 5 --warmup 1 --no-cpu --no-check > $OUT/prof_c2.log 2>&1 < /dev/null
