OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES VALUBusy --output-format csv -d $OUT/valu -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-check > $OUT/valu.log 2>&1 < /dev/null
cd $R; python tools/pmc_valu.py $OUT/valu 30 > $OUT/valu.txt; rm -rf $OUT/valu
