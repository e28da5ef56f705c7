# FETCH_SIZE / WRITE_SIZE of every kernel of a command (run on the GPU box): bash tools/pmc_kernel.sh OUTNAME <python args...>
OUT=$PWD/gpurun_out/$1; shift; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python "$@" > $OUT/pmc_write.log 2>&1
cd $R
python tools/pmc_all.py $OUT/pmc_fetch $OUT/pmc_write 30 > $OUT/pmc_traffic.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pmc_traffic.txt
