OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
for v in rs512 rs128; do
RV_LIB_DIR=$R/gpurun_ab/$v timeout 600 rocprofv3 --kernel-trace -d $OUT/$v -- python $R/tools/ubench/radix_probe.py 27 > $OUT/$v.log 2>&1 < /dev/null
done
