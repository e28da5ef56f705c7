mkdir -p gpurun_out
python -m pytest tests/test_gpu_construct.py -q -x 2>&1 | tail -4
timeout 900 python tools/fuzz.py 150 5103 2>&1 | tail -3
python bench.py --classes snp0.1,snp1,indel,identical,contigs50 --class-timeout 400 > gpurun_out/classes_c4_c.json 2> gpurun_out/classes_c4_c.err
R=$PWD; OUT=$PWD/gpurun_out/p4_rep2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o x -- python $R/bench.py --class-one repeats --L 250000000 --no-check > $OUT/prof.log 2>&1
cd $R; python tools/rocpd_seq.py $(ls $OUT/prof/*/x_results.db | head -1) k_cas_dwalk > $OUT/dwalk_seq.txt 2>&1; python tools/rocpd_seq.py $(ls $OUT/prof/*/x_results.db | head -1) k_cas_assign > $OUT/assign_seq.txt 2>&1; rm -rf $OUT/prof
