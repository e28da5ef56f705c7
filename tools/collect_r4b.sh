# the part of profiles/ that this round's last changes touch (run on the GPU box): bash tools/collect_r4b.sh OUTNAME
set -x
NAME=$1; OUT=$PWD/gpurun_out/$NAME; mkdir -p $OUT
R=$PWD
export TMPDIR=/tmp
python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --L 5000000 --steps 20 --warmup 5 --no-allcores > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 3 --no-allcores > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --config c5 --steps 2 --warmup 1 > $OUT/bench_c5_level0.json 2> $OUT/bench_c5_level0.err
python bench.py --jobs 2 --steps 5 --warmup 1 --no-cpu --no-extra > $OUT/bench_c4_jobs2.json 2> $OUT/bench_c4_jobs2.err
cd /tmp
for w in "c4" "c3 --L 5000000 --genomes 10"; do
  set -- $w; name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o x -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra --no-check "$@" > $OUT/prof_$name.log 2>&1
  python $R/tools/rocpd_stats.py $(ls $OUT/prof_$name/*/x_results.db $OUT/prof_$name/x_results.db 2>/dev/null | head -1) > $OUT/kernel_stats_$name.txt
  rm -rf $OUT/prof_$name
done
cd $R
bash tools/pmc_insts.sh $NAME/insts_c3 $R/bench.py --L 5000000 --genomes 10 --steps 2 --warmup 1 --no-cpu --no-extra --no-check > /dev/null 2>&1
