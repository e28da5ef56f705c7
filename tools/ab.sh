# Same-run A/B of two builds (or of environment switches) on the GPU box -- box-to-box and warm-up spread is ~3 %, so two variants
# are only ever compared inside one gpurun call, interleaved, a few repetitions each.
#   build the variant into its own directory (the library names stay):   gpurun_ab/<name>/libreveal_amd.so, libreveal_amd64.so
#   usage (on the box): bash tools/ab.sh OUTDIR "labelA ENV=..." "labelB RV_LIB_DIR=$PWD/gpurun_ab/<name>" ...
OUT=gpurun_out/$1; shift; mkdir -p $OUT
run() { lab=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C4 $lab', round(d['ms_per_step'],1), {k: round(v,1) for k,v in b.items()})" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --steps 10 --warmup 3 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; print('C2 $lab', round(d['ms_per_step'],2), {k: round(v,2) for k,v in b.items()})" >> $OUT/ab.txt
  env "$@" timeout 300 python bench.py --L 5000000 --genomes 10 --steps 3 --warmup 2 --no-cpu --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 $lab', round(d['ms_per_step'],1))" >> $OUT/ab.txt
}
for rep in 1 2 3; do
  for v in "$@"; do run $v; done
done
cat $OUT/ab.txt
