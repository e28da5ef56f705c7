#!/bin/bash
# round 5: config 5 level 0 (20 jobs of 5 x 5 Mbp) with the simulator's indels, with and without the rebuild of large undecided sub-indices
mkdir -p gpurun_out/c5i
python bench.py --config c5 --indelfrac 0.2 --steps 3 --warmup 1 --no-cpu > gpurun_out/c5i/c5_indel.json 2> gpurun_out/c5i/c5_indel.err
RV_CASM_NO_BIG=1 python bench.py --config c5 --indelfrac 0.2 --steps 2 --warmup 1 --no-cpu > gpurun_out/c5i/c5_indel_nobig.json 2> gpurun_out/c5i/c5_indel_nobig.err
python bench.py --L 5000000 --genomes 10 --indelfrac 0.2 --steps 10 --warmup 2 --no-cpu --no-extra > gpurun_out/c5i/c3_indel.json 2> gpurun_out/c5i/c3_indel.err
python - <<'P'
import json
for f in ("c5_indel", "c5_indel_nobig", "c3_indel"):
    try:
        d = json.loads(open("gpurun_out/c5i/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], {k: d[k] for k in d if k in ("properties_full_size", "cascade", "paths")})
    except Exception as e:
        print(f, "failed", e)
P
tail -3 gpurun_out/c5i/c5_indel.err
