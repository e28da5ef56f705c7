#!/bin/bash
# round 5: the multi-sample cascade's large undecided sub-indices (k_casmb_*): tests, then 10 x 5 Mbp with indels, before / after
mkdir -p gpurun_out/big
python -m pytest tests/test_gpu_cascade.py -x -q -k "multi" > gpurun_out/big/t_multi.txt 2>&1; tail -4 gpurun_out/big/t_multi.txt
RV_CASCADE_LOG=1 python bench.py --L 5000000 --genomes 10 --indelfrac 0.2 --steps 5 --warmup 2 --no-cpu --no-extra > gpurun_out/big/c3_indel.json 2> gpurun_out/big/c3_indel.err
RV_CASM_NO_BIG=1 python bench.py --L 5000000 --genomes 10 --indelfrac 0.2 --steps 3 --warmup 1 --no-cpu --no-extra > gpurun_out/big/c3_indel_nobig.json 2> gpurun_out/big/c3_indel_nobig.err
python - <<'P'
import json
for f in ("c3_indel", "c3_indel_nobig"):
    try:
        d = json.loads(open("gpurun_out/big/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("cascade"), d.get("properties_full_size", {}).get("all"))
    except Exception as e:
        print(f, "failed", e)
P
grep "cascade (" gpurun_out/big/c3_indel.err | tail -2
timeout 240 python tools/fuzz.py 150 5301 > gpurun_out/big/fuzz.txt 2>&1; tail -3 gpurun_out/big/fuzz.txt
