#!/bin/bash
O=gpurun_out/r6pre; mkdir -p $O
python tools/config5.py --genomes 100 --L 20000 --procs 8 > $O/config5_small.json 2> $O/config5_small.err; tail -c 700 $O/config5_small.json; tail -3 $O/config5_small.err
for t in 2 4; do RV_PICK_THREADS=$t python tools/time_native.py > $O/time_native_t$t.txt 2>&1; tail -1 $O/time_native_t$t.txt | cut -c1-420; done
python tools/dump_anchors.py $O/anchors_5x5M.npz > $O/dump.txt 2>&1; tail -1 $O/dump.txt
python bench.py --L 5000000 --genomes 5 --steps 10 --warmup 2 --no-cpu --no-extra > $O/c5job.json 2> $O/c5job.err
bash tools/prof_cmd.sh r6pre_prof_c5job --L 5000000 --genomes 5 --steps 10 --warmup 2 --no-cpu --no-extra
head -32 gpurun_out/r6pre_prof_c5job/kernel_stats.txt
python - <<'P'
import json
d = json.loads(open("gpurun_out/r6pre/c5job.json").read().strip().splitlines()[-1])
print("c5job", d["ms_per_step"], d["value"], d["breakdown_ms_per_step"], d["roofline_other"].get("bubble"))
P
