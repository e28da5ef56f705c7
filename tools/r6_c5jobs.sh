#!/bin/bash
# round 6: config 5 level 0 -- how many alignments in flight serve the GPU best?
O=gpurun_out/r6c5jobs; mkdir -p $O
for j in 2 3 4 5 6 10 12; do
  python bench.py --config c5 --jobs $j --steps 3 --warmup 1 --no-cpu --no-check > $O/c5_j$j.json 2> $O/c5_j$j.err
done
python - <<'P'
import json
for j in (2, 3, 4, 5, 6, 10, 12):
    try:
        d = json.loads(open("gpurun_out/r6c5jobs/c5_j%d.json" % j).read().strip().splitlines()[-1])
        print(j, round(d["ms_per_step"], 2), round(d["value"]))
    except Exception as e:
        print(j, "failed", e)
P
