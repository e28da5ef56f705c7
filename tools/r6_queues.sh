#!/bin/bash
# round 6: do more hardware queues let the small kernels of concurrent jobs overlap?  (HIP maps a process' streams onto GPU_MAX_HW_QUEUES queues, 4 by default)
O=gpurun_out/r6queues; mkdir -p $O
for qn in 4 8 16 32; do
  GPU_MAX_HW_QUEUES=$qn python bench.py --L 5000000 --genomes 2 --steps 10 --warmup 2 --no-cpu --no-extra --no-check --jobs 16 > $O/c2_jobs16_q$qn.json 2> $O/c2_q$qn.err
  GPU_MAX_HW_QUEUES=$qn python bench.py --config c5 --steps 3 --warmup 1 --no-cpu --no-check > $O/c5_q$qn.json 2> $O/c5_q$qn.err
done
python - <<'P'
import json
for qn in (4, 8, 16, 32):
    for f in ("c2_jobs16", "c5"):
        try:
            d = json.loads(open("gpurun_out/r6queues/%s_q%d.json" % (f, qn)).read().strip().splitlines()[-1])
            print(qn, f, round(d["ms_per_step"], 3), round(d["value"]))
        except Exception as e:
            print(qn, f, "failed", e)
P
