#!/bin/bash
# round 6: rv_batch_run -- the level loops of the anchor cascades of a batch's jobs as one set of launches
O=gpurun_out/r6batch; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 600 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_golden.py -x -q > $O/tests2.txt 2>&1; tail -3 $O/tests2.txt
for bs in 0 10 20; do
  timeout 600 python bench.py --config c5 --batch $bs --jobs 10 --steps 3 --warmup 1 --no-cpu > $O/c5_b$bs.json 2> $O/c5_b$bs.err
done
python - <<'P'
import json
for bs in (0, 10, 20):
    try:
        d = json.loads(open("gpurun_out/r6batch/c5_b%d.json" % bs).read().strip().splitlines()[-1])
        print(bs, round(d["ms_per_step"], 2), round(d["value"]), d["config"].get("batched"), d["properties_full_size"], (d["parity"]["full_size"] or {}).get("all") if isinstance(d["parity"]["full_size"], dict) else None)
    except Exception as e:
        print(bs, "failed", e)
P
RV_CASCADE_LOG=1 python bench.py --config c5 --batch 20 --jobs 10 --steps 2 --warmup 1 --no-cpu --no-check 2>&1 | grep "cascade (batch)" | tail -3
