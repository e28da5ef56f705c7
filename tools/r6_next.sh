#!/bin/bash
# round 6: the whole GPU suite (incl. the hand-off over HIP IPC), the cascade's level loop on a high-priority stream, reveal rem after the graph work
O=gpurun_out/r6next; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for p in 0 1; do
  RV_CASCADE_PRIO=$p python bench.py --config c5 --jobs 10 --steps 3 --warmup 1 --no-cpu --no-check > $O/c5_prio$p.json 2> $O/c5_prio$p.err
  RV_CASCADE_PRIO=$p python bench.py --L 5000000 --genomes 2 --steps 10 --warmup 2 --no-cpu --no-extra --no-check --jobs 16 > $O/c2j16_prio$p.json 2> $O/c2j16_prio$p.err
  RV_CASCADE_PRIO=$p python bench.py --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra > $O/c3_prio$p.json 2> $O/c3_prio$p.err
done
python - <<'P'
import json
for f in ("c5_prio0", "c5_prio1", "c2j16_prio0", "c2j16_prio1", "c3_prio0", "c3_prio1"):
    try:
        d = json.loads(open("gpurun_out/r6next/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), round(d["value"]))
    except Exception as e:
        print(f, "failed", e)
P
python tools/time_native.py > $O/time_native.txt 2>&1; tail -2 $O/time_native.txt | cut -c1-400
# two ranks on this box's one GPU (gloo): the N > 1 line's plumbing -- per-rank headline, the stream leg on every rank, no divided leg for two samples; then three samples divided over HIP IPC
RV_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --L 50000000 --genomes 2 --steps 2 --warmup 1 --no-cpu > $O/bench_2ranks_shared.json 2> $O/bench_2ranks_shared.err; tail -c 400 $O/bench_2ranks_shared.err
RV_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --L 40000000 --genomes 3 --steps 2 --warmup 1 --no-cpu --no-extra > $O/bench_2ranks_divide.json 2> $O/bench_2ranks_divide.err; tail -c 400 $O/bench_2ranks_divide.err
python - <<'P'
import json
for f in ("bench_2ranks_shared", "bench_2ranks_divide"):
    try:
        d = json.loads(open("gpurun_out/r6next/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["n_gpus"], round(d["ms_per_step"], 2), round(d["value"]), "divide:", d.get("divide"), "stream:", (d.get("stream") or {}).get("value"), (d.get("stream") or {}).get("failed"))
    except Exception as e:
        print(f, "failed", e)
P
