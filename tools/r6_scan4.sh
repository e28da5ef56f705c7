#!/bin/bash
# round 6, fourth pass: region counters in different L2 channels; the pair scan with 16 ranks per lane and byte-parallel tests
O=gpurun_out/r6scan4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 2 --no-cpu --no-extra > $O/c3.json 2> $O/c3.err; tail -c 300 $O/c3.err
python bench.py --steps 10 --warmup 2 --no-cpu --no-extra > $O/c4.json 2> $O/c4.err; tail -c 300 $O/c4.err
python - <<'P'
import json
for f in ("c3", "c4"):
    try:
        d = json.loads(open("gpurun_out/r6scan4/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), round(d["value"]), {k: round(v, 4) if isinstance(v, float) else v for k, v in d["roofline"].items() if k in ("frac", "avg_us", "launches", "achieved")},
              d.get("breakdown_ms_per_step"), d["parity"]["full_size"].get("all"))
    except Exception as e:
        print(f, "failed", e)
P
bash tools/prof_cmd.sh r6scan4_prof --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra
grep -E "k_full_scan|k_multi_pick|k_mp_zero|k_casm_so|k_casm_witness|k_casm_collect" gpurun_out/r6scan4_prof/kernel_stats.txt
python tools/scan_probe.py 250000000 > $O/probe_pair.txt 2>&1; cat $O/probe_pair.txt
timeout 300 python tools/fuzz.py 200 6301 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
