#!/bin/bash
# round 5: two-sample classes through the interval cascade (RV_CASCADE_SECOND=2) now that it rebuilds large undecided sub-indices
mkdir -p gpurun_out/second
export RV_CASCADE_SECOND=2 RV_CASM_BIG_ROOT=2000000000 RV_CASM_BIG_TOTAL=2000000000 RV_CASCADE_LOG=1
for c in "$@"; do
  timeout 300 python bench.py --class-one $c --L 250000000 > gpurun_out/second/$c.json 2> gpurun_out/second/$c.err
  python - $c <<'P'
import json, sys
c = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/second/%s.json" % c).read().strip().splitlines()[-1])
    print(c, round(d["ms_per_step"], 2), d.get("path"), d.get("cascade_why"), d.get("properties"), d.get("golden"))
except Exception as e:
    print(c, "failed", e)
P
  grep "cascade (" gpurun_out/second/$c.err | tail -1
done
