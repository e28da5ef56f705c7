#!/bin/bash
# round 6, second pass: streaming scans after the survivors' path lost its dependent loads; the seed hand-off; getmultimems' list growth
O=gpurun_out/r6scan2; mkdir -p $O
python -m pytest tests/test_gpu_cascade.py tests/test_gpu_golden.py tests/test_gpu_handoff.py tests/test_gpu_single_steps.py -x -q -k "not processes" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python -m pytest tests -m gpu -x -q -k "mems or multimums or graphrem or native" > $O/tests2.txt 2>&1; tail -3 $O/tests2.txt
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 2 --no-cpu > $O/c3.json 2> $O/c3.err; tail -c 300 $O/c3.err
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu > $O/c5.json 2> $O/c5.err; tail -c 300 $O/c5.err
python - <<'P'
import json
for f in ("c3", "c5"):
    try:
        d = json.loads(open("gpurun_out/r6scan2/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), round(d["value"]), {k: round(v, 4) if isinstance(v, float) else v for k, v in d["roofline"].items() if k in ("frac", "avg_us", "launches", "achieved")},
              d.get("breakdown_ms_per_step"))
        if "level_pipeline" in d: print("  level_pipeline", d["level_pipeline"]["ms_per_step"], d["level_pipeline"]["kernel_classes_ms_per_step"], d["level_pipeline"]["golden"])
    except Exception as e:
        print(f, "failed", e)
P
python tools/scan_probe.py --multi 10 5000000 > $O/probe_multi.txt 2>&1; cat $O/probe_multi.txt
python tools/scan_probe.py 250000000 > $O/probe_pair.txt 2>&1; cat $O/probe_pair.txt
bash tools/prof_cmd.sh r6scan2_prof --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra
head -30 gpurun_out/r6scan2_prof/kernel_stats.txt
