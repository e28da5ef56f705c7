#!/bin/bash
# round 6: passing ranks four at a time (one per 16-lane row); the native picker's calls on host threads; concurrent small jobs
O=gpurun_out/r6scan6; mkdir -p $O
python -m pytest tests/test_gpu_cascade.py tests/test_gpu_golden.py tests/test_gpu_align.py tests/test_gpu_graphrem.py tests/test_gpu_handoff.py -x -q -k "not processes" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 2 --no-cpu > $O/c3.json 2> $O/c3.err; tail -c 300 $O/c3.err
for j in 1 4 8 16; do python bench.py --L 5000000 --genomes 2 --steps 10 --warmup 2 --no-cpu --no-extra --jobs $j > $O/c2_jobs$j.json 2> $O/c2_jobs$j.err; done
python - <<'P'
import json
for f in ("c3", "c2_jobs1", "c2_jobs4", "c2_jobs8", "c2_jobs16"):
    try:
        d = json.loads(open("gpurun_out/r6scan6/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), round(d["value"]), {k: round(v, 4) if isinstance(v, float) else v for k, v in d["roofline"].items() if k in ("frac", "avg_us", "launches", "achieved")},
              d.get("breakdown_ms_per_step"), d["parity"]["full_size"].get("all") if isinstance(d["parity"]["full_size"], dict) else None)
        if "level_pipeline" in d: print("  level_pipeline", d["level_pipeline"]["ms_per_step"], d["level_pipeline"]["kernel_classes_ms_per_step"], d["level_pipeline"]["golden"])
    except Exception as e:
        print(f, "failed", e)
P
bash tools/prof_cmd.sh r6scan6_prof --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra
grep -E "k_full_scan|k_multi_pick|k_mp_zero" gpurun_out/r6scan6_prof/kernel_stats.txt
timeout 200 python tools/fuzz.py 120 6401 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
python tools/time_native.py > $O/time_native.txt 2>&1; tail -12 $O/time_native.txt
RV_PICK_THREADS=1 python tools/time_native.py > $O/time_native_1thread.txt 2>&1; tail -4 $O/time_native_1thread.txt
