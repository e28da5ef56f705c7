#!/bin/bash
# round 6, third pass: k_full_scan with 16 ranks per lane, byte-parallel evidence bits, AND-doubling windows, cooperative survivors
O=gpurun_out/r6scan3; mkdir -p $O
python -m pytest tests/test_gpu_cascade.py tests/test_gpu_golden.py tests/test_gpu_align.py tests/test_gpu_handoff.py -x -q -k "not processes" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 2 --no-cpu > $O/c3.json 2> $O/c3.err; tail -c 300 $O/c3.err
python - <<'P'
import json
for f in ("c3",):
    try:
        d = json.loads(open("gpurun_out/r6scan3/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), round(d["value"]), {k: round(v, 4) if isinstance(v, float) else v for k, v in d["roofline"].items() if k in ("frac", "avg_us", "launches", "achieved")},
              d.get("breakdown_ms_per_step"), d["parity"]["full_size"].get("all"))
        if "level_pipeline" in d: print("  level_pipeline", d["level_pipeline"]["ms_per_step"], d["level_pipeline"]["kernel_classes_ms_per_step"], d["level_pipeline"]["golden"])
    except Exception as e:
        print(f, "failed", e)
P
bash tools/prof_cmd.sh r6scan3_prof --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra
grep -E "k_full_scan|k_multi_pick|k_mp_zero|k_casm_so|k_casm_witness" gpurun_out/r6scan3_prof/kernel_stats.txt
timeout 200 python tools/fuzz.py 120 6201 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
python -m pytest tests/test_gpu_fullsize_golden.py -x -q > $O/tests_fullsize.txt 2>&1; tail -3 $O/tests_fullsize.txt
