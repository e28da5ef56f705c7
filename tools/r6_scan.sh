#!/bin/bash
# round 6: the streaming multi-sample scans (k_full_scan, k_scan_multi): tests, C3 before / after, kernel stats, PMC traffic of the root scan
O=gpurun_out/r6scan; mkdir -p $O
python -m pytest tests/test_gpu_cascade.py tests/test_gpu_golden.py tests/test_gpu_align.py tests/test_gpu_callbacks.py tests/test_gpu_preselect.py -x -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 2 --no-cpu > $O/c3.json 2> $O/c3.err; tail -c 600 $O/c3.err
RV_SCAN_V1=1 python bench.py --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra > $O/c3_v1.json 2> $O/c3_v1.err
python - <<'P'
import json
for f in ("c3", "c3_v1"):
    try:
        d = json.loads(open("gpurun_out/r6scan/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), {k: round(v, 4) if isinstance(v, float) else v for k, v in d["roofline"].items() if k in ("frac", "avg_us", "launches", "achieved")},
              d["parity"]["full_size"].get("all") if isinstance(d["parity"]["full_size"], dict) else None, d.get("breakdown_ms_per_step"))
        if "level_pipeline" in d: print("  level_pipeline", d["level_pipeline"]["ms_per_step"], d["level_pipeline"]["kernel_classes_ms_per_step"], d["level_pipeline"]["golden"])
    except Exception as e:
        print(f, "failed", e)
P
bash tools/prof_cmd.sh r6scan_prof --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra
head -45 gpurun_out/r6scan_prof/kernel_stats.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/bench.py --L 5000000 --genomes 10 --steps 3 --warmup 1 --no-cpu --no-extra --no-check > $R/$O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o w -- python $R/bench.py --L 5000000 --genomes 10 --steps 3 --warmup 1 --no-cpu --no-extra --no-check > $R/$O/pmc_w.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write "k_full_scan<0>" "10x5000000-32" > $O/pmc_scan_10x5000000-32.json; cat $O/pmc_scan_10x5000000-32.json
rm -rf $O/pmc_fetch $O/pmc_write
timeout 300 python tools/fuzz.py 200 6101 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
