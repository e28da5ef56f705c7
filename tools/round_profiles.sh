#!/bin/bash
# the records under profiles/ of a round (round 6 as committed): bash tools/round_profiles.sh -- -- kernel statistics and PMC traffic of the two judged scans, bench lines of C2 / C3 / C4 / C5 level 0
O=gpurun_out/r6prof; mkdir -p $O
R=$PWD
pmc() {   # name kernel workload bench-args...
  local name=$1 kernel=$2 wl=$3; shift 3
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-extra --no-check > $R/$O/pmc_f_$name.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o w -- python $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-extra --no-check > $R/$O/pmc_w_$name.log 2>&1
  cd $R
  python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write "$kernel" "$wl" > $O/pmc_scan_$wl.json; cat $O/pmc_scan_$wl.json
  rm -rf $O/pmc_fetch $O/pmc_write
}
pmc c3 "k_full_scan<0>" "10x5000000-32" --L 5000000 --genomes 10
pmc c4 "k_scan_pair" "2x250000000-32"
bash tools/prof_cmd.sh r6prof_c4 --steps 5 --warmup 1 --no-cpu --no-extra; cp gpurun_out/r6prof_c4/kernel_stats.txt $O/kernel_stats_c4.txt
bash tools/prof_cmd.sh r6prof_c3 --L 5000000 --genomes 10 --steps 10 --warmup 2 --no-cpu --no-extra; cp gpurun_out/r6prof_c3/kernel_stats.txt $O/kernel_stats_c3.txt
cp $O/pmc_scan_*.json profiles/ 2>/dev/null      # (bench.py reads the traffic figure from profiles/ on this box)
python bench.py --L 5000000 --genomes 10 --steps 20 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --L 5000000 --genomes 2 --steps 20 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu > $O/bench_c5_level0.json 2> $O/bench_c5.err
python bench.py --steps 20 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err
python - <<'P'
import json
for f in ("bench_c2", "bench_c3", "bench_c4", "bench_c5_level0"):
    try:
        d = json.loads(open("gpurun_out/r6prof/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, round(d["ms_per_step"], 3), round(d["value"]), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("frac", "avg_us", "launches", "traffic", "frac_of_bytes_streamed", "frac_of_traffic")},
              (d.get("parity") or {}).get("full_size", {}).get("all") if isinstance((d.get("parity") or {}).get("full_size"), dict) else None,
              "lp", (d.get("level_pipeline") or {}).get("ms_per_step"), "stream", (d.get("stream") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e)
P
python -m pytest tests -m gpu -q > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
python tools/time_native.py > $O/time_native.txt 2>&1; tail -2 $O/time_native.txt | cut -c1-300
python bench.py --config stream --pairs 20 --steps 1 --warmup 1 > $O/bench_stream.json 2> $O/bench_stream.err
python tools/scan_probe.py 250000000 5000000 > $O/scan_probe_pair.txt 2>&1; python tools/scan_probe.py --multi 10 5000000 > $O/scan_probe_multi.txt 2>&1; cat $O/scan_probe_pair.txt $O/scan_probe_multi.txt | cut -c1-200
python bench.py --config c5 --batch 20 --steps 3 --warmup 1 --no-cpu > $O/bench_c5_level0_batched.json 2> $O/bench_c5b.err
python - <<'P'
import json
for f in ("bench_stream", "bench_c5_level0_batched"):
    try:
        d = json.loads(open("gpurun_out/r6prof/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 2), round(d["value"]))
    except Exception as e:
        print(f, "failed", e)
P
