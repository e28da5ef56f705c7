# FETCH_SIZE per kernel of tools/ubench/fetch_calib.bin -> factor per access width (run on the GPU box)
OUT=$PWD/gpurun_out/fetch_calib; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- $R/tools/ubench/fetch_calib.bin > $OUT/log.txt 2>&1
cd $R
python - <<'PY' > gpurun_out/fetch_calib/factors.txt
import csv, glob, collections
per = collections.defaultdict(list)
for f in glob.glob("gpurun_out/fetch_calib/f/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            d[r["Dispatch_Id"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    for k, v in d.items():
        per[name[k]].append(v)
print("# FETCH_SIZE (KB) per launch over 1 GiB = 1048576 KB of distinct bytes; factor = bytes / (FETCH_SIZE x 1024)")
for k in sorted(per):
    v = per[k]
    print("%-10s launches %d  FETCH_SIZE_KB %s  factor %.3f" % (k, len(v), " ".join("%.0f" % x for x in v), 1048576.0 / (sum(v) / len(v))))
PY
cat gpurun_out/fetch_calib/factors.txt
