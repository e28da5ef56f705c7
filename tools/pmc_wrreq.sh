# write requests of every kernel by size (TCC_EA0_WRREQ: all, _64B: the full ones): bash tools/pmc_wrreq.sh OUTNAME <python args...>
OUT=$PWD/gpurun_out/$1; shift; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/pmc_wr -o w -- python "$@" > $OUT/pmc_wr.log 2>&1
cd $R
python - $OUT/pmc_wr > $OUT/wrreq.txt <<'PY'
import csv, glob, os, re, sys
d = sys.argv[1]
per = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:48]
        e = per.setdefault(k, {})
        key = (r["Dispatch_Id"], r["Counter_Name"])
        e[key] = e.get(key, 0.0) + float(r["Counter_Value"])
print("%-50s %6s %14s %14s %8s %12s" % ("kernel", "calls", "wrreq/launch", "64B/launch", "64B frac", "MB/launch"))
rows = []
for k, e in per.items():
    disp = sorted({d for d, _ in e})
    tot = sum(v for (d, c), v in e.items() if c == "TCC_EA0_WRREQ_sum") / len(disp)
    b64 = sum(v for (d, c), v in e.items() if c == "TCC_EA0_WRREQ_64B_sum") / len(disp)
    rows.append((tot, k, len(disp), b64))
for tot, k, n, b64 in sorted(rows, reverse=True)[:24]:
    print("%-50s %6d %14.0f %14.0f %8.3f %12.1f" % (k, n, tot, b64, b64 / tot if tot else 0, (b64 * 64 + (tot - b64) * 32) / 1e6))
PY
rm -rf $OUT/pmc_wr
cat $OUT/wrreq.txt
