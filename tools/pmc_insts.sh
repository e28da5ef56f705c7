# instruction counters of every kernel: bash tools/pmc_insts.sh OUTNAME <python args...>
OUT=$PWD/gpurun_out/$1; shift; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/pmc -o w -- python "$@" > $OUT/pmc.log 2>&1
cd $R
python - $OUT/pmc > $OUT/insts.txt <<'PY'
import csv, glob, os, re, sys
d = sys.argv[1]
per = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:40]
        e = per.setdefault(k, {})
        e.setdefault("_d", set()).add(r["Dispatch_Id"])
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
print("%-42s %6s " % ("kernel", "calls") + " ".join("%14s" % n[3:] for n in names) + "   (per launch; per wave in brackets for VALU)")
rows = []
for k, e in per.items():
    n = len(e["_d"])
    rows.append((e.get("SQ_INSTS_VALU", 0) , k, n, [e.get(x, 0) / n for x in names]))
for _, k, n, v in sorted(rows, reverse=True)[:24]:
    print("%-42s %6d " % (k, n) + " ".join("%14.0f" % x for x in v) + "   [%.0f VALU/wave]" % (v[1] / v[0] if v[0] else 0))
PY
rm -rf $OUT/pmc
cat $OUT/insts.txt
