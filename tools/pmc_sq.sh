# wave-cycle and LDS counters of every kernel: bash tools/pmc_sq.sh OUTNAME <python args...>   (one counter set per pass)
OUT=$PWD/gpurun_out/$1; shift; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o w -- python "$@" > $OUT/pmc$i.log 2>&1
done
cd $R
python - $OUT > $OUT/sq.txt <<'PY'
import csv, glob, os, re, sys
d = sys.argv[1]
per = {}
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:34]
        e = per.setdefault(k, {})
        e.setdefault("_d" + r["Counter_Name"], set()).add(r["Dispatch_Id"])
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = []
for k, e in per.items():
    v = {c: e[c] / max(1, len(e["_d" + c])) for c in e if not c.startswith("_d")}
    rows.append((v.get("SQ_WAVE_CYCLES", 0), k, v))
for _, k, v in sorted(rows, reverse=True)[:14]:
    w = max(v.get("SQ_WAVES", 1), 1)
    print(k)
    print("   " + "  ".join("%s=%.3g" % (c[3:], x) for c, x in sorted(v.items())))
    print("   per wave: " + "  ".join("%s=%.0f" % (c[3:], x / w) for c, x in sorted(v.items()) if c != "SQ_WAVES"))
PY
rm -rf $OUT/pmc?
cat $OUT/sq.txt
