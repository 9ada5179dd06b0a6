"""Sum rocprofv3 --pmc counter CSVs (tools/sq_counters.sh passes) per kernel: python tools/sq_summary.py OUTDIR [filter]"""
import csv, glob, os, sys, collections
out = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
        if os.environ.get("SQ_BY_LDS"):               # same-named instances of different translation units (64 x 2 / 32 x 1 networks)
            k += f" [kernel id {r.get('Kernel_Id', '?')}, {r.get('VGPR_Count', '?')}+{r.get('Accum_VGPR_Count', '?')} registers]"
        if flt and flt not in r["Kernel_Name"]:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / max(cnt[(k, c)], 1):16.0f} per launch")
