"""One step of bench.py from a rocprofv3 --kernel-trace CSV: start us, end us, duration us, stream, kernel (t = 0: end of the previous
Adam step), then per stream the busy time and the gaps.  usage: python tools/step_trace.py <dir with *kernel_trace.csv> [step from the end = 2]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in csv.DictReader(open(f))]
rows.sort()
ends = [e for s, e, n, q in rows if "adam_step_kernel" in n]
t0, t1 = ends[-back - 1], ends[-back]
sel = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and e <= t1]
qs = {q: i for i, q in enumerate(sorted({q for *_, q in sel}))}
print(f"one step of bench.py under rocprofv3 --kernel-trace: {(t1 - t0) / 1e3:.1f} us, {len(sel)} kernels; start us, end us, duration us, stream, kernel")
for s, e, n, q in sel:
    k = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[-60:]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{qs[q]} {k}")
for q, i in qs.items():
    mine = [(s, e) for s, e, n, qq in sel if qq == q]
    busy = sum(e - s for s, e in mine)
    print(f"stream q{i}: {len(mine)} kernels, busy {busy / 1e3:.1f} us, first start {(mine[0][0] - t0) / 1e3:.1f}, last end {(mine[-1][1] - t0) / 1e3:.1f}")
