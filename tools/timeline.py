"""GPU timeline of bench.py steps from a rocprofv3 --kernel-trace CSV: wall per step, union-busy time (any kernel running),
idle gaps, and the per-kernel share; usage: python tools/timeline.py <dir with *kernel_trace.csv> [steps]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in csv.DictReader(open(f))]
rows.sort()
# steps are delimited by adam_step_kernel launches
ends = [e for s, e, n, q in rows if "adam_step_kernel" in n]
if len(ends) < nsteps + 1:
    print("not enough steps", len(ends)); sys.exit(1)
t0, t1 = ends[-nsteps - 1], ends[-1]
sel = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and e <= t1]
wall = (t1 - t0) / nsteps / 1e3
busy = 0; cur_s, cur_e = None, None
for s, e, n, q in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, n, q in sel)
print(f"steps {nsteps}: wall {wall:.1f} us/step, union-busy {busy / nsteps / 1e3:.1f} us/step, idle {wall - busy / nsteps / 1e3:.1f} us/step, "
      f"sum of kernel durations {tot / nsteps / 1e3:.1f} us/step, kernels/step {len(sel) / nsteps:.1f}")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in sel:
    k = n.split("(")[0].replace("void ", "")[-70:]
    agg[k][0] += 1; agg[k][1] += e - s
small = sum(v[1] for v in agg.values() if v[1] / v[0] < 10e3)
print(f"kernels shorter than 10 us: {sum(v[0] for v in agg.values() if v[1] / v[0] < 10e3) / nsteps:.1f} per step, {small / nsteps / 1e3:.1f} us/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {v[0] / nsteps:6.1f} x {v[1] / v[0] / 1e3:8.1f} us = {v[1] / nsteps / 1e3:8.1f} us/step  {k}")
