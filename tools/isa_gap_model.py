#!/usr/bin/env python3
"""Issue-time model of one kernel from its device assembly, per segment between in-kernel stamps (s_memtime; build the instance
with -DDNS_BWD_TRACE: EXTRA=-DDNS_BWD_TRACE tools/isa_one.sh 64 2 "3, 2, 3, true, false, false, 4" /tmp/k.s).

MI355X_MICROARCH, one wave per SIMD: a vector / LDS / memory instruction holds the wave's issue for ~4 cycles, `s_nop N` for
4 (N + 1), a v_mfma_f32_32x32x16 for 8 of its 32; costs add, and the gap between two matrix instructions runs max(32, sum).
Per segment: n = matrix instructions, lin = sum of the issue costs, mod = sum over gaps of max(32, cost), w = s_waitcnt count.
`mod` is the segment's time if nothing ever waits; what the stamps measure beyond it is waits.
usage: tools/isa_gap_model.py file.s [substring of the mangled kernel name = mlp_bwd_kernel]"""
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
flt = sys.argv[2] if len(sys.argv) > 2 else "mlp_bwd_kernel"
cur, seg, cost, res = None, 0, 0, {}
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = flt in m.group(1)
        continue
    if line.startswith(".Lfunc_end"):
        cur = None
    if not cur:
        continue
    if "s_memtime" in line:
        r = res.setdefault(seg, {"n": 0, "lin": 0, "mod": 0, "w": 0, "valu": 0, "lds": 0, "vmem": 0})
        r["lin"] += cost
        r["mod"] += cost
        seg, cost = seg + 1, 0
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", line)
    if not m:
        continue
    op, rest = m.group(1), m.group(2)
    r = res.setdefault(seg, {"n": 0, "lin": 0, "mod": 0, "w": 0, "valu": 0, "lds": 0, "vmem": 0})
    if op.startswith("v_mfma"):
        r["lin"] += cost
        r["mod"] += max(32, cost) if r["n"] else cost
        r["n"] += 1
        cost = 8
        continue
    c = 0
    if op.startswith("v_"):
        c, r["valu"] = 4, r["valu"] + 1
    elif op.startswith("s_nop"):
        c = 4 * (int(rest.split()[0]) + 1)
    elif op.startswith("ds_"):
        c, r["lds"] = 4, r["lds"] + 1
    elif op.startswith("buffer_") or op.startswith("global_"):
        c, r["vmem"] = 4, r["vmem"] + 1
    elif op.startswith("s_waitcnt"):
        r["w"] += 1
    elif op.startswith("s_"):
        c = 1
    cost += c
for s in sorted(res):
    print(s, res[s])
