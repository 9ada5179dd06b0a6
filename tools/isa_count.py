#!/usr/bin/env python3
"""Static instruction mix of the kernels in a device assembly file (hipcc -S --offload-device-only): per function the number of
matrix / vector / scalar / LDS / global instructions and the most frequent vector opcodes.
usage: tools/isa_count.py file.s [substring of the mangled name]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, c, out = None, None, []
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur, c = m.group(1), collections.Counter()
        out.append((cur, c))
        continue
    if line.startswith(".Lfunc_end"):
        cur = None
    if cur is None:
        continue
    m = re.match(r"^\s+(v_mfma\w+|v_\w+|s_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)", line)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_mfma"):
        c["mfma"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
        c[op] += 1
    elif op.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
        c[op] += 1
    elif op.startswith("global_load") or op.startswith("buffer_load"):
        c["gload"] += 1
    elif op.startswith("global_store") or op.startswith("global_atomic"):
        c["gstore"] += 1
    elif op.startswith("scratch"):
        c["scratch"] += 1
for name, c in out:
    if flt in name:
        print(name)
        print("   ", {k: c[k] for k in ("mfma", "valu", "salu", "lds", "gload", "gstore", "scratch", "waitcnt")})
        print("   ", [(k, v) for k, v in c.most_common(60) if k.startswith("v_") or k.startswith("ds_")][:26])
