#!/usr/bin/env python3
"""Instruction STREAM of one kernel in a device assembly file (hipcc -S --offload-device-only) as one letter per instruction:
M matrix, v vector, a accvgpr move, l LDS read, w LDS write, g global load, s global store / atomic, W s_waitcnt, . scalar,
B branch, | label, n s_nop -- to see at a glance whether the vector work sits BETWEEN the matrix instructions or in front of them.
usage: tools/isa_stream.py file.s <substring of the mangled name> [width]"""
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
flt = sys.argv[2]
width = int(sys.argv[3]) if len(sys.argv) > 3 else 160
cur = None
out = []
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1) if flt in m.group(1) else None
        continue
    if line.startswith(".Lfunc_end"):
        cur = None
    if cur is None:
        continue
    if re.match(r"^\.LBB\w+:", line):
        out.append("|")
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_mfma"): c = "M"
    elif op.startswith("v_accvgpr"): c = "a"
    elif op.startswith("v_"): c = "v"
    elif op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"): c = "l"
    elif op.startswith("ds_"): c = "w"
    elif op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): c = "g"
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): c = "s"
    elif op.startswith("scratch_"): c = "X"
    elif op.startswith("s_waitcnt"): c = "W"
    elif op.startswith("s_nop"): c = "n"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): c = "B"
    elif op.startswith("s_barrier"): c = "#"
    else: c = "."
    out.append(c)
s = "".join(out)
for i in range(0, len(s), width):
    print(s[i:i + width])
