#!/usr/bin/env python3
"""Register / scratch use of every kernel of one source file: hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel.
usage: tools/kernel_regs.py <file.hip> [filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "")
        cur = re.sub(r"\(.*", "", cur).replace("dns::sp::", "").replace("dns::", "").replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, r in rows.items():
    if flt in k:
        print(f"{k:70s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):4d} scratch {r.get('ScratchSize [bytes/lane]', -1):5d} "
              f"occ {r.get('Occupancy [waves/SIMD]', -1)} spillV {r.get('VGPRs Spill', -1)} LDS {r.get('LDS Size [bytes/block]', -1)}")
