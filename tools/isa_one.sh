#!/bin/bash
# Device assembly + resource usage of ONE instance of the MLP backward kernel (seconds instead of the translation unit's minute).
# usage: tools/isa_one.sh NN NL "NCH, MT, PREC, WITH_DW[, SH, XS, NW]" out.s    e.g. tools/isa_one.sh 64 2 "3, 2, 3, true" /tmp/k.s
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
printf '#define DNS_BWD_NN %s\n#define DNS_BWD_NL %s\n#include "mlp_split_bwd.inc"\n' "$1" "$2" > $tmp/one.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I$here/dns_slam_amd/csrc "-DDNS_BWD_DEV_ONE=$3" $EXTRA \
  -S --offload-device-only -Rpass-analysis=kernel-resource-usage $tmp/one.hip -o "$4" 2> $tmp/res.txt || { cat $tmp/res.txt; exit 1; }
grep -E "VGPRs:|AGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|SGPRs:" $tmp/res.txt | sed 's/.*remark: *//' | tr '\n' ';'; echo
python3 $here/tools/isa_count.py "$4" mlp_bwd_kernel
rm -rf $tmp
