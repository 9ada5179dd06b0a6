#!/bin/bash
# Device assembly + resource usage of ONE instance of the half-rows MLP backward kernel (seconds instead of the translation unit's minute).
# usage: tools/isa_half.sh NN NL "NS0, MT, WITH_DW" out.s    e.g. tools/isa_half.sh 64 2 "5, 2, true" /tmp/h.s
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
printf '#define DNS_BWD_NN %s\n#define DNS_BWD_NL %s\n#include "mlp_half_bwd.inc"\n' "$1" "$2" > $tmp/one.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I$here/dns_slam_amd/csrc "-DDNS_HBWD_DEV_ONE=$3" $EXTRA \
  -S --offload-device-only -Rpass-analysis=kernel-resource-usage $tmp/one.hip -o "$4" 2> $tmp/res.txt || { cat $tmp/res.txt; exit 1; }
grep -E "VGPRs:|AGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|SGPRs:|LDS Size" $tmp/res.txt | sed 's/.*remark: *//' | tr '\n' ';'; echo
python3 $here/tools/isa_count.py "$4" mlp_half_bwd_kernel
rm -rf $tmp
