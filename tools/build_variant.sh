#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..."  ->  dns_slam_amd/libdns_hip_NAME.so: the product library with the mlp3 translation
# units (bwd 64x2 only unless ALL=1) rebuilt with extra flags (timing experiments; load with DNS_HIP_LIB=...)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../dns_slam_amd/csrc"
mkdir -p build_$NAME
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
TUS="mlp3_bwd_64_2"
[ -n "$ALL" ] && TUS="mlp3_bwd_32_1 mlp3_bwd_32_2 mlp3_bwd_64_1 mlp3_bwd_64_2 mlp3_fwd"
for t in $TUS; do [ -f $t.hip ] && $HIPCC $BASE $FLAGS -c $t.hip -o build_$NAME/$t.o & done; wait
OBJS=""
for o in build/*.o; do b=$(basename $o); if [ -f build_$NAME/$b ]; then OBJS="$OBJS build_$NAME/$b"; else OBJS="$OBJS $o"; fi; done
TORCH_LIB=$(python3 -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
g++ -shared -fPIC $OBJS -L$TORCH_LIB -lamdhip64 -Wl,-rpath,$TORCH_LIB -o ../libdns_hip_$NAME.so
echo built ../libdns_hip_$NAME.so
