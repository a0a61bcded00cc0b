#!/bin/bash
# Build devis_amd/libdevis_hip_alt.so: the shipped objects with ONE source recompiled under extra flags, for same-box
# A/B timing (DEVIS_HIP_LIB=devis_amd/libdevis_hip_alt.so python tools/...).   tools/build_alt.sh conv_march -DDEVIS_MARCH_MINW=3
set -e
cd "$(dirname "$0")/../devis_amd/csrc"
f=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
mkdir -p build_alt
/opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o build_alt/$f.o
objs=""
for o in build/*.o; do b=$(basename $o); if [ "$b" = "$f.o" ]; then objs="$objs build_alt/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../libdevis_hip_alt.so
echo "built ../libdevis_hip_alt.so ($f $@)"
