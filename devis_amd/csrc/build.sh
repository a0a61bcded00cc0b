#!/bin/bash
# Build libdevis_hip.so in-tree for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# packed fp32 VALU ops are off: v_pk_fma_f32 gave run-to-run different low lanes when kernels of another stream
# shared the CU; none of these kernels is VALU-bound enough to miss them
# (-mllvm -amdgpu-mfma-vgpr-form was measured: the prologue variant of the 16-channel conv gains 7 %, its residual
# variant and the head dgrad lose 25 % - not used)
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
mkdir -p build
pids=()
for f in layout upcat conv_direct conv_f32_mfma conv_mfma conv_march conv_s2_march conv_march32 conv_deepk conv2d_mfma conv_s2_dgrad conv_thin conv_wgrad_mfma conv_wgrad_march conv_wgrad_s2_march pointwise_mfma gemm_convt head1x1 pointwise_wgrad_mfma norm attgate evid evalmaps optim capi; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ mfma_common.h -nt build/$f.o ] || [ wgrad_reduce.h -nt build/$f.o ] || [ ../../include/devis_hip.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libdevis_hip.so
echo "built $(realpath ../libdevis_hip.so)"
