#!/usr/bin/env bash
# Build libcruse_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libcruse_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
objs=()
pids=()
mkdir -p "$here/build"
for k in 0 1 2 3; do
  "$HIPCC" $FLAGS -DCM_TU=$k -c "$here/conv_mfma.hip" -o "$here/build/conv_mfma_$k.o" &
  pids+=($!)
  objs+=("$here/build/conv_mfma_$k.o")
done
for f in stft stft_general conv wgrad_mfma wgrad_rd pointwise gemm gemm_bf16 gru gru_tf gru_w16 tdloss deepfilter generic extras; do
  "$HIPCC" $FLAGS -c "$here/$f.hip" -o "$here/build/$f.o" &
  pids+=($!)
  objs+=("$here/build/$f.o")
done
"$HIPCC" $FLAGS -x hip -c "$here/abi.cpp" -o "$here/build/abi.o"
objs+=("$here/build/abi.o")
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"
