#!/usr/bin/env bash
# Rebuild only the named sources (default: none) and relink libcruse_hip.so from build/*.o.  usage: relink.sh [gru_tf gemm_bf16 ...]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in "$@"; do
  if [ "$f" = conv_mfma ]; then
    for k in 0 1 2 3; do "$HIPCC" $FLAGS -DCM_TU=$k -c "$here/conv_mfma.hip" -o "$here/build/conv_mfma_$k.o" & pids+=($!); done
    continue
  fi
  if [ "$f" = abi ]; then "$HIPCC" $FLAGS -x hip -c "$here/abi.cpp" -o "$here/build/abi.o" & else "$HIPCC" $FLAGS -c "$here/$f.hip" -o "$here/build/$f.o" & fi
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for f in stft stft_general conv conv_mfma_0 conv_mfma_1 conv_mfma_2 conv_mfma_3 wgrad_mfma wgrad_rd pointwise gemm gemm_bf16 gru gru_tf gru_w16 tdloss deepfilter generic extras abi; do objs+=("$here/build/$f.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$here/../libcruse_hip.so"
echo "relinked"
