#!/bin/bash
# Host-side sanitizer build of the C-ABI library ("race detection / sanitizers"): AddressSanitizer +
# UndefinedBehaviorSanitizer on the HOST code of every translation unit (argument validation, workspace planners, parameter
# walks, launch geometry), device code compiled as usual (-fno-gpu-sanitize: GPU ASan is not available on this pool).
# Output: mri-super-resolution_amd/libinrhip_asan.so (git-ignored).  Run on the CPU box only:
#   tools/asan_build.sh && tools/asan_run.sh          -> profiles/r05_asan.txt
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/mri-super-resolution_amd/csrc"
OBJ="$CSRC/_obj/asan"
OUT="$ROOT/mri-super-resolution_amd/libinrhip_asan.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
pids=()
for src in api gemm_f32 kernels metrics rams siren_small hybrid_fit; do
  extra=""
  if [ "$src" = gemm_f32 ]; then extra="-Xclang -target-feature -Xclang -packed-fp32-ops"; fi
  ( "$HIPCC" $FLAGS $extra -I "$ROOT/include" -I "$CSRC" -c "$CSRC/$src.hip" -o "$OBJ/$src.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$OUT.tmp" "$OBJ"/*.o
mv "$OUT.tmp" "$OUT"
echo "$OUT"
