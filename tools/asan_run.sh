#!/bin/bash
# Runs the host-side sweep (tools/asan_sweep.py) and tests/test_abi_cpu.py against libinrhip_asan.so with the ASan runtime
# pre-loaded into the Python process; writes profiles/r05_asan.txt.  CPU box only (GPU ASan is not available on this pool).
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LIBASAN="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
OUT="${1:-$ROOT/profiles/r05_asan.txt}"
export INR_LIB="$ROOT/mri-super-resolution_amd/libinrhip_asan.so"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98"
cd "$ROOT"
{
  echo "# host-side ASan + UBSan run of the C ABI ($(date -u +%Y-%m-%dT%H:%MZ), git $(git rev-parse --short HEAD))"
  echo "# library: $INR_LIB ($(stat -c %s "$INR_LIB") bytes), runtime: $LIBASAN"
  echo "## tools/asan_sweep.py"
  LD_PRELOAD="$LIBASAN" python3 tools/asan_sweep.py 2>&1 | tail -40
  echo "exit code: ${PIPESTATUS[0]}"
  echo "## tests/test_abi_cpu.py tests/test_compat_cpu.py"
  LD_PRELOAD="$LIBASAN" python3 -m pytest tests/test_abi_cpu.py tests/test_compat_cpu.py -q -p no:cacheprovider 2>&1 | tail -15
  echo "exit code: ${PIPESTATUS[0]}"
} > "$OUT"
cat "$OUT"
