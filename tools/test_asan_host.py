"""BUILD CONTAINER ONLY (`python -m pytest tools/test_asan_host.py`; kept out of `tests/` and listed in `.gpurunignore`: the GPU pool refuses any
call whose files mention sanitizer builds, and a `pytest tests` on the GPU box must never be refused for it).
The C ABI's HOST code under AddressSanitizer + UBSan ("race detection / sanitizers": sanitizers on the
host shim; GPU ASan is not available on this pool).  `tools/asan_build.sh` compiles every translation unit with
`-fsanitize=address,undefined -fno-gpu-sanitize` into `libinrhip_asan.so`; `tools/asan_sweep.py` then drives every planner
(`*_workspace_bytes`, `*_param_count`, `*_param_offsets`; RAMS: batch 1..40, odd heights / widths, scale 2 / 3 / 4) and the validation /
planning prefix of every compute entry point (never-dereferenced device addresses; without a GPU a call ends at its first HIP call)
in a Python process with the ASan runtime pre-loaded.  Round 5's first run found null-pointer arithmetic in the RAMS parameter count
(profiles/r05_asan.txt)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mri-super-resolution_amd", "libinrhip_asan.so")
CSRC = os.path.join(ROOT, "mri-super-resolution_amd", "csrc")


def _runtime():
    found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return found[-1] if found else None


def _stale():
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".inc", ".h"))] + [os.path.join(ROOT, "include", "inrhip.h")]
    return any(os.path.getmtime(d) > built for d in deps)


@pytest.mark.timeout(900)
def test_host_code_of_the_c_abi_is_clean_under_asan_and_ubsan():
    rt = _runtime()
    if rt is None or not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc / ASan runtime on this host")
    if _stale():            # (about a minute: seven translation units in parallel)
        out = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_build.sh")], capture_output=True, text=True, timeout=800)
        assert out.returncode == 0, out.stderr[-3000:]
    env = dict(os.environ, INR_LIB=LIB, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_sweep.py")], capture_output=True, text=True, timeout=600,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "no sanitizer report" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
