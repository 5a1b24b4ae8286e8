"""Builds libinrhip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
Translation units are compiled in parallel into ``csrc/_obj/*.o`` (only the stale ones) and linked.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ = os.path.join(CSRC, "_obj")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libinrhip.so")
SOURCES = ("api.hip", "gemm_f32.hip", "kernels.hip", "metrics.hip", "rams.hip", "siren_small.hip",
           "hybrid_fit.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Per-source code-generation flags.  The GEMM translation unit is compiled WITHOUT packed fp32 VALU instructions (v_pk_mul_f32 /
# v_pk_fma_f32): beside MFMAs they cost the deferred-epilogue GEMM 3 % (profiles/r04_nt_ab.txt); its VALU-bound kernels switch them
# back on one by one (INR_PACKED_F32 in csrc/gemm_hp.inc -- a kernel with the feature can inline helpers compiled without it, not
# the other way round, which is why the default is set here and not by an attribute on the GEMM kernels).
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
SOURCE_FLAGS = {"gemm_f32.hip": NO_PACKED_F32}


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
        [os.path.join(INCLUDE, "inrhip.h")]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in _sources()] + _headers() + [os.path.abspath(__file__)]   # the flags live in this file
    return any(os.path.getmtime(d) > built for d in deps)


def build_diagnostic(defines=("-DINR_STAMPS",), out=None, verbose=False, source_flags=True) -> str:
    """A diagnostic build (time stamps, ablations) into its OWN file -- never over the product library.  Select it with
    ``INR_LIB=<path>``; ``inr_build_flags()`` of such a library is non-zero and the binding refuses it otherwise.
    ``source_flags=False`` drops SOURCE_FLAGS (the A/B partner of the product's code-generation choices)."""
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = out or os.path.join(PKG_DIR, "libinrhip_diag.so")
    tmp = f"{out}.tmp.{os.getpid()}"
    with tempfile.TemporaryDirectory(dir=PKG_DIR, prefix="_diag_obj_") as objdir:
        jobs, objs = [], []
        for src in _sources():
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            extra = SOURCE_FLAGS.get(src, []) if source_flags else []
            jobs.append([hipcc] + FLAGS + extra + list(defines) + ["-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj])
            objs.append(obj)

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            return subprocess.run(cmd, capture_output=True, text=True)

        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as pool:
            for res in pool.map(run, jobs):
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        res = run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, out)          # readers only ever see a complete file
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources into ``libinrhip.so`` if missing or out of date; returns its path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libinrhip.so")
    os.makedirs(OBJ, exist_ok=True)
    # several ranks of one job may find the library stale at the same moment: one builds, the others wait and re-check
    import fcntl
    lock = open(os.path.join(OBJ, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not _stale():
            return LIB_PATH
        return _build_locked(hipcc, force, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(hipcc: str, force: bool, verbose: bool) -> str:
    newest_header = max(os.path.getmtime(h) for h in _headers() + [os.path.abspath(__file__)])
    jobs = []
    for src in _sources():
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), newest_header):
            jobs.append([hipcc] + FLAGS + SOURCE_FLAGS.get(src, []) + ["-I", INCLUDE, "-I", CSRC, "-c", path, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        return subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as pool:
        for res in pool.map(run, jobs):
            if res.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in _sources()]
    # Link beside the target and rename over it: the unlocked `_stale()` fast path of another rank (and `_lib.lib()`, which
    # never takes the lock) then only ever sees a COMPLETE library, and a process that has the old file mapped keeps its
    # (unlinked) inode instead of having the pages rewritten under it.
    tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    res = run(link)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    if "--diag" in sys.argv:
        print(build_diagnostic(tuple(a for a in sys.argv[1:] if a.startswith("-D")) or ("-DINR_STAMPS",), verbose=True))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
