#!/usr/bin/env python3
"""Condense gpurun_out/prof/* (rocprofv3 kernel trace + PMC passes of bench.py) into the committed profiles/ files."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
prof = os.path.join(root, "gpurun_out", "prof")
out_dir = os.path.join(root, "profiles")
def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


shutil.copy(newest(prof + "/kt/**/*_kernel_stats.csv"), os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
with open(os.path.join(out_dir, f"{tag}_bench_under_rocprof.json"), "w") as fh:
    fh.write("".join(l for l in open(prof + "/kt.log") if l.startswith("{")))


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "inr::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in agg.items()}


import hashlib, subprocess


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(root, "mri-super-resolution_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(root, "mri-super-resolution_amd", "_build.py"), "rb") as fh:      # the compiler flags are part of the code
        h.update(b"_build.py\0" + fh.read())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None


hbm = {"source_hash": source_hash(), "git_head": git_head(),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --steps 3 --warmup 1 --no-cpu-baseline "
               "--no-extras`; per-dispatch averages in bytes (counter unit = KiB).  FETCH_SIZE is doubled per "
               "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced 16-B/lane reads); WRITE_SIZE is exact.",
       "kernels": {}}
for name, sub, mult in (("fetch", "pmc_fetch", 2.0), ("write", "pmc_write", 1.0)):
    for k, d in per_kernel(newest(f"{prof}/{sub}/**/*counter_collection.csv")).items():
        (avg, n), = d.values()
        e = hbm["kernels"].setdefault(k, {})
        e[name + "_raw_bytes"], e[name + "_bytes"], e["dispatches"] = avg * 1024, avg * 1024 * mult, n
for fam in ("gemm_f32", "gemm_h3", "gemm_hp"):
    tot_b = sum((d.get("fetch_bytes", 0) + d.get("write_bytes", 0)) * d["dispatches"] for k, d in hbm["kernels"].items() if fam in k)
    tot_n = sum(d["dispatches"] for k, d in hbm["kernels"].items() if fam in k)
    if tot_n:
        hbm[fam + "_avg_hbm_bytes_per_launch"] = tot_b / tot_n
# the GEMM family that carried the timed steps of this run (bench.py reads this key): the profiled command also runs the
# exact-fp32 leg, whose gemm_f32 kernels must not be averaged in
for fam in ("gemm_hp", "gemm_h3", "gemm_f32"):
    if fam + "_avg_hbm_bytes_per_launch" in hbm:
        hbm["gemm_avg_hbm_bytes_per_launch"] = hbm[fam + "_avg_hbm_bytes_per_launch"]
        hbm["gemm_family"] = fam
        break
# per GEMM class of the pre-split path (template arguments: gemm_hp_pkd / pkc / nt kernels <EPI, ..>, EPI 2 = HPE_MUL = input-grad,
# 0 / 1 / 4 = forward; gemm_hp_kernel<1, 3> = row-contraction parameter gradient)
import re
by_class = {}
for k, d in hbm["kernels"].items():
    m = re.search(r"gemm_hp_(pkd|pkc|nt)_kernel<(\d+)", k) or re.search(r"gemm_hp_kernel<(\d+), (\d+)", k)
    if not m:
        continue
    if "gemm_hp_kernel<" in k:
        cls = "gemm_param_grad" if m.group(1) == "1" else ("gemm_input_grad" if m.group(2) == "2" else "gemm_forward")
    else:
        cls = "gemm_input_grad" if m.group(2) == "2" else "gemm_forward"
    e = by_class.setdefault(cls, [0.0, 0])
    e[0] += (d.get("fetch_bytes", 0) + d.get("write_bytes", 0)) * d["dispatches"]
    e[1] += d["dispatches"]
hbm["hbm_bytes_per_launch_by_class"] = {c: b / n for c, (b, n) in by_class.items() if n}
json.dump(hbm, open(os.path.join(out_dir, f"{tag}_pmc_hbm.json"), "w"), indent=1)
sq = {"note": "rocprofv3 --pmc (SQ/GRBM pass) on the same command; per-dispatch averages.  SQ_WAVE/WAIT/ACTIVE count quad-cycles, "
              "SQ_VALU_MFMA_BUSY_CYCLES = MFMA pipe cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs.",
      "kernels": {k: {c: v[0] for c, v in d.items()} for k, d in per_kernel(newest(prof + "/pmc_sq/**/*counter_collection.csv")).items()}}
for k, d in sq["kernels"].items():
    if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["GRBM_GUI_ACTIVE"] > 0:
        d["mfma_pipe_utilisation"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8)
json.dump(sq, open(os.path.join(out_dir, f"{tag}_pmc_sq.json"), "w"), indent=1)
for key in ("gemm_f32_avg_hbm_bytes_per_launch", "gemm_h3_avg_hbm_bytes_per_launch", "gemm_hp_avg_hbm_bytes_per_launch",
            "gemm_avg_hbm_bytes_per_launch"):
    if key in hbm:
        print("%s: %.0f MB" % (key, hbm[key] / 1e6))
for k, d in sq["kernels"].items():
    if "gemm" in k:
        print(k[:70], "MFMA util %.3f  bank conflicts %.3g" % (d.get("mfma_pipe_utilisation", 0), d.get("SQ_LDS_BANK_CONFLICT", 0)))

# the full bench line of the round and the RAMS forward summaries, when present
final = os.path.join(root, "gpurun_out", f"{tag}_bench_final.log")
if os.path.exists(final):
    lines = [l for l in open(final) if l.startswith("{")]
    if lines:
        with open(os.path.join(out_dir, f"{tag}_bench.json"), "w") as fh:
            fh.write(lines[-1])
for b in (25, 1):
    src = os.path.join(prof, f"rams_b{b}_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(out_dir, f"{tag}_rams_b{b}_kernel_stats.csv"))
