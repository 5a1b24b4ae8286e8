#!/bin/bash
# GPU box: sustained clock / power traces of the fused fit step on the product library and on the diagnostic builds of
# gemm_hp.inc (-DHP_ABLATE: 4 = K-loop only, no epilogue work; 11 = epilogue only: no LDS-DMA, no MFMA, no fragment reads;
# 2 = everything but the MFMAs).  Every build runs >= 30 s under load, rocm-smi sampled ~2x a second (tools/power_trace.sh);
# round 3 had ONE in-load sample per ablated build (verdict r03, weak 4).
#   build first (CPU container):  python mri-super-resolution_amd/_build.py --diag -DHP_ABLATE=4   (-> libinrhip_abl4.so), 11, 2
#   bash tools/ablate_power.sh > gpurun_out/r04_ablate_power.txt
ROOT=$(pwd)
mkdir -p gpurun_out
for a in "" 4 11 2; do
  if [ -z "$a" ]; then unset INR_LIB; tag="product library (K-loop + epilogue)"; steps=4200; else export INR_LIB=$ROOT/mri-super-resolution_amd/libinrhip_abl$a.so; tag="-DHP_ABLATE=$a"; steps=7000; fi
  [ -n "$a" ] && [ ! -f "$INR_LIB" ] && { echo "## $tag: library not built"; continue; }
  echo "## $tag"
  bash tools/power_trace.sh gpurun_out/ptrace_$a.txt python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-extras > /tmp/ap.log 2>&1
  python tools/power_summary.py gpurun_out/ptrace_$a.txt
  python - <<PY
import json
for l in open("/tmp/ap.log"):
    if l.startswith("{"):
        d = json.loads(l); c = d["roofline"].get("all_gemm_launches", d["roofline"])["per_class"]
        print("ms/step %.3f  " % d["ms_per_step"] + "  ".join("%s %.3f ms" % (k[5:], v["avg_ms"]) for k, v in c.items()))
PY
done
