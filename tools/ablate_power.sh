#!/bin/bash
# GPU box: does each HALF of the dominant GEMMs already sit at the package power cap?  Runs the fused fit steps on the product
# library and on two diagnostic builds of gemm_hp.inc (-DHP_ABLATE: 4 = K-loop only, no epilogue work; 11 = epilogue only: no
# LDS-DMA, no MFMA, no fragment reads; 2 = everything but the MFMAs) and samples rocm-smi (shader clock, package power) once a second meanwhile.
#   build first (CPU container):  python mri-super-resolution_amd/_build.py --diag -DHP_ABLATE=4   (-> libinrhip_abl4.so), same for 11
#   bash tools/ablate_power.sh > gpurun_out/r03_ablate_power.txt
ROOT=$(pwd)
for a in "" 4 11 2; do
  if [ -z "$a" ]; then unset INR_LIB; tag="product library (K-loop + epilogue)"; else export INR_LIB=$ROOT/mri-super-resolution_amd/libinrhip_abl$a.so; tag="-DHP_ABLATE=$a"; fi
  [ -n "$a" ] && [ ! -f "$INR_LIB" ] && { echo "## $tag: library not built"; continue; }
  echo "## $tag"
  python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ap.log 2>&1 &
  BP=$!
  sleep 9
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 1
  done
  wait $BP
  python - <<PY
import json
for l in open("/tmp/ap.log"):
    if l.startswith("{"):
        d = json.loads(l); c = d["roofline"].get("all_gemm_launches", d["roofline"])["per_class"]
        print("ms/step %.3f  " % d["ms_per_step"] + "  ".join("%s %.3f ms" % (k[5:], v["avg_ms"]) for k, v in c.items()))
PY
done
