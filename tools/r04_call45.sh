#!/bin/bash
# end of round 4: sustained clock / power trace of the fused fit step on the final product library (>= 30 s under load)
mkdir -p gpurun_out
echo "## product library at the end of round 4 (kernel sources $(python -c 'import bench; print(bench.source_hash())'))"
bash tools/power_trace.sh gpurun_out/ptrace_final.txt python bench.py --steps 4200 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ap.log 2>&1
python tools/power_summary.py gpurun_out/ptrace_final.txt
python - <<PY
import json
for l in open("/tmp/ap.log"):
    if l.startswith("{"):
        d = json.loads(l); c = d["roofline"].get("all_gemm_launches", d["roofline"])["per_class"]
        print("ms/step %.3f  " % d["ms_per_step"] + "  ".join("%s %.3f ms" % (k[5:], v["avg_ms"]) for k, v in c.items()))
PY
