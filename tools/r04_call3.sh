#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t3.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r4_t3.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r4_bench3.json 2> gpurun_out/r4_bench3.err; python - <<'PY'
import json
for l in open("gpurun_out/r4_bench3.json"):
    if l.startswith("{"):
        d=json.loads(l); print("ms/step", d["ms_per_step"], {k[5:]:round(v["avg_ms"],4) for k,v in d["roofline"]["all_gemm_launches"]["per_class"].items()}, "fp32", d.get("fp32_mfma",{}).get("ms_per_step"))
PY
