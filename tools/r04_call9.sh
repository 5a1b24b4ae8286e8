#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4_smoke.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t9.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4_t9.log
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.log 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_final.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
        print("compat", d["compat_loop"]["ms_per_step"], d["compat_loop"]["vs_fused_step"])
        q = d["quality"]; print("quality", q["delta_db"], q["se_db"], q["delta_db_trimmed"], q["se_db_trimmed"], q["delta_db_median"], q["seeds_ending_on_a_spike"])
        print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"])
PY
