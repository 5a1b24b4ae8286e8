#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.log 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_final.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
        print("compat", d["compat_loop"]["ms_per_step"], "quality", d["quality"]["delta_db_trimmed"], d["quality"]["delta_db_median"], "cpu", d["cpu_baseline"]["value"])
PY
