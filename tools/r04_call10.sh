#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash tools/prof_all.sh > gpurun_out/r4_prof_all.log 2>&1; echo "prof_all rc=$?"; tail -4 gpurun_out/r4_prof_all.log
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.log 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_final.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
