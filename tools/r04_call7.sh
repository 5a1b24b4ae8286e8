#!/bin/bash
# round 4, final measurements: kernel trace + PMC passes of the short bench, step-time table, RAMS traces, the full bench line
set -o pipefail
mkdir -p gpurun_out
bash tools/prof_all.sh > gpurun_out/r4_prof_all.log 2>&1; echo "prof_all rc=$?"; tail -4 gpurun_out/r4_prof_all.log
timeout -k 10 400 python tools/step_time_table.py gpurun_out/r04_step_time_table.json > gpurun_out/r4_step_table.log 2>&1; tail -3 gpurun_out/r4_step_table.log
bash tools/prof_rams.sh > gpurun_out/r4_prof_rams.log 2>&1; tail -2 gpurun_out/r4_prof_rams.log
bash tools/prof_rams_train.sh > gpurun_out/r4_prof_rams_train.log 2>&1; tail -3 gpurun_out/r4_prof_rams_train.log
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.log 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04_bench_final.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
        print("compat", d.get("compat_loop"))
        print("rams", {k: v for k, v in d.get("rams", {}).items() if isinstance(v, dict)})
        print("quality", {k: d["quality"][k] for k in ("psnr_db_mean", "delta_db", "se_db", "train_voxels_per_s")})
        print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu_baseline"))
PY
