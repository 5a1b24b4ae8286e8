#!/bin/bash
# round 4, GPU call 1: RAMS regression tests, K-loop probe under power trace, sustained ablation traces, full CPU baseline
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_rams.py -x -q > gpurun_out/r4_t1.log 2>&1; echo "rams tests rc=$?" 
tail -3 gpurun_out/r4_t1.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/kloop_probe tools/kloop_probe.hip 2> gpurun_out/r4_probe_build.log || exit 1
timeout -k 10 300 bash tools/power_trace.sh gpurun_out/r4_probe_trace.txt /tmp/kloop_probe --secs 12 --rounds 2 --dma 1 > gpurun_out/r4_probe_dma1.txt 2>&1 || { echo probe failed; tail -5 gpurun_out/r4_probe_dma1.txt; exit 1; }
python tools/power_summary.py gpurun_out/r4_probe_trace.txt gpurun_out/r4_probe_dma1.txt > gpurun_out/r4_probe_dma1_power.txt
cat gpurun_out/r4_probe_dma1.txt gpurun_out/r4_probe_dma1_power.txt
timeout -k 10 200 bash tools/power_trace.sh gpurun_out/r4_probe_trace0.txt /tmp/kloop_probe --secs 10 --rounds 1 --dma 0 > gpurun_out/r4_probe_dma0.txt 2>&1 && python tools/power_summary.py gpurun_out/r4_probe_trace0.txt gpurun_out/r4_probe_dma0.txt > gpurun_out/r4_probe_dma0_power.txt
cat gpurun_out/r4_probe_dma0.txt gpurun_out/r4_probe_dma0_power.txt
timeout -k 10 600 bash tools/ablate_power.sh > gpurun_out/r04_ablate_power.txt 2>&1
cat gpurun_out/r04_ablate_power.txt
timeout -k 10 600 python tools/cpu_baseline_full.py > gpurun_out/r04_cpu_baseline.json 2> gpurun_out/r04_cpu_baseline.err
tail -30 gpurun_out/r04_cpu_baseline.json
