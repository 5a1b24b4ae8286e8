"""CPU baseline exactly as BASELINE.md section 3 states it, once, off the driver line: the torch-CPU port of the reference loop
(oracle/torch_port) on ALL 524,288 LR rows of the synthetic 128^3 fit, 5 warm-up + 20 timed full-batch steps at the best thread
count of a sweep, with the host's `lscpu` model beside it.  ~4-6 minutes on the GPU box's host.
  python tools/cpu_baseline_full.py > gpurun_out/r04_cpu_baseline.json        (then copied to profiles/)
bench.py keeps its short form (1 + 3 steps) and cites this file."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mri_super_resolution_amd import drivers  # noqa: E402


def main():
    ncpu = os.cpu_count() or 1
    sweep = sorted({max(1, ncpu // 8), max(1, ncpu // 4), max(1, ncpu // 2), ncpu}) if ncpu >= 16 else [ncpu]
    vol = bench.synthetic_volume(bench.SIDE, seed=0)
    B_np = drivers.fourier_matrix(3, seed=0)
    steps, warmup = int(os.environ.get("CPU_STEPS", 20)), int(os.environ.get("CPU_WARMUP", 5))
    rec = bench.cpu_baseline(65536, steps, warmup, B_np, vol, sweep)
    rec["sample"] = (f"all 524288 LR rows of the synthetic 128^3 fit, {steps} timed full-batch steps after {warmup} warm-up at "
                     f"{rec['cores']} threads (BASELINE.md section 3 as stated), thread count picked by a 2-step sweep on 65536 rows")
    try:
        lscpu = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=20).stdout
        keep = ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "CPU(s):", "CPU max MHz")
        rec["lscpu"] = {l.split(":")[0].strip(): l.split(":", 1)[1].strip() for l in lscpu.splitlines()
                        if any(l.startswith(k) for k in keep)}
    except Exception as e:  # noqa: BLE001
        rec["lscpu"] = {"error": str(e)[:100]}
    rec["nproc"] = ncpu
    rec["torch"] = torch.__version__
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
