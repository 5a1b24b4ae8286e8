#!/usr/bin/env python3
"""Measured fit-step time t(N) of Siren(256,512,3,1) on ONE GPU at the row counts the multi-GPU plan deals with:
whole volumes (z = 24 / 28 / 34: 98,304 / 114,688 / 139,264 rows), the row shards the 8-rank plan creates (a 34-slice volume
over 3 or 2 ranks: 46,421 / 46,422 / 69,632 rows), small fits (4,096 / 16,384) and the synthetic 128^3 (524,288).
`fused` = SirenFitter.step (inr_siren_fit: the whole-volume path), `sharded` = ShardedSirenFitter.step with a world of one
(inr_siren_loss_grad + inr_adam_step + loss copy per step, no all-reduce: what a gang member runs between collectives).
Writes JSON (default profiles/r05_step_time_table.json); mri-super-resolution_amd/dist.py carries a copy as its default
cost model.   python tools/step_time_table.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import ops  # noqa: E402

ROWS = (4096, 16384, 32768, 46421, 46422, 65536, 69632, 98304, 114688, 139264, 262144, 524288)


def measure(fitter, x, t, steps, reps=3):
    fitter.step(x, t, 3)
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fitter.step(x, t, steps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_step_time_table.json")
    table = []
    for n in ROWS:
        g = torch.Generator(device="cuda").manual_seed(n)
        x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous()
        t = torch.rand(n, device="cuda", generator=g)
        steps = max(20, min(400, int(2.0e7 / n)))
        torch.manual_seed(0)
        fused = measure(inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4), x, t, steps)
        torch.manual_seed(0)
        sharded = measure(inr.ShardedSirenFitter(inr.Siren(256, 512, 3, 1).cuda(), global_rows=2 * n, lr=1e-4), x, t, steps)
        row = {"rows": n, "steps_timed": steps, "fused_ms_per_step": fused, "sharded_ms_per_step": sharded,
               "fused_rows_per_s": n / fused * 1e3, "sharded_rows_per_s": n / sharded * 1e3}
        table.append(row)
        print(json.dumps(row), flush=True)
    ref = table[-1]["fused_rows_per_s"]
    for r in table:
        r["fused_rate_vs_128cube"] = r["fused_rows_per_s"] / ref
        r["sharded_rate_vs_128cube"] = r["sharded_rows_per_s"] / ref
    caps = ops.device_caps(0)
    with open(out_path, "w") as fh:
        json.dump({"network": "Siren(256,512,3,1), Adam, full-batch MSE", "device": caps["arch"], "compute_units": caps["compute_units"],
                   "note": "best of 3 timed blocks per row count, host clock around stream-synchronised blocks of fused steps",
                   "table": table}, fh, indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
