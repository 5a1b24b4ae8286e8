"""-m gpu: bench.py prints ONE JSON line with the keys the driver's contract names (a short run: 2 steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--no-extras", "--cpu-sample", "2048", "--cpu-steps", "1", "--cpu-warmup", "0"], capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] == pytest.approx(524288 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    # the whole step against the peak of the operand type used, beside the dominant kernel's HBM view
    m = r["mfma"]
    assert m["algorithmic_tflops"] == pytest.approx(524288 * 5245952 / (d["ms_per_step"] * 1e-3) / 1e12, rel=1e-6)
    assert m["frac"] == pytest.approx(m["algorithmic_tflops"] / m["peak"]) and m["executed_tflops"] == pytest.approx(3 * m["algorithmic_tflops"])
    assert "gemm_hp_pkd_kernel" in r["kernel"] and r["algorithmic_bytes_per_launch"] == 3 * 524288 * 512 * 4


def test_bench_two_ranks_rehearsal_gloo():
    """The N > 1 launch exactly as the driver does it (torch.distributed.run, one rank per "GPU"), rehearsed with both ranks
    on the one test card and gloo standing in for RCCL: one line from rank 0, whole-job value over both ranks."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, INR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--eleven-steps", "6"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(2 * 524288 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)   # both ranks' rows / max time
    assert "cpu_baseline" not in d        # rank 0 times the CPU baseline at N = 1 only
    # north_star's multi-GPU figure: the 11 committed volumes through run_volumes over both ranks (one gang + ten whole fits)
    e = d["eleven_patients"]
    assert e["steps"] == 6 and e["seconds"] > 0 and len(e["per_rank_busy_s"]) == 2 and all(b > 0 for b in e["per_rank_busy_s"])
    assert len(e["plan"]["gangs"]) == 1 and e["plan"]["gangs"][0][1] == [0, 1]
    assert sorted([e["plan"]["gangs"][0][0]] + [j for w in e["plan"]["whole"] for j in w]) == list(range(11))
    assert e["coordinate_steps_per_s"] == pytest.approx(sum(64 * 64 * z for z in [24] * 3 + [28] * 5 + [34] * 3) * 6 / e["seconds"], rel=1e-6)
    assert 15.0 < e["psnr_db_mean"] < 45.0 and e["final_loss_max"] < 0.2


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (a driver that does not wrap it in torch.distributed.run):
    bench.py starts the ranks itself as a child job -- before anything has touched the GPU -- and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(INR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras",
                          "--no-cpu-baseline", "--no-eleven"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] == pytest.approx(2 * 524288 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)
