"""GPU: bench.py's multi-rank path end to end on one GPU -- torchrun, index-range shards, barrier,
max-over-ranks, one JSON line from rank 0.  Two ranks share the GPU and talk over gloo (RCCL refuses
duplicate devices); on the 8-GPU node the same code runs with one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bench(world, extra=()):
    env = dict(os.environ, RLS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    base = [str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--log2-points", "20",
            "--no-cpu-baseline", *extra]
    if world == 1:
        cmd = [sys.executable, *base]
    else:
        port = 29700 + os.getpid() % 200
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), *base]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_one_json_line():
    one = _bench(1)
    two = _bench(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for r in (one, two):
        assert r["metric"].startswith("BSDF Gsamples/sec") and r["unit"] == "Gsamples/s"
        assert r["scaling"] == "weak" and r["dtype"] == "f32" and r["vs_baseline"] is None
        assert r["config"]["points_per_gpu"] == 1 << 20 and r["steps"] == 3 and r["warmup"] == 1
        assert r["value"] > 0 and r["ms_per_step"] > 0
        rf = r["roofline"]
        assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
        assert rf["algorithmic_bytes_per_launch"] == 124 * (1 << 20)
    # value counts the samples of ALL ranks: samples = world * n * 2 * steps over the slowest rank's time
    assert abs(two["value"] - 2 * (1 << 20) * 2 * 3 / (two["ms_per_step"] * 3e-3) / 1e9) / two["value"] < 0.02
    assert "x2" in two["config"]["sharding"]


@pytest.mark.gpu
def test_rccl_control_path_on_one_gpu():
    """`torchrun --nproc-per-node 1 bench.py --gpus 1` with backend nccl (= RCCL): init_process_group("nccl", device_id),
    barrier and the device all_reduce(MAX) of the timing run for real; the value agrees with the plain one-process line"""
    base = [str(ROOT / "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "10", "--log2-points", "26",
            "--no-cpu-baseline", "--arena-candidates", "2"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("RLS_DIST_BACKEND", None)

    def run(cmd):
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(ROOT))
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout
        return json.loads(lines[0])

    port = 29900 + os.getpid() % 90
    rccl = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                "127.0.0.1", "--master-port", str(port), *base])
    plain = run([sys.executable, *base])
    assert rccl["config"]["control_plane"] == "torch.distributed nccl" and plain["config"]["control_plane"] == "single process"
    assert rccl["n_gpus"] == 1
    print("RCCL-launched", rccl["value"], "plain", plain["value"])
    assert abs(rccl["value"] / plain["value"] - 1) < 0.03
