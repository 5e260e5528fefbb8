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
    # the line says who took part: both ranks seen (gathered over the process group), each with its own kernel time
    rk = two["ranks"]
    assert rk["ranks_seen"] == 2 and rk["world_size"] == 2 and [d["rank"] for d in rk["devices"]] == [0, 1]
    assert len({d["pid"] for d in rk["devices"]}) == 2 and len(rk["per_rank_kernel_ms"]) == 2
    assert rk["min_kernel_ms"] <= rk["max_kernel_ms"] and rk["max_kernel_ms"] == max(rk["per_rank_kernel_ms"])
    assert rk["distinct_devices"] == 1                     # this test shares one GPU (gloo); RCCL would have refused
    assert one["ranks"]["ranks_seen"] == 1


@pytest.mark.gpu
def test_workloads_block_and_config_presets():
    """the `workloads` block (BASELINE configs 3, 4, 5) at two ranks, and the --config presets"""
    two = _bench(2, ("--workloads", "configs", "--block-log2-points", "16"))
    recs = two["workloads"]
    assert [r["name"] for r in recs] == ["disney_integrate", "sss_probe", "skin"]
    assert [r["baseline_config"] for r in recs] == [3, 4, 5]
    for r in recs:
        assert r["points_per_gpu"] == 1 << 16 and r["points_total"] == 2 << 16 and r["value"] > 0
        assert r["warmup"] >= 10 and len(r["per_rank_kernel_ms"]) == 2
        rf = r["roofline"]
        assert rf["kernel_ms"] > 0 and rf["algorithmic_bytes_per_point"] > 0
    # config 4: both byte counts are printed -- 92 B the kernel moves, 104 B by SURVEY's definition
    probe = recs[1]["roofline"]
    assert probe["algorithmic_bytes_per_point"] == 92 and probe["survey_bytes_per_point"] == 104
    assert probe["frac_survey_bytes"] > probe["frac"]
    env = dict(os.environ)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "5", "--log2-points", "18", "--steps", "3",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["name"] == "skin" and line["config"]["baseline_config"] == 5 and "workloads" not in line


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
    # two separate 2^26-point runs on a shared box: kernel time, not wall time, and a wide gate
    assert abs(rccl["roofline"]["kernel_ms"] / plain["roofline"]["kernel_ms"] - 1) < 0.08
    assert rccl["ranks"]["ranks_seen"] == 1 and rccl["ranks"]["backend"] == "nccl"
