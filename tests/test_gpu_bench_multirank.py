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


def _bench(world, extra=(), cpu_baseline=False):
    env = dict(os.environ, RLS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    base = [str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--log2-points", "20",
            *(("--cpu-seconds", "0.3") if cpu_baseline else ("--no-cpu-baseline",)), *extra]
    if world == 1:
        cmd = [sys.executable, *base]
    else:
        port = 29700 + os.getpid() % 200
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), *base]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    return _parse(p.stdout)


def _parse(stdout):
    """the compact headline is the LAST line of stdout, parses on its own and fits a 2000-character tail; the long records
    come before it, one JSON line each, and are attached here under `_records` for the assertions below"""
    lines = stdout.splitlines()
    last = lines[-1]
    assert last.startswith("{") and len(last) < 1800, (len(last), last[:200])
    assert len(stdout[-2000:].splitlines()[-1]) == len(last)          # what a 2000-character tail keeps is the whole line
    head = json.loads(last)
    assert "record" not in head and "roofline" in head and "config" in head and "ranks" in head
    recs = [json.loads(l) for l in lines[:-1] if l.startswith("{")]
    assert all("record" in r for r in recs) and [r["record"] for r in recs].count("headline_detail") == 1
    head["_records"] = recs
    head["_detail"] = [r for r in recs if r["record"] == "headline_detail"][0]
    head["workloads"] = [r for r in recs if r["record"] == "workload"]
    return head


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
    assert two["ranks"]["ranks_seen"] == 2 and two["ranks"]["distinct_devices"] == 1 and "devices" not in two["ranks"]
    rk = two["_detail"]["ranks"]
    assert two["_detail"]["value"] == two["value"] and two["_detail"]["roofline"]["kernel_ms"] == two["roofline"]["kernel_ms"]
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
    line = _parse(p.stdout)
    assert line["config"]["name"] == "skin" and line["config"]["baseline_config"] == 5 and line["workloads"] == []


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
        return _parse(p.stdout)

    port = 29900 + os.getpid() % 90
    rccl = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                "127.0.0.1", "--master-port", str(port), *base])
    plain = run([sys.executable, *base])
    assert rccl["_detail"]["config"]["control_plane"] == "torch.distributed nccl"
    assert plain["_detail"]["config"]["control_plane"] == "single process"
    assert rccl["n_gpus"] == 1
    print("RCCL-launched", rccl["value"], "plain", plain["value"])
    # two separate 2^26-point runs on a shared box: kernel time, not wall time, and a wide gate
    assert abs(rccl["roofline"]["kernel_ms"] / plain["roofline"]["kernel_ms"] - 1) < 0.08
    assert rccl["ranks"]["ranks_seen"] == 1 and rccl["ranks"]["backend"] == "nccl"


@pytest.mark.gpu
@pytest.mark.parametrize("config", [4, 5])
def test_eight_rank_dress_rehearsal(config):
    """BASELINE configs 4 and 5 are 8-GPU jobs.  Eight ranks (gloo, sharing this box's one GPU) at 2^20 points each: the
    shards tile [0, 8 * 2^20), their output checksums add up (mod 2^64) to one process's checksum over the same 2^23 points,
    and the last line is the compact headline with ranks_seen == 8 -- so the driver's 8-GPU run yields a parsed record."""
    eight = _bench(8, ("--config", str(config), "--checksum"))          # _bench passes --log2-points 20
    n = 1 << 20
    assert eight["n_gpus"] == 8 and eight["ranks"]["ranks_seen"] == 8 and eight["ranks"]["world_size"] == 8
    assert eight["config"]["baseline_config"] == config and eight["config"]["points_total"] == 8 * n
    v = eight["_detail"]["validation"]
    assert v["shards"] == [[g * n, n] for g in range(8)]
    assert len(set(v["shard_checksums"])) == 8                           # eight different shards, not one shard eight times
    assert int(v["checksum"], 16) == sum(int(c, 16) for c in v["shard_checksums"]) % (1 << 64)
    env = dict(os.environ)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", str(config), "--log2-points", "23", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--checksum"], capture_output=True, text=True, env=env,
                       timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    one = _parse(p.stdout)
    assert one["_detail"]["validation"]["shards"] == [[0, 8 * n]]
    assert one["validation"]["checksum"] == eight["validation"]["checksum"]


@pytest.mark.gpu
def test_multi_rank_line_is_self_sufficient():
    """an N > 1 line carries `roofline` AND `cpu_baseline` (rank 0 times the CPU leg after the closing
    barrier, outside the timed region, while the other ranks wait), `ranks` keeps distinct_devices, and both clocks of the
    timed region are there: the slowest rank's own K steps (ms_per_step, what `value` is quoted on) and rank 0's
    barrier-to-barrier wall (ms_per_step_wall).  Eight ranks over gloo on this box's one GPU."""
    eight = _bench(8, cpu_baseline=True)
    assert eight["n_gpus"] == 8 and eight["ranks"]["ranks_seen"] == 8
    assert eight["ranks"]["distinct_devices"] == 1 and eight["ranks"]["world_size"] == 8
    rf, cb = eight["roofline"], eight["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["kernel_ms"] > 0 and 0 < rf["frac"] < 1
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["with_alloc"]["kind"] == "port+alloc" and 0 < cb["with_alloc"]["value"] <= cb["value"] * 1.25
    assert eight["_detail"]["cpu_baseline"]["ranks_waiting"] == 7
    # rank 0's wall clock between the barriers contains the slowest rank's own time up to the skew of the barrier exits
    assert eight["ms_per_step_wall"] > 0 and eight["ms_per_step_wall"] >= 0.5 * eight["ms_per_step"]
    # rank 0 measured the clock its kernel ran at (eight ranks share the GPU here: no statement about its value)
    assert 0.5 < rf["effective_clock_ghz"] <= 2.45 and rf.get("issue_slot_frac_at_clock") is not None
    one = _bench(1, cpu_baseline=True)
    assert "ranks_waiting" not in one["_detail"]["cpu_baseline"] and one["cpu_baseline"]["with_alloc"]["value"] > 0


def _errors(stdout):
    out = []
    for l in stdout.splitlines():
        if l.startswith("{"):
            d = json.loads(l)
            if "error" in d:
                out.append(d)
    return out


@pytest.mark.gpu
def test_more_ranks_than_gpus_under_rccl_fails_fast_and_readably():
    """N = visible GPUs + 2 with backend nccl.  Bare (`python bench.py --gpus N`): the PARENT counts the
    devices before spawning and exits 2 with one JSON error line, no torchrun.  Under torchrun (how the driver starts N > 1):
    every rank leaves before the rendezvous with its own error line (one reason), the job is down in seconds instead of dying in set_device or
    waiting out a rendezvous."""
    import time
    import torch
    have = torch.cuda.device_count()
    world = have + 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("RLS_DIST_BACKEND", None)
    base = [str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--log2-points", "18", "--no-cpu-baseline"]
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, *base], capture_output=True, text=True, env=env, timeout=300, cwd=str(ROOT))
    assert p.returncode == 2 and time.perf_counter() - t0 < 60, (p.returncode, p.stderr[-2000:])
    errs = _errors(p.stdout)
    assert len(errs) == 1 and errs[0]["gpus_visible"] == have and f"--gpus {world}" in errs[0]["error"], p.stdout[-1000:]
    assert json.loads(p.stdout.splitlines()[-1]) == errs[0] and "Traceback" not in p.stderr
    port = 29100 + os.getpid() % 150
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), *base], capture_output=True, text=True, env=env, timeout=300,
                       cwd=str(ROOT))
    took = time.perf_counter() - t0
    assert p.returncode != 0 and took < 90, (p.returncode, took)
    errs = _errors(p.stdout)
    # every rank that reached the check says why (whichever exits first makes torchrun stop the rest): >= 1 line, one reason
    assert 1 <= len(errs) <= world and len({e["error"] for e in errs}) == 1, p.stdout[-1000:]
    assert errs[0]["gpus_visible"] == have and errs[0]["world_size"] == world
    assert not any(l.startswith("{") and "value" in l for l in p.stdout.splitlines())          # no result line


@pytest.mark.gpu
def test_a_rank_that_raises_takes_the_job_down_inside_a_minute():
    """rank 1 of 2 raises after the device gather while rank 0 goes on to the opening barrier.  The
    failing rank prints one JSON line {"error", "rank"} and exits non-zero without running the process group's destructors;
    torchrun stops rank 0; the job's exit code is non-zero and no result line is printed."""
    import time
    env = dict(os.environ, RLS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", RLS_BENCH_FAIL_RANK="1")
    port = 29250 + os.getpid() % 150
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup",
                        "1", "--log2-points", "20", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300,
                       cwd=str(ROOT))
    took = time.perf_counter() - t0
    assert p.returncode != 0, p.stdout[-2000:]
    assert took < 60, took
    errs = _errors(p.stdout)
    # the rank that failed says why; rank 0 may add its own line when it sees the peer's connection close (a consequence, and
    # it names itself) before torchrun stops it
    cause = [e for e in errs if e["rank"] == 1]
    assert len(cause) == 1 and cause[0]["world_size"] == 2 and "RLS_BENCH_FAIL_RANK" in cause[0]["error"], errs
    assert all(e["rank"] in (0, 1) for e in errs) and len(errs) <= 2
    assert not any(l.startswith("{") and '"value"' in l for l in p.stdout.splitlines())
    # the same through the bare launch (parent -> torchrun child -> ranks): the parent returns the children's failure
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log2-points", "20",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300, cwd=str(ROOT))
    assert p.returncode != 0 and time.perf_counter() - t0 < 60
    assert 1 in [e["rank"] for e in _errors(p.stdout)]


@pytest.mark.gpu
def test_rendezvous_gives_up_after_the_timeout():
    """a rank that never arrives: the others leave init_process_group after RLS_DIST_TIMEOUT_S (default 120 s) instead of
    torch's 10-30 minutes.  One process that believes it is rank 0 of 2, nobody else comes."""
    import time
    env = dict(os.environ, RLS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29450 + os.getpid() % 40), RANK="0",
               WORLD_SIZE="2", LOCAL_RANK="0", LOCAL_WORLD_SIZE="2", RLS_DIST_TIMEOUT_S="8")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-points", "16",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300, cwd=str(ROOT))
    took = time.perf_counter() - t0
    assert p.returncode != 0 and took < 60, (p.returncode, took)
    errs = _errors(p.stdout)
    assert len(errs) == 1 and errs[0]["rank"] == 0, p.stdout[-1500:]


@pytest.mark.gpu
def test_default_shape_with_cpu_baseline():
    """the default command's shape at small sizes: configs block (3 records, each with its own cpu_baseline) printed before
    a headline that carries roofline and cpu_baseline"""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--log2-points", "18", "--workloads", "configs",
                        "--block-log2-points", "14", "--cpu-seconds", "0.3", "--block-cpu-seconds", "0.2", "--steps", "3",
                        "--warmup", "2"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    line = _parse(p.stdout)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Gsamples/s" and cb["sample"]
    assert cb["with_alloc"]["kind"] == "port+alloc" and cb["with_alloc"]["value"] > 0
    assert line["ms_per_step_wall"] > 0 and line["roofline"]["effective_clock_ghz"] > 0.5
    assert [r["name"] for r in line["workloads"]] == ["disney_integrate", "sss_probe", "skin"]
    assert all(r["cpu_baseline"]["value"] > 0 for r in line["workloads"])
    saved = json.loads((ROOT / "gpurun_out" / "bench_workloads.json").read_text())
    assert [r["record"] for r in saved] == ["headline_detail", "workload", "workload", "workload"]
