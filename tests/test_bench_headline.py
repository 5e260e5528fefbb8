"""CPU: bench.py's last stdout line stays small enough for a reader that keeps a 2000-character tail, whatever the rank
count, device names and CPU model strings are -- and carries `roofline` and `cpu_baseline`.  (Round 3's 37 KB line left the
driver's record unparsed.)  The real runs are covered on the GPU by tests/test_gpu_bench_multirank.py."""
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def _worst_case_detail(world=8):
    dev = {"index": 7, "name": "AMD Instinct MI355X " * 3, "uuid": "GPU-" + "f" * 32, "pci_bus_id": "0000:ff:00.0", "rank": 7,
           "host": "h" * 60, "pid": 1234567}
    return {
        "metric": "BSDF Gsamples/sec (eval+sample+pdf)", "value": 12345.6789, "unit": "Gsamples/s", "n_gpus": world, "steps": 40,
        "warmup": 10, "ms_per_step": 123.45678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "w" * 400, "name": "ggx_reflect_refract_host_materials", "baseline_config": 2, "math": "exact",
                   "points_per_gpu": 1 << 27, "points_total": world << 27, "samples_per_point": 144,
                   "sharding": f"index-range x{world}, no collective", "control_plane": "torch.distributed nccl",
                   "placement": {"probe": list(range(64))}, "warmup_note": "n" * 300},
        "roofline": {"bound": "hbm", "achieved": 4242.42, "peak": 8000.0, "unit": "GB/s", "frac": 0.5303, "traffic": 8321499136,
                     "kernel": "ggx_kernel<5, 0, 1>", "kernel_ms": 1.97031, "algorithmic_bytes_per_launch": 8321499136,
                     "issue_slot_frac": 0.6012, "frac_note": "x" * 900, "counter_source": {"a": "b" * 200}},
        "ranks": {"ranks_seen": world, "world_size": world, "backend": "nccl", "devices": [dev] * world,
                  "distinct_devices": world, "per_rank_kernel_ms": [1.97] * world, "min_kernel_ms": 1.96, "max_kernel_ms": 1.98},
        "other_math_mode": {"math": "fast", "kernel_ms": 1.31, "value": 102.6, "hbm_frac": 0.8, "parity": "p" * 500},
        "validation": {"shards": [[0, 1]] * world, "shard_checksums": ["0" * 16] * world, "checksum": "f" * 16, "what": "x"},
        "cpu_baseline": {"value": 0.345678, "unit": "Gsamples/s", "cores": 256, "kind": "port", "per_core_msamples": 5.4,
                         "cpu_model": "AMD EPYC 9575F 64-Core Processor " * 4, "mean_value": 0.33, "note": "n" * 200,
                         "sample": "s" * 400, "sample_short": "t" * 300},
    }


def test_headline_fits_a_2000_character_tail():
    b = _bench()
    for world in (1, 8, 64):
        text = json.dumps(b.headline(_worst_case_detail(world)))
        assert len(text) < b.HEADLINE_MAX_BYTES <= 1800
        line = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, k
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
        assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
        assert "workload" in line["config"] and "model" not in line["config"]
        assert "devices" not in line["ranks"] and line["ranks"]["ranks_seen"] == world


def test_headline_without_the_optional_objects():
    """N > 1 (no cpu_baseline), --checksum (no other_math_mode), a VALU-bound headline without a committed instruction mix"""
    b = _bench()
    d = _worst_case_detail(8)
    for k in ("cpu_baseline", "other_math_mode"):
        d.pop(k)
    d["roofline"].update(bound="valu", achieved=None, frac=None, traffic=None, issue_slot_frac=None)
    text = json.dumps(b.headline(d))
    line = json.loads(text)
    assert len(text) < 1800 and "cpu_baseline" not in line and line["validation"]["checksum"] == "f" * 16
    assert line["roofline"]["bound"] == "valu" and line["roofline"]["frac"] is None


def test_headline_survives_unbounded_strings():
    """every free-text field grown past any sensible size (kernel name, sharding, metric, unit, cpu model): the
    line still fits, still carries the contract's keys with roofline and cpu_baseline numbers -- optional objects go first --
    and nothing on the output path raises."""
    b = _bench()
    d = _worst_case_detail(64)
    d["metric"] = "m" * 3000
    d["unit"] = "u" * 500
    d["data"] = "d" * 500
    d["config"]["sharding"] = "s" * 3000
    d["config"]["name"] = "ggx_reflect_refract"
    d["roofline"]["kernel"] = "k" * 3000
    d["roofline"].update(effective_clock_ghz=1.9, issue_slot_frac_at_clock=0.75, valu_busy_frac=0.97)
    d["cpu_baseline"]["cpu_model"] = "c" * 3000
    d["cpu_baseline"]["with_alloc"] = {"value": 0.3, "kind": "port+alloc", "what": "w" * 1000}
    d["ms_per_step_wall"] = 2.0
    text = json.dumps(b.headline(d))
    assert len(text) < b.HEADLINE_MAX_BYTES
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == d["value"] and line["roofline"]["frac"] == d["roofline"]["frac"]
    assert line["roofline"]["effective_clock_ghz"] == 1.9 and line["roofline"]["valu_busy_frac"] == 0.97
    assert line["cpu_baseline"]["value"] == d["cpu_baseline"]["value"] and line["cpu_baseline"]["with_alloc"]["value"] == 0.3


def test_emit_prints_the_headline_last(tmp_path, capsys):
    b = _bench()
    detail = _worst_case_detail(8)
    records = [{"record": "workload", "name": f"w{i}", "roofline": {"frac_note": "y" * 2000}} for i in range(22)]
    out = tmp_path / "sub" / "records.json"
    b.emit(detail, records, str(out))
    lines = capsys.readouterr().out.splitlines()
    assert len(lines) == 1 + 22 + 1
    last = json.loads(lines[-1])
    assert "record" not in last and last["value"] == detail["value"] and len(lines[-1]) < 1800
    assert ("\n".join(lines))[-2000:].splitlines()[-1] == lines[-1]
    assert [json.loads(l)["record"] for l in lines[:-1]] == ["headline_detail"] + ["workload"] * 22
    saved = json.loads(out.read_text())
    assert len(saved) == 23 and saved[0]["record"] == "headline_detail" and saved[0]["ranks"]["devices"]


def test_default_block_is_the_three_configurations():
    b = _bench()
    assert [(w, l) for w, l, _ in b.BLOCK_CONFIGS] == [("disney_integrate", 26), ("sss_probe", 25), ("skin", 27)]
    sys_argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = b.parse_args()
        assert a.workloads == "configs" and a.gpus == 1 and a.workload == "ggx_reflect_refract" and a.log2_points == 26
        sys.argv = ["bench.py", "--gpus", "8"]
        assert b.parse_args().workloads == "configs"
        sys.argv = ["bench.py", "--config", "5"]
        a = b.parse_args()
        assert a.workloads == "none" and a.workload == "skin" and a.log2_points == 27
    finally:
        sys.argv = sys_argv


# ---- failure shape: a launch that cannot work says so in ONE line, fast, with a non-zero exit ----------
def _run_bench(argv, env_extra=None, timeout=120):
    import os
    import subprocess
    import time
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=timeout,
                       cwd=str(ROOT))
    return p, time.perf_counter() - t0


def _error_lines(stdout):
    out = []
    for l in stdout.splitlines():
        if l.startswith("{"):
            d = json.loads(l)
            if "error" in d:
                out.append(d)
    return out


def test_more_ranks_than_gpus_fails_in_the_parent_before_spawning():
    """`python bench.py --gpus 8` on a box with fewer GPUs (here: none, unless the suite runs on a GPU box): the parent
    counts the devices BEFORE it spawns torchrun, prints one JSON error line and exits non-zero -- no child ever dies inside
    torch.cuda.set_device, nothing waits for a rendezvous"""
    import torch
    have = torch.cuda.device_count()
    p, took = _run_bench(["--gpus", str(have + 7), "--steps", "2", "--warmup", "1"])
    assert p.returncode == 2, (p.returncode, p.stderr[-2000:])
    errs = _error_lines(p.stdout)
    assert len(errs) == 1 and errs[0]["world_size"] == have + 7 and errs[0]["bench"] == "bench.py", p.stdout[-2000:]
    assert ("GPU(s) visible" in errs[0]["error"]) if have else ("no GPU visible" in errs[0]["error"])
    assert "torch.distributed.run" not in p.stderr and "Traceback" not in p.stderr       # nothing was launched, nothing raised
    assert json.loads(p.stdout.splitlines()[-1]) == errs[0]                              # the LAST line is the error line
    assert took < 60


def test_one_rank_without_a_gpu_says_so_in_one_line():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p, took = _run_bench(["--steps", "2", "--warmup", "1"])
    assert p.returncode == 2 and took < 60
    errs = _error_lines(p.stdout)
    assert len(errs) == 1 and "no GPU visible" in errs[0]["error"] and errs[0]["rank"] == 0 and "Traceback" not in p.stderr


def test_torchrun_with_more_ranks_than_gpus_leaves_before_the_rendezvous():
    """the driver starts N > 1 itself (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`): every rank
    sees LOCAL_WORLD_SIZE > device count and leaves at once, before init_process_group -- the job is down in seconds with an
    error line, not after a rendezvous timeout"""
    import os
    import subprocess
    import time
    import torch
    have = torch.cuda.device_count()
    world = have + 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("RLS_DIST_BACKEND", None)
    port = 29300 + os.getpid() % 150
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300, cwd=str(ROOT))
    took = time.perf_counter() - t0
    assert p.returncode != 0 and took < 90, (p.returncode, took)
    errs = _error_lines(p.stdout)
    # every rank that got as far as the check says why (whichever exits first makes torchrun stop the rest): at least one line,
    # all with the same reason
    assert 1 <= len(errs) <= world and len({e["error"] for e in errs}) == 1, p.stdout[-2000:]
    assert errs[0]["world_size"] == world and ("GPU(s) visible" in errs[0]["error"] or "no GPU visible" in errs[0]["error"])


def test_a_rank_that_hangs_gives_up_at_the_deadline():
    """a rank stuck where no collective's timeout can see it leaves after RLS_BENCH_DEADLINE_S (default 1200 s; the default run
    takes ~30 s) with the one-line error and a non-zero exit, instead of burning the caller's half hour"""
    p, took = _run_bench(["--steps", "2", "--warmup", "1"], {"RLS_BENCH_DEADLINE_S": "2", "RLS_BENCH_STALL_S": "60"})
    assert p.returncode == 1 and took < 30, (p.returncode, took)
    errs = _error_lines(p.stdout)
    assert len(errs) == 1 and "deadline" in errs[0]["error"] and errs[0]["rank"] == 0
