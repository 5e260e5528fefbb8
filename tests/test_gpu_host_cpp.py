"""GPU: the C++ host mirror (rlshaders_amd/host/rls_batch.hpp) end to end -- C++ stub -> C ABI -> HIP
kernels -- against the reference-derived probe values of SURVEY.md 8(c)."""
import json
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_arnold_stub_example_matches_survey_kat():
    from rlshaders_amd import build
    build.build_library()
    exe = build.build_host_examples()
    p = subprocess.run([str(exe), "4096"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = json.loads(p.stdout.strip().splitlines()[-1])
    kat = json.loads((ROOT / "tests" / "golden" / "survey_kat.json").read_text())["ggx"]
    for a, b in zip(got["L"], kat["L"]):
        assert abs(a - b) <= 2e-9 + 1e-7 * abs(b)
    assert abs(got["f"] - kat["f"]) <= 1e-5 * kat["f"]
    assert abs(got["pdf"] - kat["pdf"]) <= 1e-5 * kat["pdf"]
    # rlSkin through the C++ mirror: layer weights are Fresnel complements in (0, 1)
    sk = got["skin"]
    assert 0.0 < sk["specularFresnel"] < 1.0 and 0.0 < sk["sheenFresnel"] < 1.0 and 0.0 < sk["sssWeight"] <= 1.0
    # SssSampler::integrateScatter through the C++ mirror: the uniformly lit plane converges to the
    # closed-form mass of the profile inside maxRadius (tests/test_oracle_scatter.py)
    import math
    dist, albedo, light = (0.05, 0.1, 0.2), (0.8, 0.5, 0.3), (2.0, 1.0, 0.5)
    rmax = 3 * max(dist)
    for d, a, e, m in zip(dist, albedo, light, got["scatter_mean"]):
        want = a * e / math.pi * (1 - (math.exp(-rmax / d) + 3 * math.exp(-rmax / (3 * d))) / 4)
        assert abs(m / want - 1) < 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("log2n,shards", [(20, 3), (22, 2), (16, 8)])
def test_multi_gpu_example_shards_add_up(log2n, shards):
    """the C++ sharded runtime: one thread + context + arena per shard (several per device on a one-GPU box); the
    shard checksums of the GGX reflect+refract outputs add up to the single-shard checksum"""
    from rlshaders_amd import build
    build.build_library()
    build.build_host_examples()
    exe = ROOT / "rlshaders_amd" / "build" / "example_multi_gpu"
    p = subprocess.run([str(exe), str(log2n), str(shards)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout, p.stderr)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    assert got["points"] == 1 << log2n and got["shards"] == shards and got["devices"] >= 1
    assert got["sharded_checksum"] == got["single_checksum"] != 0
