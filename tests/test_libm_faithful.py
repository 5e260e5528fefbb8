"""CPU: the device elementary functions (rlshaders_amd/csrc/rls_libm.hpp), compiled for the host, against the host
libm, bit for bit.  This is what makes the sampled directions of the HIP kernels identical to the CPU closures'
instead of merely close (the visible-normal slope equations amplify a 1-ulp difference in atan2f/acosf/tanf past
1e-5 on ~0.5 % of points)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


import sys

sys.path.insert(0, str(ROOT / "tests"))
import cases  # noqa: E402


def host_libm_uses_fma() -> bool:
    """glibc's ifunc resolvers pick the -mfma builds of sinf/cosf/expf/logf/powf when the CPU has AVX2 and FMA -- decided
    here from the functions' RESULTS on the 42 arguments that tell the two builds apart (cases.host_libm_flavour)"""
    return cases.host_libm_flavour()["flavour"] != "sse2"


def test_host_libm_flavour_probe():
    """the probe behind the strict parity gates: the committed discriminating arguments (found by sweeping all 2^32
    through both flavours of rls_libm.hpp) really discriminate, and this host's libm sides with exactly one flavour"""
    r = cases.host_libm_flavour()
    assert r["discriminating"] == 42 and {k: v["discriminating"] for k, v in r["functions"].items()} == \
        {"sinf": 12, "cosf": 22, "expf": 2, "powf5": 6}
    assert r["flavour"] in ("fma", "sse2"), r
    fma = sum(v["libm_sides_with_fma"] for v in r["functions"].values())
    sse = sum(v["libm_sides_with_sse2"] for v in r["functions"].values())
    assert {fma, sse} == {42, 0}, r
    if r["flavour"] == "fma":
        assert all(v["libm_differs_from_fma_port"] == 0 for v in r["functions"].values()), r
        assert cases.strict_parity()


def test_device_libm_matches_host_libm(tmp_path):
    exe = tmp_path / "libm_faithful"
    fma = host_libm_uses_fma()
    # the header follows the FMA build of glibc by default (RLM_GLIBC_FMA = 1); on a host whose glibc runs the SSE2
    # build the other flavour is the one that has to match
    subprocess.run(["g++", "-O2", "-std=gnu++17", "-ffp-contract=off", f"-DRLM_GLIBC_FMA={1 if fma else 0}",
                    f"-I{ROOT / 'rlshaders_amd' / 'csrc'}",
                    str(ROOT / "tests" / "native" / "libm_faithful.cpp"), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe), "2000000"], capture_output=True, text=True, check=True).stdout
    rows = {l.split()[0]: tuple(int(x) for x in l.split()[1:]) for l in out.strip().splitlines()}
    assert len(rows) == 15, out
    for name, (bad, total, maxulp) in rows.items():
        assert total >= 4_000_000, (name, total)
        assert bad == 0, (name, bad, total, maxulp, "host libm FMA build" if fma else "host libm SSE2 build")
