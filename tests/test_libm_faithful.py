"""CPU: the device elementary functions (rlshaders_amd/csrc/rls_libm.hpp), compiled for the host, against the host
libm, bit for bit.  This is what makes the sampled directions of the HIP kernels identical to the CPU closures'
instead of merely close (the visible-normal slope equations amplify a 1-ulp difference in atan2f/acosf/tanf past
1e-5 on ~0.5 % of points)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def host_libm_uses_fma() -> bool:
    """glibc's ifunc resolvers pick the -mfma builds of sinf/cosf/expf/logf/powf when the CPU has AVX2 and FMA"""
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except (OSError, StopIteration):
        return True
    return "fma" in flags and "avx2" in flags


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
