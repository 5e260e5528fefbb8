"""CPU: the product says which host libm its EXACT kernels follow and checks the caller's libm against it
(rls_libm_flavour, rls_host_libm_matches: include/rlshaders_amd.h) -- no device needed.  The answer must agree with the
test suite's own probe (tests/cases.py:host_libm_flavour), which decides whether the GPU tests demand bit equality."""
import subprocess
import sys
from pathlib import Path

import cases
import rlshaders_amd as R

ROOT = Path(__file__).resolve().parent.parent


def test_flavour_args_include_is_in_sync():
    subprocess.run([sys.executable, str(ROOT / "tools" / "gen_libm_flavour_inc.py"), "--check"], check=True)


def test_product_and_test_probe_agree():
    flavour = R.libm_flavour()
    assert flavour in ("glibc-fma", "glibc-sse2")
    bad = R.host_libm_mismatches()
    assert bad >= 0
    mine = cases.host_libm_flavour()
    # the library follows glibc's FMA build (RLM_GLIBC_FMA = 1): it matches exactly on the hosts the probe calls "fma"
    assert flavour == "glibc-fma"
    assert (bad == 0) == (mine["flavour"] == "fma"), (bad, mine)
    if mine["flavour"] == "sse2":          # every discriminating argument sides with the other build
        assert bad >= mine["discriminating"]


def test_c_program_can_ask_without_a_device(tmp_path):
    src = tmp_path / "ask.c"
    src.write_text('#include <stdio.h>\n#include "rlshaders_amd.h"\n'
                   'int main(void){ int bad = -1; if (rls_host_libm_matches(&bad) != RLS_OK) return 2;\n'
                   ' if (rls_host_libm_matches(0) != RLS_ERR_INVALID_ARGUMENT) return 3;\n'
                   ' printf("%s %d\\n", rls_libm_flavour(), bad); return 0; }\n')
    exe = tmp_path / "ask"
    lib = ROOT / "rlshaders_amd" / "lib"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe), f"-L{lib}",
                    "-lrlshaders_amd", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == R.libm_flavour() and int(out[1]) == R.host_libm_mismatches()
    # the other way round: glibc told (GLIBC_TUNABLES) to behave as on a CPU without AVX2 / FMA resolves sinf / cosf / expf /
    # powf to its SSE2 build -- the check must then report exactly the discriminating arguments, and nothing else
    if cases.host_libm_flavour()["flavour"] == "fma":
        import os
        env = dict(os.environ, GLIBC_TUNABLES="glibc.cpu.hwcaps=-AVX2,-FMA")
        out = subprocess.run([str(exe)], capture_output=True, text=True, check=True, env=env).stdout.split()
        golden = cases.host_libm_flavour()["discriminating"]
        assert int(out[1]) in (0, golden), out          # 0: a glibc that ignores the tunable (< 2.33 spelling); else all of them
        if int(out[1]) == 0:
            import pytest
            pytest.skip("this glibc does not honour GLIBC_TUNABLES=glibc.cpu.hwcaps=-AVX2,-FMA")
        assert int(out[1]) == golden == 42
