"""CPU: the oracle's batch drivers under AddressSanitizer + UBSan (GPU sanitizers are not available on
the pool, so memory-safety checking happens on the CPU build of the checker itself)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

DRIVER = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
O.LIB_PATH = O.ORACLE_DIR / "build" / "librls_oracle_asan.so"
import ctypes
O._lib = ctypes.CDLL(str(O.LIB_PATH))
O._lib.orc_hardware_threads.restype = ctypes.c_int
import cases
n = 5001
c = cases.ggx_mixed(3, n); x = cases.xi(3, n, 6)
g = O.Ggx(c["wo"], c["N"], c["T"], KsColor=c["KsColor"], ior=c["ior"], roughness=c["roughness"], anisotropic=c["anisotropic"], nthreads=3)
g.reflect_refract(x[0], x[1], x[2], x[3]); g.integrate(3, 5); g.microfacet(x[0], x[1], True); g.ndf_pdf(c["wo"])
d = cases.disney_mixed(3, n)
D = O.Disney(d["wo"], d["N"], d["T"], base_color=d["base_color"], nthreads=2, **{k: d[k] for k in O.DISNEY_SCALARS})
D.sample_eval_pdf(O.RAY_GLOSSY, x[0], x[1]); D.sample_eval_pdf(O.RAY_DIFFUSE, x[0], x[1]); D.integrate(2, 9, streamed=True)
s = cases.sss_mixed(3, n)
S = O.Sss(n, s["dist"], s["albedo"], N=s["N"], T=s["T"], nthreads=2)
S.probe(x[0], x[1]); S.nd_sample(x[0]); S.mis_pdf(x[:3] - 0.5, s["N"], True)
k = cases.skin_mixed(3, n)
O.skin(k["wo"], k["N"], k["T"], k["params"], x, nthreads=2)
O.gen_frame(1, 0, 777); O.gen_aniso(1, 5, 100)
print("asan-ok")
'''


def test_oracle_under_asan_ubsan():
    p = subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", DRIVER], capture_output=True, text=True, env=env, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0 and "asan-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr, r.stderr[-3000:]
