#!/usr/bin/env python3
"""Find the fp32 arguments on which glibc's two x86-64 builds of sinf / cosf / expf / powf(x, 5) differ -- i.e. on which
rls_libm.hpp compiled with RLM_GLIBC_FMA = 1 and = 0 differ -- by running both over ALL 2^32 bit patterns on the host
(tests/native/libm_probe.cpp, 8 threads, a few minutes), and write them to tests/golden/libm_flavour_args.json.
tests/cases.py (host_libm_flavour) evaluates the host libm on exactly these arguments to decide which build it runs."""
import ctypes as C
import json
import sys
import threading
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import cases  # noqa: E402

fns = cases.libm_probe_functions()
FN = ["sinf", "cosf", "expf", "powf5"]
CH = 1 << 24
fp = C.POINTER(C.c_float)
found = {k: [] for k in FN}
lock = threading.Lock()


def work(k, chunks):
    for c in chunks:
        bits = (np.arange(CH, dtype=np.uint64) + np.uint64(c) * np.uint64(CH)).astype(np.uint32)
        x = bits.view(np.float32)
        res = {}
        for fl, f in fns.items():
            port, libm = np.empty(CH, np.float32), np.empty(CH, np.float32)
            f(k, CH, x.ctypes.data_as(fp), port.ctypes.data_as(fp), libm.ctypes.data_as(fp))
            res[fl] = port.view(np.uint32)
        a, b = res["fma"], res["sse2"]
        nan = np.isnan(a.view(np.float32)) & np.isnan(b.view(np.float32))
        d = (a != b) & ~nan
        if d.any():
            with lock:
                for i in np.nonzero(d)[0]:
                    found[FN[k]].append({"x_bits": int(bits[i]), "fma_bits": int(a[i]), "sse2_bits": int(b[i])})


for k in range(4):
    th = [threading.Thread(target=work, args=(k, range(t, 256, 8))) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    found[FN[k]].sort(key=lambda e: e["x_bits"])
    print(FN[k], len(found[FN[k]]), flush=True)
out = {"note": "every fp32 argument (all 2^32 bit patterns swept) on which rls_libm.hpp built with RLM_GLIBC_FMA = 1 and = 0 "
               "differ: the arguments that tell glibc's FMA build of the function from its SSE2 build "
               "(tools/find_libm_flavour_args.py)", "functions": found}
json.dump(out, open(ROOT / "tests" / "golden" / "libm_flavour_args.json", "w"), indent=1)
