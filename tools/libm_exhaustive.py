#!/usr/bin/env python3
"""Sweep the device's elementary functions (RLS_MATH_EXACT, through rls_libm_eval) over ALL 2^32 fp32
arguments of the unary ones and over 2^30 argument pairs of atan2f / powf / division, against
(a) the same source compiled for the host and (b) the host libm.  Writes a JSON summary.

usage: tools/libm_exhaustive.py [--out profiles/rNN_libm_exhaustive.json] [--log2-chunk 26] [--quick]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import rlshaders_amd as R  # noqa: E402

FN = dict(sqrtf=0, div=1, atan2f=2, acosf=3, tanf=4, sinf=5, cosf=6, expf=7, logf=8, powf=9,
          tanf_bounded=10, sinf_bounded=11, cosf_bounded=12)
BINARY = {"div", "atan2f", "powf"}
# the forms the closure kernels call: specified for |x| < 120 and NaN only (include/rlshaders_amd.h)
BOUNDED = {"tanf_bounded", "sinf_bounded", "cosf_bounded"}
# exhaustive slices of the two-argument functions: every fp32 value of one argument against a fixed other one
# (name -> (function id, which argument sweeps, the fixed value))
SLICES = {
    "powf(x, 5)": (9, 0, 5.0),            # the Schlick weights of rlDisney (src/rlDisney.cpp:576)
    "powf(x, 0.5)": (9, 0, 0.5),
    "powf(0.25, y)": (9, 1, 0.25),        # a2^(1 - xi) of sampleGTR1Direction (src/rlDisney.cpp:399)
    "atan2f(y, 1)": (2, 0, 1.0),
    "atan2f(y, -0.3)": (2, 0, -0.3),
    "atan2f(0.7, x)": (2, 1, 0.7),
    "atan2f(-1e-3, x)": (2, 1, -1e-3),
    "x / 3": (1, 0, 3.0),
    "x / 0.3333": (1, 0, 0.3333),
    "1 / y": (1, 1, 1.0),
    "2 / y": (1, 1, 2.0),
    "0.31830988 / y": (1, 1, 0.31830988),
}


def host_lib() -> C.CDLL:
    out = ROOT / "tests" / "native" / "build"
    out.mkdir(exist_ok=True)
    so = out / "liblibm_host.so"
    src = ROOT / "tests" / "native" / "libm_host.cpp"
    hdr = ROOT / "rlshaders_amd" / "csrc" / "rls_libm.hpp"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=gnu++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
                        f"-I{ROOT / 'rlshaders_amd' / 'csrc'}", str(src), "-o", str(so), "-lm"], check=True)
    lib = C.CDLL(str(so))
    fp = C.POINTER(C.c_float)
    lib.libm_host_eval.argtypes = [C.c_int, C.c_int64, fp, fp, fp, fp, C.c_int]
    lib.libm_host_eval.restype = None
    return lib


def ulp_key(a: np.ndarray) -> np.ndarray:
    i = a.view(np.int32).astype(np.int64)
    return np.where(i < 0, -(i & 0x7FFFFFFF), i)


def compare(dev: np.ndarray, ref: np.ndarray):
    """-> (mismatches, max ulp distance); NaN == NaN whatever the payload"""
    bad = dev.view(np.uint32) != ref.view(np.uint32)
    bad &= ~(np.isnan(dev) & np.isnan(ref))
    bad &= ~((dev == 0) & (ref == 0) & (np.signbit(dev) == np.signbit(ref)))
    if not bad.any():
        return 0, 0, None
    d = np.abs(ulp_key(dev[bad]) - ulp_key(ref[bad]))
    fin = np.isfinite(dev[bad]) & np.isfinite(ref[bad])
    k = int(np.argmax(bad))
    return int(bad.sum()), int(d[fin].max()) if fin.any() else -1, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--log2-chunk", type=int, default=26)
    ap.add_argument("--quick", action="store_true", help="every 64th chunk only")
    ap.add_argument("--only", default="")
    ap.add_argument("--slices", action="store_true", help="also sweep the exhaustive one-argument slices of powf / atan2f / division")
    ap.add_argument("--slices-only", action="store_true")
    ap.add_argument("--math", default="exact", choices=["exact", "fast"],
                    help="fast: characterise the RLS_MATH_FAST (hardware) forms -- error in ulp against the host libm on "
                         "the arguments closures produce")
    args = ap.parse_args()
    hl = host_lib()
    ctx = R.Context(0)
    ctx.set_math_mode(args.math == "fast")
    nthreads = len(os.sched_getaffinity(0))
    chunk = 1 << args.log2_chunk
    fp = C.POINTER(C.c_float)
    res = {}
    work = ([] if args.slices_only else list(FN.items())) + \
        ([(k, v[0]) for k, v in SLICES.items()] if (args.slices or args.slices_only) else [])
    for name, fn in work:
        if args.only and name not in args.only.split(","):
            continue
        sl = SLICES.get(name)
        binary = name in BINARY
        total_chunks = (1 << 32) // chunk if (not binary or sl is not None) else (1 << 30) // chunk
        if sl is not None:
            binary = False          # reported as exhaustive over the swept argument
        bad_port = bad_libm = 0
        max_port = max_libm = 0
        first = None
        done = 0
        fast_max = fast_gt1 = fast_gt4 = fast_nonfinite = 0
        fast_sum = fast_abs = 0.0
        t0 = time.time()
        gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + fn)
        for c in range(total_chunks):
            if args.quick and c % 64:
                continue
            if sl is not None:
                base = c * chunk - 2 ** 31
                sweep = (torch.arange(chunk, device="cuda", dtype=torch.int64) + base).to(torch.int32).view(torch.float32)
                fixed = torch.full((chunk,), sl[2], device="cuda", dtype=torch.float32)
                x, y = (sweep, fixed) if sl[1] == 0 else (fixed, sweep)
            elif binary:
                # raw bit patterns for both arguments: every exponent, signs, denormals, NaN/Inf included
                xb = torch.randint(-2 ** 31, 2 ** 31, (chunk,), device="cuda", dtype=torch.int32, generator=gen)
                yb = torch.randint(-2 ** 31, 2 ** 31, (chunk,), device="cuda", dtype=torch.int32, generator=gen)
                if c % 2:   # every other chunk: moderate magnitudes, where the closures live
                    xb = (torch.rand(chunk, device="cuda", generator=gen) * 200 - 100).view(torch.int32)
                    yb = (torch.rand(chunk, device="cuda", generator=gen) * 40 - 20).view(torch.int32)
                x, y = xb.view(torch.float32), yb.view(torch.float32)
            else:
                base = c * chunk - 2 ** 31
                x = (torch.arange(chunk, device="cuda", dtype=torch.int64) + base).to(torch.int32).view(torch.float32)
                y = None
            out = torch.empty(chunk, device="cuda", dtype=torch.float32)
            two = binary or sl is not None
            R._capi.check(ctx.lib.rls_libm_eval(ctx.handle, fn, chunk, x.data_ptr(), y.data_ptr() if two else None,
                                                out.data_ptr()))
            torch.cuda.synchronize()
            xd, od = x.cpu().numpy(), out.cpu().numpy()
            yd = y.cpu().numpy() if two else None
            port, libm = np.empty(chunk, np.float32), np.empty(chunk, np.float32)
            hl.libm_host_eval(fn, chunk, xd.ctypes.data_as(fp), yd.ctypes.data_as(fp) if two else None,
                              port.ctypes.data_as(fp), libm.ctypes.data_as(fp), nthreads)
            if args.math == "fast":
                # the domain the closures use each function on (normal, finite, moderate magnitudes)
                ax = np.abs(xd)
                keep = np.isfinite(xd) & (ax > 1e-30) & (ax < 1e30)
                if name.startswith(("sinf", "cosf", "tanf")):
                    keep &= ax <= 6.2831855
                if name.startswith("tanf"):
                    with np.errstate(all="ignore"):
                        keep &= np.abs(np.cos(xd.astype(np.float64))) > 1e-3
                if name == "acosf":
                    keep &= ax <= 1.0
                if name in ("sqrtf", "logf"):
                    keep &= xd > 0
                if name == "expf":
                    keep &= ax < 87.0
                if binary:
                    ay = np.abs(yd)
                    keep &= np.isfinite(yd) & (ay > 1e-30) & (ay < 1e30)
                if name == "powf":
                    keep &= (xd > 0) & (ax < 1e3) & (ax > 1e-3) & (ay < 20.0)
                if name == "div":
                    with np.errstate(all="ignore"):
                        q = np.abs(xd.astype(np.float64) / yd.astype(np.float64))
                    keep &= (q > 1e-30) & (q < 1e30)
                xd, od, port, libm = xd[keep], od[keep], port[keep], libm[keep]
                in_domain = int(keep.sum())
                fin = np.isfinite(od) & np.isfinite(libm)
                d = np.abs(ulp_key(od[fin]) - ulp_key(libm[fin]))
                fast_max = max(fast_max, int(d.max()) if d.size else 0)
                fast_gt1 += int((d > 1).sum()); fast_gt4 += int((d > 4).sum()); fast_sum += float(d.sum())
                fast_nonfinite += int((~fin).sum())
                if fin.any():
                    fast_abs = max(fast_abs, float(np.abs(od[fin].astype(np.float64) - libm[fin].astype(np.float64)).max()))
                done += in_domain
                continue
            if name in BOUNDED:
                keep = (np.abs(xd) < 120.0) | np.isnan(xd)
                xd, od, port, libm = xd[keep], od[keep], port[keep], libm[keep]
                in_domain = int(keep.sum())
            else:
                in_domain = chunk
            b, m, k = compare(od, port)
            bad_port += b; max_port = max(max_port, m)
            if b and first is None:
                first = dict(x=float(xd[k]), x_bits=hex(int(xd.view(np.uint32)[k])), device=float(od[k]), port=float(port[k]))
            b, m, _ = compare(od, libm)
            bad_libm += b; max_libm = max(max_libm, m)
            done += in_domain
        if args.math == "fast":
            res[name] = dict(arguments_in_domain=done, max_ulp=fast_max, mean_ulp=fast_sum / max(done, 1),
                             fraction_beyond_1ulp=fast_gt1 / max(done, 1), fraction_beyond_4ulp=fast_gt4 / max(done, 1),
                             max_abs_err=fast_abs, nonfinite_mismatch=fast_nonfinite, seconds=round(time.time() - t0, 1))
            print(name, json.dumps(res[name]), flush=True)
            continue
        res[name] = dict(arguments=done, exhaustive=(not binary and not args.quick),
                         domain="|x| < 120 or NaN" if name in BOUNDED else "all bit patterns",
                         mismatch_vs_same_source_on_host=bad_port, max_ulp_vs_same_source=max_port,
                         mismatch_vs_host_libm=bad_libm, max_ulp_vs_host_libm=max_libm,
                         first_mismatch=first, seconds=round(time.time() - t0, 1))
        print(name, json.dumps(res[name]), flush=True)
    if args.math == "fast":
        summary = dict(mode="RLS_MATH_FAST", functions=res,
                       note="error of the hardware forms in ulp of the host libm's result, over every fp32 argument in the "
                            "domain the closures use (|x| in (1e-30, 1e30); angles within 2 pi, tanf away from its poles; "
                            "acosf on [-1, 1]; expf below 87; powf for bases in (1e-3, 1e3) and |y| < 20)")
        if args.out:
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            Path(args.out).write_text(json.dumps(summary, indent=1))
        ctx.close()
        return
    summary = dict(mode="RLS_MATH_EXACT", host_threads=nthreads, functions=res,
                   note="device = rls_libm_eval on gfx950; same source on host = rlshaders_amd/csrc/rls_libm.hpp built with "
                        "g++ -ffp-contract=off; host libm = glibc of this image (its FMA multiarch sinf/cosf/expf/powf "
                        "contract the fp64 polynomial, which moves the final rounding of a ~1e-7 fraction of arguments by 1 ulp)")
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(summary, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
