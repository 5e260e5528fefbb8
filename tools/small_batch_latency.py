#!/usr/bin/env python3
"""Latency of one small flush through the n^2-spp loops: lanes per point chosen automatically (4 / 16 / 64 for batches too small
to fill the GPU; the sums still grow in sample order: fold, rls_loops.hpp) against one lane per point (RLS_INTEGRATE_GROUP=1).
usage: tools/small_batch_latency.py  -> one JSON line"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import rlshaders_amd as R  # noqa: E402


def timed(fn, ctx, reps=30):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ctx = R.Context(0)
    rows = []
    for log2n in (8, 10, 12, 14, 16, 18):
        n = 1 << log2n
        wo, N, T = R.gen_frame(ctx, 1234, 0, n)
        u = lambda s, lo=0.0, hi=1.0: R.gen_uniform(ctx, 1234, 0, n, s, lo, hi)
        base = torch.stack([u(8 + j) for j in range(3)])
        d = R.DisneySampler(ctx, wo, N, T, base_color=base, **{k: u(32 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)})
        g = R.GgxSampler(ctx, wo, N, T, specColor=base, ior=u(6, 1.05, 2.55), roughness=u(5, 0.05, 1.0), anisotropic=0.0)
        out = d.integrate(8, 7)
        outg = g.integrate(8, 7)
        row = {"points": n}
        for label, env in (("auto", None), ("one_lane", "1")):
            if env is None:
                os.environ.pop("RLS_INTEGRATE_GROUP", None)
            else:
                os.environ["RLS_INTEGRATE_GROUP"] = env
            row[f"disney_64spp_us_{label}"] = round(timed(lambda: d.integrate(8, 7, out=out), ctx), 1)
            row[f"ggx_64spp_us_{label}"] = round(timed(lambda: g.integrate(8, 7, out=outg), ctx), 1)
        os.environ.pop("RLS_INTEGRATE_GROUP", None)
        rows.append(row)
    print(json.dumps({"note": "microseconds per call, 64 samples per point and lobe, EXACT; auto = lanes per point chosen from the batch size",
                      "rows": rows}))


if __name__ == "__main__":
    main()
