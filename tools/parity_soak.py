#!/usr/bin/env python3
"""Parity soak: every closure family on 2^24 device-generated shading points per seed, GPU (RLS_MATH_EXACT)
against the oracle, counted in output words that differ.  usage: tools/parity_soak.py [--out FILE]
[--log2-points 24] [--seeds 4242,99,7,1234]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import parity_sweep  # noqa: E402
import rlshaders_amd as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--log2-points", type=int, default=24)
    ap.add_argument("--seeds", default="4242,99,7,1234")
    ap.add_argument("--groups", default="1", help="lanes per point of the n^2-spp loops, e.g. 1,4,16 (RLS_INTEGRATE_GROUP)")
    ap.add_argument("--spp-n", type=int, default=2, help="spp_n of the n^2-spp loops (3 -> 9 samples: ragged for 4 and 16 lanes)")
    ap.add_argument("--uniform-draws", type=int, default=0,
                    help="instead: this many random UNIFORM parameter sets per seed through the UNIFORM_ALL kernels (sweep_uniform)")
    ap.add_argument("--by-reference", default="",
                    help="instead: parameters by reference against the same values as planes (sweep_by_reference), for these table "
                         "sizes, e.g. 1,37,65536")
    args = ap.parse_args()
    ctx = R.Context(0)
    total = {}
    t0 = time.time()
    seeds = [int(s) for s in args.seeds.split(",")]
    groups = args.groups.split(",")
    for seed in seeds:
        for group in groups:
            if args.by_reference:
                rep = {}
                for m in [int(x) for x in args.by_reference.split(",")]:
                    for name, r in parity_sweep.sweep_by_reference(ctx, 1 << args.log2_points, seed, m, spp_n=args.spp_n).items():
                        t = rep.setdefault(name, dict(words_differing=0, words=0, max_rel_err=0.0, beyond_1e5=0))
                        for q in ("words_differing", "words", "beyond_1e5"):
                            t[q] += r[q]
                        t["max_rel_err"] = max(t["max_rel_err"], r["max_rel_err"])
            elif args.uniform_draws:
                rep = parity_sweep.sweep_uniform(ctx, 1 << args.log2_points, seed, draws=args.uniform_draws)
            else:
                rep = parity_sweep.sweep(ctx, 1 << args.log2_points, seed, spp_n=args.spp_n, group=group)
            for name, r in rep.items():
                t = total.setdefault(name, dict(words_differing=0, words=0, max_rel_err=0.0, beyond_1e5=0))
                t["words_differing"] += r["words_differing"]
                t["words"] += r["words"]
                t["beyond_1e5"] += r["beyond_1e5"]
                t["max_rel_err"] = max(t["max_rel_err"], r["max_rel_err"])
            if args.out:                  # after every seed: a call that runs into the lease's time limit keeps what it has
                done = seeds[:seeds.index(seed) + 1]
                Path(args.out).write_text(json.dumps(dict(partial=True, seeds_done=done, closures=total,
                                                          words=sum(t["words"] for t in total.values()),
                                                          words_differing=sum(t["words_differing"] for t in total.values())), indent=1))
    summary = dict(points_per_seed=1 << args.log2_points, seeds=seeds, lane_groups=groups, spp_n=args.spp_n, mode="RLS_MATH_EXACT", uniform_draws_per_seed=args.uniform_draws, by_reference_table_sizes=args.by_reference,
                   seconds=round(time.time() - t0, 1), closures=total,
                   words=sum(t["words"] for t in total.values()),
                   words_differing=sum(t["words_differing"] for t in total.values()))
    print(json.dumps(summary, indent=1))
    if args.out:
        Path(args.out).write_text(json.dumps(summary, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
