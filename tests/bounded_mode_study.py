#!/usr/bin/env python3
"""The study behind the decision NOT to ship a tolerance-bounded arithmetic mode.

Runs tests/native/bounded_study.c (the oracle with parts of config 2's angle round trip replaced by exactly rounded vector
algebra) over 2^24 points of the bench's seeded workload per variant and prints, per variant: the fraction of points with any
of the twelve outputs beyond 1e-5 of the oracle, beyond 1e-5 / k for safety factors k (a predictor bounds the error from
above, so it flags at least those), and the smallest fraction of lanes single-feature threshold predictors must flag to
catch every exceeding point.  CPU only, test infrastructure; output committed as profiles/r03_bounded_mode_study.txt.

usage: python tests/bounded_mode_study.py [log2 points]"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "native" / "bounded_study.c"
EXE = ROOT / "tests" / "native" / "build" / "bounded_study"
NAMES = {1: "A   first round trip (atan2f + sphericalDirection) by algebra",
         2: "B   tanf(acosf(v.z)) by |(v.x, v.y)| / v.z",
         4: "C   cosf / sinf(atan2f(v.y, v.x)) by v.x / h, v.y / h",
         5: "A+C both azimuth round trips by algebra, acosf / tanf kept exact",
         7: "A+B+C the whole angle round trip by algebra (what RLS_MATH_FAST does, with exact + - * / sqrt here)"}


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    EXE.parent.mkdir(exist_ok=True)
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
                    "-o", str(EXE), str(SRC), "-lm", "-lpthread"], check=True)
    print(f"config 2 (rlGgx reflect + refract, mixed parameters, seed 1234), {1 << log2n} points per variant; the oracle itself as "
          "variant 0 differs from itself on 0 points (checked).")
    with tempfile.TemporaryDirectory() as tmp:
        procs = {}
        for b in [0] + sorted(NAMES):
            procs[b] = subprocess.Popen([str(EXE), str(b), str(log2n), "1234", f"{tmp}/v{b}.bin"])
        for b, p in procs.items():
            assert p.wait() == 0
        r0 = np.fromfile(f"{tmp}/v0.bin", dtype=np.float32).reshape(-1, 6)
        assert (r0[:, 0] == 0).all(), "variant 0 must reproduce the oracle"
        for b in sorted(NAMES):
            r = np.fromfile(f"{tmp}/v{b}.bin", dtype=np.float32).reshape(-1, 6)
            e, th, al, a1, a2, cv = r.T
            bad = e > 1e-5
            print(f"\nvariant {b}: {NAMES[b]}")
            print(f"  outputs that differ from the oracle at all: {np.mean(e > 0):.4f} of the points; median error {np.median(e):.2e}, "
                  f"p99 {np.quantile(e, 0.99):.2e}, max {e.max():.2e}")
            print("  points with an output beyond 1e-5 / k:  " +
                  "  ".join(f"k={k}: {np.mean(e > 1e-5 / k):.4f}" for k in (1, 3, 10, 30, 100)))
            am = np.minimum(a1, a2)
            for name, f in (("theta' (stretched view angle)", th), ("min(alphaX, alphaY)", al), ("min |A^2 - 1| of the two samples", am),
                            ("1 - cos(view)", 1 - cv)):
                lo, hi = f[bad].min(), f[bad].max()
                inside = np.mean((f >= lo) & (f <= hi))
                print(f"  exceeding points span {name} in [{lo:.3g}, {hi:.3g}]: a threshold on it flags {inside:.4f} of all points")


if __name__ == "__main__":
    main()
