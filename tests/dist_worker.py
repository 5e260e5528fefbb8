"""Worker for tests/test_sharding_gloo.py: one rank of a world_size-N CPU run over gloo.  Each rank
generates ITS index range of the synthetic batch with the counter-based generator, runs the closure
path on it (the oracle stands in for the GPU kernels on this GPU-less box), and the ranks exchange
only a barrier, a max of the elapsed times and their 64-bit checksums."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import cases  # noqa: E402
import oracle_lib as O  # noqa: E402
from rlshaders_amd.sharding import Ranks, shard_range  # noqa: E402


def checksum(arrs) -> int:
    h = 0
    for a in arrs:
        b = np.ascontiguousarray(a).view(np.uint32).astype(np.uint64)
        h = (h + int(((b * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)).sum(dtype=np.uint64))) & ((1 << 64) - 1)
    return h


def main():
    total = int(sys.argv[1])
    r = Ranks(backend="gloo", launched=True)
    first, n = shard_range(total, r.rank, r.world)
    c = cases.ggx_mixed(1234, n, first=first)
    x = cases.xi(1234, n, 4, first=first)
    g = O.Ggx(c["wo"], c["N"], c["T"], KsColor=c["KsColor"], ior=c["ior"], roughness=c["roughness"],
              anisotropic=c["anisotropic"])
    r.barrier()
    t0 = time.perf_counter()
    out = g.reflect_refract(x[0], x[1], x[2], x[3])
    r.barrier()
    elapsed = r.max_over_ranks([time.perf_counter() - t0])[0]
    sums = r.gather_u64(checksum(out))
    # what bench.py gathers to make its line self-evidencing: who took part, and every rank's own time
    import os
    seen = r.gather_objects({"rank": r.rank, "pid": os.getpid(), "first": first, "count": n})
    if r.rank == 0:
        print(json.dumps({"world": r.world, "total": total, "elapsed": elapsed, "checksums": sums, "ranks_seen": seen,
                          "shards": [shard_range(total, k, r.world) for k in range(r.world)]}), flush=True)
    r.close()


if __name__ == "__main__":
    main()
