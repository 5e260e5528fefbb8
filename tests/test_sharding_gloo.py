"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- index-range shards, no data-path
collective, barrier + max-over-ranks timing, checksum gather -- gives the same result as one process."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_shard_ranges_tile_the_batch():
    from rlshaders_amd.sharding import shard_range
    for total in (0, 1, 7, 64, 1000, (1 << 30) + 3):
        for world in (1, 2, 3, 8):
            pos = 0
            for rank in range(world):
                lo, n = shard_range(total, rank, world)
                assert lo == pos and n >= 0
                pos += n
            assert pos == total
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _run(world: int, total: int) -> dict:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29600 + (os.getpid() % 300) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "dist_worker.py"), str(total)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_equal_one_rank():
    total = 100_003                       # odd: the two shards differ in size
    one = _run(1, total)
    two = _run(2, total)
    assert two["world"] == 2 and len(two["checksums"]) == 2
    assert two["shards"] == [[0, total // 2], [total // 2, total - total // 2]]
    # checksum of checksums: additive hash, so shard sums add up to the single-process sum
    assert sum(two["checksums"]) % (1 << 64) == one["checksums"][0]
    assert two["elapsed"] > 0
