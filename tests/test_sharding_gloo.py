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
    # gather_objects: every rank reports itself, in rank order, from its own process
    seen = two["ranks_seen"]
    assert [d["rank"] for d in seen] == [0, 1] and len({d["pid"] for d in seen}) == 2
    assert [[d["first"], d["count"]] for d in seen] == two["shards"]
    assert len(one["ranks_seen"]) == 1


def test_ranks_is_inert_without_a_launcher(monkeypatch):
    """a single process whose environment happens to export RANK / MASTER_PORT (a cluster shell) must not start a
    rendezvous: the one-rank process group is opt-in (launched=True)"""
    import torch.distributed as dist
    from rlshaders_amd.sharding import Ranks
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("MASTER_PORT", "1")          # a port nobody may bind: an init would fail loudly
    monkeypatch.setenv("WORLD_SIZE", "1")
    r = Ranks(backend="gloo")
    assert r.dist is None and not dist.is_initialized()
    assert r.max_over_ranks([1.5]) == [1.5] and r.gather_objects("x") == ["x"] and r.gather_u64(7) == [7]
    r.close()


def test_ranks_reuses_an_existing_process_group(tmp_path):
    """a caller that already owns the default process group: Ranks joins it and close() leaves it alive"""
    import torch.distributed as dist
    from rlshaders_amd.sharding import Ranks
    dist.init_process_group("gloo", init_method=f"file://{tmp_path / 'rdv'}", rank=0, world_size=1)
    try:
        r = Ranks(backend="gloo", launched=True)
        assert r.dist is not None and r.world == 1 and r.rank == 0
        assert r.gather_objects({"a": 1}) == [{"a": 1}]
        r.close()
        assert dist.is_initialized()
    finally:
        dist.destroy_process_group()


def test_rendezvous_gives_up_after_the_timeout():
    """a rank whose peer never arrives leaves init_process_group after Ranks.timeout_s (default 120 s,
    RLS_DIST_TIMEOUT_S) instead of torch's 10-30 minutes -- a hung rank cannot burn the driver's half hour.  One process that
    believes it is rank 0 of 2; nobody else comes."""
    import time
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from rlshaders_amd.sharding import Ranks\n"
            "r = Ranks(backend='gloo', launched=True)\n" % str(ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40), RANK="0", WORLD_SIZE="2",
               LOCAL_RANK="0", RLS_DIST_TIMEOUT_S="5")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    took = time.perf_counter() - t0
    assert p.returncode != 0 and 4 < took < 60, (p.returncode, took, p.stderr[-800:])
    assert "imeout" in p.stderr or "timed out" in p.stderr, p.stderr[-800:]
    from rlshaders_amd.sharding import Ranks
    assert Ranks.DEFAULT_TIMEOUT_S == 120.0
