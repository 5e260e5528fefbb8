"""Oracle (CPU): the in-kernel-sampling integrators take the global index of their first point, so a batch split
into shards -- the unit of multi-GPU sharding and of chunked streaming -- draws the numbers of the unsplit batch."""
import numpy as np

import cases
import oracle_lib as O
from gpu_util_cpu import disney_oracle, ggx_oracle


def _cut(case, sl):
    return {k: (v[..., sl] if isinstance(v, np.ndarray) else v) for k, v in case.items()}


def test_oracle_integrators_shard_by_first_index():
    n, h, spp_n, seed = 1024, 300, 4, 31
    parts = (slice(0, h), slice(h, n))
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    whole = ggx_oracle(O, c).integrate(spp_n, seed)
    pieces = [ggx_oracle(O, _cut(c, sl)).integrate(spp_n, seed, first_index=sl.start) for sl in parts]
    for j in range(2):
        assert np.array_equal(np.concatenate([p[j] for p in pieces], axis=-1).view(np.uint32), whole[j].view(np.uint32))
    # without the offset the second shard repeats the first shard's scrambles and differs
    wrong = ggx_oracle(O, _cut(c, parts[1])).integrate(spp_n, seed)
    assert not np.array_equal(wrong[0], whole[0][:, h:])
    P = np.stack([O.gen_uniform(seed, 0, n, 40 + j, 0.0, 4.0) for j in range(3)])
    lt = O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0))
    whole = ggx_oracle(O, c).direct_lighting(P, lt, spp_n, seed)
    pieces = [ggx_oracle(O, _cut(c, sl)).direct_lighting(np.ascontiguousarray(P[:, sl]), lt, spp_n, seed,
                                                         first_index=sl.start) for sl in parts]
    for j in range(2):
        assert np.array_equal(np.concatenate([p[j] for p in pieces], axis=-1).view(np.uint32), whole[j].view(np.uint32))
    cd = cases.disney_mixed(cases.SEED_PARITY, n)
    whole = disney_oracle(O, cd).integrate(spp_n, seed)
    pieces = [disney_oracle(O, _cut(cd, sl)).integrate(spp_n, seed, first_index=sl.start) for sl in parts]
    for k in whole:
        assert np.array_equal(np.concatenate([p[k] for p in pieces], axis=-1).view(np.uint32), whole[k].view(np.uint32)), k
    dist = np.stack([O.gen_uniform(seed, 0, n, 32 + j, 0.02, 0.3) for j in range(3)])
    sc = O.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    mk = lambda sl: O.Sss(sl.stop - sl.start, np.ascontiguousarray(dist[:, sl]), (0.8, 0.5, 0.3),
                          N=np.ascontiguousarray(c["N"][:, sl]), T=np.ascontiguousarray(c["T"][:, sl]))
    whole = O.integrate_scatter(mk(slice(0, n)), c["N"], sc, spp_n, seed)
    pieces = [O.integrate_scatter(mk(sl), np.ascontiguousarray(c["N"][:, sl]), sc, spp_n, seed, first_index=sl.start)
              for sl in parts]
    for j in range(2):
        assert np.array_equal(np.concatenate([p[j] for p in pieces], axis=-1).view(np.uint32), whole[j].view(np.uint32))
