"""GPU parity of the whole-node entry points rls_ggx_shade (shader_evaluate of rlGgx, src/rlGgx.cpp:248-327) and
rls_disney_shade (src/rlDisney.cpp:685-727) against the oracle, and their consistency with the single-purpose entry points
they are made of."""
import os

import numpy as np
import pytest

import cases
import rlshaders_amd as R
from gpu_util import dev, disney_oracle, disney_sampler, ggx_oracle, ggx_sampler, host

pytestmark = pytest.mark.gpu

LIGHTS = (dict(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0), mis_mode=0),
          dict(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=2))


def _with_group(g, fn):
    os.environ["RLS_INTEGRATE_GROUP"] = str(g)
    try:
        return fn()
    finally:
        del os.environ["RLS_INTEGRATE_GROUP"]


def _lights(oracle, specs=LIGHTS):
    lo = [oracle.make_light(**kw) for kw in specs]
    return lo, [R._capi.SphereLight.from_buffer_copy(bytes(l)) for l in lo]


def _slab(n):
    return (cases.xi(cases.SEED_PARITY, n, 3) * np.array([[4.0], [4.0], [1.0]], np.float32)).astype(np.float32)


@pytest.mark.parametrize("traced", [True, False])
def test_ggx_shade_matches_oracle(gpu, oracle, traced):
    n, spp_n, seed, first = 1 << 12, 3, 41, 1 << 36
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    P = _slab(n)
    u = lambda j, lo=0.0, hi=1.0: oracle.gen_uniform(cases.SEED_PARITY, 0, n, oracle.S_PARAM0 + j, lo, hi)
    kdc, ktc = np.stack([u(j) for j in range(3)]), np.stack([u(3 + j) for j in range(3)])
    kd, kdr, ks, kt = u(6), u(7), u(8), u(9)
    k = np.arange(n) % 8
    kd = np.where(k == 0, np.float32(0.0), kd).astype(np.float32)          # sampleDiffuse false
    kt = np.where(k == 1, np.float32(0.0), kt).astype(np.float32)          # no transmission
    ksc = np.where((k == 2)[None, :], np.float32(0.0), c["KsColor"]).astype(np.float32)   # black KsColor: no integrateGlossy
    c = dict(c, KsColor=ksc)
    lo, lg = _lights(oracle)
    env = (0.7, 0.8, 0.9)
    ref = ggx_oracle(oracle, c, nthreads=oracle.hardware_threads()).shade(
        P, lo, spp_n, seed, Kd_color=kdc, Kd=kd, Kd_roughness=kdr, Ks=ks, Kt_color=ktc, Kt=kt, env=env, traced=traced,
        first_index=first)
    s = ggx_sampler(gpu, c)
    run = lambda lights=lg: {q: host(v) for q, v in s.shade(
        dev(P), lights, spp_n, seed, KdColor=dev(kdc), Kd=dev(kd), diffuseRoughness=dev(kdr), Ks=dev(ks), KtColor=dev(ktc),
        Kt=dev(kt), env=env, traced=traced, first_index=first).items()}
    got = _with_group(1, run)
    for q in ref:
        st = cases.summarize(cases.rel_err(got[q], ref[q]))
        print("ggx shade", traced, q, st)
        cases.assert_tight(st, q)
    assert (got["direct_diffuse"][:, k == 0] == 0).all() and (got["indirect_diffuse"][:, k == 0] == 0).all()
    assert (got["refraction"][:, k == 1] == 0).all() and (got["refraction"][:, k > 2] != 0).any()
    assert (got["indirect_specular"][:, k == 2] == 0).all()
    assert (ref["direct_specular"] > 0).mean() > 0.1 and (ref["indirect_diffuse"] > 0).mean() > 0.5
    # sg->out.RGB is the sum of the AOVs in the reference's order (311, 323)
    want = ((got["direct_diffuse"] + got["direct_specular"]) + got["refraction"]) + (got["indirect_diffuse"] + got["indirect_specular"])
    assert np.array_equal(want.view(np.uint32), got["out"].view(np.uint32))
    # the light loop inside is the light-loop entry point (same streams, same order)
    dd, ds = _with_group(1, lambda: [host(t) for t in s.directLighting(dev(P), lg, spp_n, seed, KdColor=dev(kdc), Kd=dev(kd),
                                                                     diffuseRoughness=dev(kdr), Ks=dev(ks), first_index=first)])
    assert np.array_equal(dd.view(np.uint32), got["direct_diffuse"].view(np.uint32))
    assert np.array_equal(ds.view(np.uint32), got["direct_specular"].view(np.uint32))
    # no lights: the direct AOVs are black, the others unchanged
    dark = _with_group(1, lambda: run(None))
    assert not dark["direct_diffuse"].any() and not dark["direct_specular"].any()
    for q in ("refraction", "indirect_diffuse", "indirect_specular"):
        assert np.array_equal(dark[q].view(np.uint32), got[q].view(np.uint32)), q
    for g in (4, 16):
        alt = _with_group(g, run)
        for q in ref:
            cases.assert_same_bits(alt[q], got[q], (g, q))    # sums in sample order whatever the group width


def test_disney_shade_matches_oracle(gpu, oracle):
    n, spp_n, seed, first = 1 << 12, 4, 43, 98765
    c = cases.disney_mixed(cases.SEED_PARITY, n)
    P = _slab(n)
    lo, lg = _lights(oracle)
    env = (0.7, 0.8, 0.9)
    ref = disney_oracle(oracle, c).shade(P, lo, spp_n, seed, env=env, first_index=first)
    d = disney_sampler(gpu, c)
    run = lambda lights=lg: {q: host(v) for q, v in d.shade(dev(P), lights, spp_n, seed, env=env, first_index=first).items()}
    got = _with_group(1, run)
    for q in ref:
        st = cases.summarize(cases.rel_err(got[q], ref[q]))
        print("disney shade", q, st)
        cases.assert_tight(st, q)
    want = (got["direct_diffuse"] + got["direct_specular"]) + (got["indirect_diffuse"] + got["indirect_specular"])
    assert np.array_equal(want.view(np.uint32), got["out"].view(np.uint32))
    dd, ds = _with_group(1, lambda: [host(t) for t in d.directLighting(dev(P), lg, spp_n, seed, first_index=first)])
    assert np.array_equal(dd.view(np.uint32), got["direct_diffuse"].view(np.uint32))
    assert np.array_equal(ds.view(np.uint32), got["direct_specular"].view(np.uint32))
    assert (ref["indirect_diffuse"] > 0).mean() > 0.5 and (ref["indirect_specular"] > 0).mean() > 0.5
    for g in (4, 16, 64):
        alt = _with_group(g, run)
        for q in ref:
            cases.assert_same_bits(alt[q], got[q], (g, q))    # sums in sample order whatever the group width


def test_shade_hostile_inputs_and_argument_checks(gpu, oracle):
    from test_gpu_hostile_inputs import _poison, _same
    n, spp_n, seed = 1 << 11, 2, 3
    rng = np.random.default_rng(21)
    lo, lg = _lights(oracle)
    c = cases.ggx_mixed(cases.SEED_EDGE, n)
    for q in ("wo", "N", "T", "roughness", "ior", "anisotropic", "KsColor"):
        c[q] = _poison(c[q], rng)
    P = _poison(_slab(n), rng)
    kw = dict(Kd=0.6, Ks=0.4, Kt=0.5)
    ref = ggx_oracle(oracle, c, nthreads=4).shade(P, lo, spp_n, seed, Kd_color=(0.9, 0.5, 0.3), Kd_roughness=0.4,
                                                  Kt_color=(0.2, 0.9, 0.9), **kw)
    s = ggx_sampler(gpu, c)
    got = _with_group(1, lambda: {q: host(v) for q, v in s.shade(dev(P), lg, spp_n, seed, KdColor=(0.9, 0.5, 0.3),
                                                                 diffuseRoughness=0.4, KtColor=(0.2, 0.9, 0.9), **kw).items()})
    for q in ref:
        _same(got[q], ref[q], f"ggx shade {q}")
    d = cases.disney_mixed(cases.SEED_EDGE, n)
    for q in ("wo", "N", "T", "base_color", "roughness", "metallic", "clearcoat", "clearcoat_gloss", "subsurface"):
        d[q] = _poison(d[q], rng)
    ref = disney_oracle(oracle, d).shade(P, lo, spp_n, seed)
    got = _with_group(1, lambda: {q: host(v) for q, v in disney_sampler(gpu, d).shade(dev(P), lg, spp_n, seed).items()})
    for q in ref:
        _same(got[q], ref[q], f"disney shade {q}")
    with pytest.raises(R.RlsError):
        s.shade(dev(P), lg * 5, spp_n, seed)                              # ten lights
    with pytest.raises(R.RlsError):
        s.shade(dev(P), lg, 17, seed)                                     # spp_n > 16


@pytest.mark.parametrize("n", [1, 5, 67])
def test_tiny_batches_pick_lane_groups(gpu, oracle, n):
    """a handful of shading points: the host gives each point 64 lanes (pick_group), workgroups are mostly padding lanes,
    and the packed queues carry the requests of lanes that share a point -- the results agree with the oracle up to the
    order of the sums"""
    spp_n, seed = 4, 91
    c = cases.ggx_mixed(cases.SEED_PARITY, 128)
    c = {k: np.ascontiguousarray(v[..., :n]) for k, v in c.items()}
    d = cases.disney_mixed(cases.SEED_PARITY, 128)
    d = {k: np.ascontiguousarray(v[..., :n]) for k, v in d.items()}
    P = _slab(128)[:, :n].copy()
    lo, lg = _lights(oracle)
    ref = ggx_oracle(oracle, c).shade(P, lo, spp_n, seed, Kt=0.5)
    got = {q: host(v) for q, v in ggx_sampler(gpu, c).shade(dev(P), lg, spp_n, seed, Kt=0.5).items()}
    for q in ref:
        scale = np.abs(ref[q]).max() + 1e-6
        assert np.abs(got[q] - ref[q]).max() <= 2e-4 * scale, (q, got[q], ref[q])
    ref = disney_oracle(oracle, d).shade(P, lo, spp_n, seed)
    got = {q: host(v) for q, v in disney_sampler(gpu, d).shade(dev(P), lg, spp_n, seed).items()}
    for q in ref:
        scale = np.abs(ref[q]).max() + 1e-6
        assert np.abs(got[q] - ref[q]).max() <= 2e-4 * scale, (q, got[q], ref[q])
    s = cases.skin_mixed(cases.SEED_PARITY, 128)
    sp = {k: np.ascontiguousarray(v[..., :n]) for k, v in s["params"].items()}
    wo, N, T = (np.ascontiguousarray(s[k][:, :n]) for k in ("wo", "N", "T"))
    kw = dict(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    ref = oracle.skin_integrate(wo, N, T, sp, N, oracle.make_scene(**kw), spp_n, seed, lights=lo[:1])
    sk = R.SkinShader(gpu, dev(wo), dev(N), dev(T), **{k: dev(v) for k, v in sp.items()})
    got = {q: host(v) for q, v in sk.integrate(dev(N), R.make_scene(**kw), spp_n, seed, lights=lg[:1]).items()}
    for q in ref:
        scale = np.abs(ref[q]).max() + 1e-6
        assert np.abs(got[q] - ref[q]).max() <= 5e-4 * scale, (q, got[q], ref[q])
