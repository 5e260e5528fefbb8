"""GPU parity of rls_skin_integrate (rlSkin's shader_evaluate over n^2 samples per layer with the mean-Fresnel hand-down,
src/rlSkin.cpp:174-254) and rls_ggx_integrate_refract (src/rlGgx.h:205-245) against the oracle."""
import os

import numpy as np
import pytest

import cases
import rlshaders_amd as R
from gpu_util import dev, ggx_oracle, ggx_sampler, host

pytestmark = pytest.mark.gpu

KEYS = ("sheen", "specular", "sss", "out", "sheenFresnel", "specularFresnel", "sssWeight")


def _with_group(g, fn):
    os.environ["RLS_INTEGRATE_GROUP"] = str(g)
    try:
        return fn()
    finally:
        del os.environ["RLS_INTEGRATE_GROUP"]


def _skin(gpu, c, params):
    return R.SkinShader(gpu, dev(c["wo"]), dev(c["N"]), dev(c["T"]), **{k: dev(v) for k, v in params.items()})


@pytest.mark.parametrize("geometry", ["sphere", "plane"])
def test_skin_integrate_matches_oracle(gpu, oracle, geometry):
    n, spp_n, seed, first = 1 << 11, 4, 77, 123456789
    c = cases.skin_mixed(cases.SEED_PARITY, n)
    p = c["params"]
    if geometry == "sphere":
        kw = dict(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
        P = c["N"]
    else:
        kw = dict(geometry="plane", plane_normal=(0.0, 0.0, 1.0), light_dir=(0.0, 0.6, 0.8), use_cavity_fade=False,
                  gate_point=(0.1, 0.0, 0.0), gate_normal=(1.0, 0.0, 0.0))
        P = np.zeros((3, n), np.float32)
        P[:2] = np.stack([oracle.gen_uniform(seed, 0, n, 40 + j, -0.5, 0.5) for j in range(2)])
        c = dict(c, N=np.tile(np.array([[0.0], [0.0], [1.0]], np.float32), (1, n)),
                 T=np.tile(np.array([[1.0], [0.0], [0.0]], np.float32), (1, n)))
        c["wo"] = cases.frame(cases.SEED_PARITY, n)[0]
        c["wo"][2] = np.abs(c["wo"][2]) + 0.05
        c["wo"] /= np.linalg.norm(c["wo"], axis=0, keepdims=True).astype(np.float32)
    env = (1.0, 0.9, 0.8)
    ref = oracle.skin_integrate(c["wo"], c["N"], c["T"], p, P, oracle.make_scene(**kw), spp_n, seed, env=env,
                                first_index=first, nthreads=4)
    sk = _skin(gpu, c, p)
    scene = R.make_scene(**kw)
    got = _with_group(1, lambda: {k: host(v) for k, v in sk.integrate(dev(P), scene, spp_n, seed, env=env,
                                                                      first_index=first).items()})
    for k in KEYS:
        st = cases.summarize(cases.rel_err(got[k], ref[k]))
        diff = int((got[k].view(np.uint32) != ref[k].view(np.uint32)).sum())
        print("skin integrate", geometry, k, st, "words differing", diff)
        cases.assert_tight(st, (geometry, k))
    assert (ref["sss"] > 0).mean() > 0.3                      # the probe rays do find lit surface
    for g in (4, 16):
        alt = _with_group(g, lambda: {k: host(v) for k, v in sk.integrate(dev(P), scene, spp_n, seed, env=env,
                                                                          first_index=first).items()})
        for k in KEYS:
            cases.assert_same_bits(alt[k], got[k], (g, k))    # sums in sample order whatever the group width


def test_skin_integrate_layer_gates_and_presets(gpu, oracle):
    """weights at and below AI_EPSILON skip their lobe (src/rlSkin.cpp:191,214,244); black colours sample nothing and
    hand down getAvgReflectWeight() = 1; the reference's own scene blocks as uniform parameters"""
    n, spp_n, seed = 1 << 10, 4, 9
    c = cases.skin_mixed(cases.SEED_PARITY, n)
    k = np.arange(n) % 6
    p = dict(c["params"])
    p["sheen_weight"] = np.where(k == 0, np.float32(0.0), np.where(k == 1, np.float32(1e-4), p["sheen_weight"])).astype(np.float32)
    p["specular_weight"] = np.where(k == 2, np.float32(0.0), p["specular_weight"]).astype(np.float32)
    p["sss_weight"] = np.where(k == 3, np.float32(0.0), p["sss_weight"]).astype(np.float32)
    p["specular_color"] = np.where(k == 4, np.float32(0.0), p["specular_color"]).astype(np.float32)
    kw = dict(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    ref = oracle.skin_integrate(c["wo"], c["N"], c["T"], p, c["N"], oracle.make_scene(**kw), spp_n, seed, nthreads=4)
    got = _with_group(1, lambda: {q: host(v) for q, v in _skin(gpu, c, p).integrate(dev(c["N"]), R.make_scene(**kw),
                                                                                   spp_n, seed).items()})
    for q in KEYS:
        cases.assert_tight(cases.summarize(cases.rel_err(got[q], ref[q])), q)
    assert (got["sheen"][:, k <= 1] == 0).all() and (got["sss"][:, k == 3] == 0).all()
    assert np.array_equal(got["specularFresnel"][k == 4], p["specular_weight"][k == 4])
    for name, preset in cases.SKIN_PRESETS.items():
        ref = oracle.skin_integrate(c["wo"], c["N"], c["T"], preset, c["N"], oracle.make_scene(**kw), spp_n, seed, nthreads=4)
        sk = R.SkinShader(gpu, dev(c["wo"]), dev(c["N"]), dev(c["T"]), **preset)
        got = _with_group(1, lambda: {q: host(v) for q, v in sk.integrate(dev(c["N"]), R.make_scene(**kw), spp_n, seed).items()})
        for q in KEYS:
            cases.assert_tight(cases.summarize(cases.rel_err(got[q], ref[q])), (name, q))


@pytest.mark.parametrize("traced", [True, False])
def test_integrate_refract_matches_oracle(gpu, oracle, traced):
    n, spp_n, seed, first = 1 << 12, 4, 31, 1 << 35
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    ex = (np.arange(n) % 3 == 0).astype(np.uint8)               # a third of the points leave the medium: TIR happens
    env = (2.0, 1.0, 0.5)
    ref, rtir = ggx_oracle(oracle, c, exiting=ex).integrate_refract(spp_n, seed, traced=traced, env=env, first_index=first)
    s = ggx_sampler(gpu, c, exiting=ex)
    got, tir = _with_group(1, lambda: [host(t) for t in s.integrateRefract(spp_n, seed, traced=traced, env=env,
                                                                          want_tir=True, first_index=first)])
    st = cases.summarize(cases.rel_err(got, ref))
    print("integrateRefract", traced, st, "tir mean", float(tir.mean()))
    cases.assert_tight(st, "result")
    assert np.array_equal(tir.view(np.uint32), rtir.view(np.uint32))
    assert tir[ex == 1].mean() > 0.01 and (tir[ex == 0] == 0).all()
    if traced:
        for g in (4, 16):
            alt, atir = _with_group(g, lambda: [host(t) for t in s.integrateRefract(spp_n, seed, traced=True, env=env,
                                                                                   want_tir=True, first_index=first)])
            assert np.array_equal(atir, tir)
            cases.assert_same_bits(alt, got, g)    # sums in sample order whatever the group width


@pytest.mark.parametrize("spp_n", [3, 4])
def test_skin_integrate_with_light_loops(gpu, oracle, spp_n):
    """the light loops of src/rlSkin.cpp:193-198 / 217-222 (evalLightSample per light, before integrateGlossy): their
    BSDF samples enter getAvgReflectWeight's mean, so the hand-down scalars change with the lights; spp_n = 3 makes the
    sample count 9 (the mean is a division by the count, not a multiplication by its reciprocal)"""
    n, seed, first = 1 << 11, 5, 1 << 34
    c = cases.skin_mixed(cases.SEED_PARITY, n)
    p = dict(c["params"])
    # a tenth of the points with a black sheen colour: integrateGlossy samples nothing there, the light loop does
    black = (np.arange(n) % 10) == 0
    p["sheen_color"] = np.where(black[None, :], np.float32(0.0), p["sheen_color"]).astype(np.float32)
    kw = dict(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    P = c["N"]
    specs = (dict(center=(0.5, 0.5, 4.0), radius=1.0, radiance=(2.0, 1.5, 1.0), mis_mode=0),
             dict(center=(-3.0, 0.0, 1.0), radius=0.7, radiance=(0.2, 0.4, 3.0), mis_mode=2),
             dict(center=(0.0, 3.0, -1.0), radius=0.9, radiance=(1.0, 0.1, 0.1), mis_mode=1))
    lo = [oracle.make_light(**s) for s in specs]
    lg = [R._capi.SphereLight.from_buffer_copy(bytes(l)) for l in lo]
    env = (0.3, 0.3, 0.4)
    ref = oracle.skin_integrate(c["wo"], c["N"], c["T"], p, P, oracle.make_scene(**kw), spp_n, seed, env=env,
                                first_index=first, nthreads=4, lights=lo)
    dark = oracle.skin_integrate(c["wo"], c["N"], c["T"], p, P, oracle.make_scene(**kw), spp_n, seed, env=env,
                                 first_index=first, nthreads=4)
    sk = _skin(gpu, c, p)
    scene = R.make_scene(**kw)
    got = _with_group(1, lambda: {k: host(v) for k, v in sk.integrate(dev(P), scene, spp_n, seed, env=env,
                                                                      first_index=first, lights=lg).items()})
    for k in KEYS:
        st = cases.summarize(cases.rel_err(got[k], ref[k]))
        print("skin integrate lit", spp_n, k, st)
        cases.assert_tight(st, ("lit", k))
    # the lights add light and move the Fresnel means
    assert (ref["sheen"].sum(0) > dark["sheen"].sum(0)).mean() > 0.3
    assert (ref["sheenFresnel"] != dark["sheenFresnel"]).mean() > 0.5
    # black sheen colour, no lights: no samples at all, getAvgReflectWeight() = 1 (src/rlGgx.h:181-184)
    on = black & (p["sheen_weight"] > 1e-4)
    assert np.array_equal(dark["sheenFresnel"][on], p["sheen_weight"][on])
    assert (ref["sheenFresnel"][on] != p["sheen_weight"][on]).mean() > 0.5     # with lights: the light loops' mean
    # without lights the call is the one it was
    unlit = _with_group(1, lambda: {k: host(v) for k, v in sk.integrate(dev(P), scene, spp_n, seed, env=env,
                                                                        first_index=first).items()})
    for k in KEYS:
        cases.assert_tight(cases.summarize(cases.rel_err(unlit[k], dark[k])), ("unlit", k))
    for g in (4, 16):
        alt = _with_group(g, lambda: {k: host(v) for k, v in sk.integrate(dev(P), scene, spp_n, seed, env=env,
                                                                          first_index=first, lights=lg).items()})
        for k in KEYS:
            cases.assert_same_bits(alt[k], got[k], (g, k))    # sums in sample order whatever the group width
