"""Oracle checks for the light loops added around the rlGgx one: several lights (`while (AiLightsGetSample(sg))`), the
rlDisney loop (src/rlDisney.cpp:695-705) and the two rlSkin loops (src/rlSkin.cpp:193-198, 217-222), all with the
documented stand-ins for the closed light services (parity unpinned).  Anchors: a closed form (the Disney diffuse lobe
with roughness 0.5 at normal incidence is Lambert's), MIS consistency of each Disney lobe's sample / eval / pdf triple,
linearity in the light set, and the sample-count bookkeeping of getAvgReflectWeight (src/rlGgx.h:103,181-184)."""
import math

import numpy as np
import pytest

import cases
import oracle_lib as O
from gpu_util_cpu import disney_oracle, ggx_oracle


def _points(n, wo, rng=None):
    N = np.zeros((3, n), np.float32); N[2] = 1
    T = np.zeros((3, n), np.float32); T[0] = 1
    if rng is not None:
        a = rng.uniform(0, 2 * math.pi, n)
        T[0], T[1] = np.cos(a), np.sin(a)
    w = np.asarray(wo, np.float64); w /= np.linalg.norm(w)
    WO = np.repeat(w.astype(np.float32)[:, None], n, axis=1)
    return np.ascontiguousarray(WO), N, T, np.zeros((3, n), np.float32)


def test_disney_diffuse_small_distant_light_closed_form():
    """view and light both at the normal: F_L = F_V = 0, so the diffuse lobe is baseColor / pi x (1 - metallic) x N.L
    and the direct diffuse AOV of a small light is radiance x that x its solid angle"""
    n = 256
    wo, N, T, P = _points(n, (0.0, 0.0, 1.0))
    d = O.Disney(wo, N, T, base_color=(0.8, 0.6, 0.4), nthreads=4, roughness=0.5, metallic=0.25, subsurface=0.0,
                 specular=0.0, sheen=0.0, clearcoat=0.0)
    R, dist = 0.05, 13.0
    lt = O.make_light(center=(0.0, 0.0, dist), radius=R, radiance=(2.0, 1.0, 0.5), mis_mode=1)
    dd, ds = d.direct_lighting(P, lt, 4, 1)
    omega = 2 * math.pi * (1 - math.sqrt(1 - (R / dist) ** 2))
    want = [rad * bc / math.pi * 0.75 * omega for rad, bc in zip((2.0, 1.0, 0.5), (0.8, 0.6, 0.4))]
    np.testing.assert_allclose(dd.astype(np.float64).mean(axis=1), want, rtol=2e-3)
    assert np.isfinite(ds).all() and (ds >= 0).all()


@pytest.mark.parametrize("rough,radius", [(0.2, 1.5), (0.5, 0.8)])
def test_mis_consistency_of_both_disney_triples(rough, radius):
    """clearcoat = 0: with a clearcoat the reference's specular triple is NOT self-consistent (next test)"""
    rng = np.random.default_rng(5)
    n = 8192
    wo, N, T, P = _points(n, (0.5, 0.0, 0.85), rng)
    d = O.Disney(wo, N, T, base_color=(0.8, 0.6, 0.4), nthreads=O.hardware_threads(), roughness=rough, metallic=0.3,
                 specular=0.5, clearcoat=0.0, clearcoat_gloss=0.6, sheen=0.3, anisotropic=0.3, subsurface=0.2)
    out = {}
    for mode in (0, 1, 2):
        lt = O.make_light(center=(-2.0, 0.3, 3.5), radius=radius, mis_mode=mode)
        dd, ds = d.direct_lighting(P, lt, 4, 9)
        assert np.isfinite(dd).all() and np.isfinite(ds).all() and (dd >= 0).all() and (ds >= 0).all()
        out[mode] = (dd.astype(np.float64).mean(axis=1), ds.astype(np.float64).mean(axis=1))
    for mode in (1, 2):
        np.testing.assert_allclose(out[mode][0], out[0][0], rtol=0.03)      # diffuse lobe (cosine sampling)
        np.testing.assert_allclose(out[mode][1], out[0][1], rtol=0.05)      # specular lobe (VNDF + GTR1 sampling)
    assert out[0][0].min() > 0 and out[0][1].min() > 0


def test_clearcoat_sampler_and_pdf_disagree_as_in_the_reference():
    """A property of the reference, kept: sampleGTR1Direction draws the clearcoat half vector with a2 = roughness^2
    (src/rlDisney.cpp:393-404) while evalSpecularPdf's D_GTR1 uses the clearcoat-gloss alpha (545-551), so with
    clearcoat > 0 the BSDF-sampling estimate of the specular lobe differs from the light-sampling one (the light-
    sampling estimate does not use the sampler and stays the reference integral of evalBrdf)."""
    rng = np.random.default_rng(5)
    n = 8192
    wo, N, T, P = _points(n, (0.5, 0.0, 0.85), rng)
    d = O.Disney(wo, N, T, base_color=(0.8, 0.6, 0.4), nthreads=O.hardware_threads(), roughness=0.5, metallic=0.3,
                 specular=0.5, clearcoat=1.0, clearcoat_gloss=0.6, sheen=0.0, anisotropic=0.3, subsurface=0.2)
    est = {}
    for mode in (1, 2):
        lt = O.make_light(center=(-2.0, 0.3, 3.5), radius=0.8, mis_mode=mode)
        est[mode] = d.direct_lighting(P, lt, 4, 9)[1].astype(np.float64).mean(axis=1)
    assert np.all(est[2] < 0.95 * est[1])


def test_light_sets_add_up():
    """the AOVs of a light set are the per-light AOVs added in array order; light 0 of a set draws the numbers it draws
    alone; a light below every horizon adds exactly nothing"""
    n, spp_n, seed = 2048, 3, 21
    P = np.stack([O.gen_uniform(seed, 0, n, 40 + j, 0.0, 4.0) for j in range(3)])
    A = O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0))
    B = O.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0), mis_mode=1)
    for make, case in ((ggx_oracle, cases.ggx_mixed(cases.SEED_PARITY, n)), (disney_oracle, cases.disney_mixed(cases.SEED_PARITY, n))):
        o = make(O, case)
        a = o.direct_lighting(P, A, spp_n, seed)
        ab = o.direct_lighting(P, [A, B], spp_n, seed)
        ba = o.direct_lighting(P, [B, A], spp_n, seed)
        for j in range(2):
            assert np.array_equal(o.direct_lighting(P, [A], spp_n, seed)[j].view(np.uint32), a[j].view(np.uint32))
            assert np.all(ab[j] >= a[j])
            # the same two lights in the other order: other sample streams, the same integral
            np.testing.assert_allclose(ab[j].astype(np.float64).mean(axis=1), ba[j].astype(np.float64).mean(axis=1), rtol=0.15)
    far = np.zeros((3, n), np.float32)
    N = np.zeros((3, n), np.float32); N[2] = 1
    T = np.zeros((3, n), np.float32); T[0] = 1
    wo = np.ascontiguousarray(np.repeat(np.array([[0.3], [0.0], [0.954]], np.float32), n, axis=1))
    g = O.Ggx(wo, N, T, roughness=0.4, ior=1.5)
    under = O.make_light(center=(0.0, 0.0, -6.0), radius=1.0, radiance=(9.0, 9.0, 9.0))
    a = g.direct_lighting(far, A, spp_n, seed)
    au = g.direct_lighting(far, [A, under], spp_n, seed)
    for j in range(2):
        assert np.array_equal(a[j].view(np.uint32), au[j].view(np.uint32))


def test_skin_light_loops_feed_the_fresnel_mean():
    """getAvgReflectWeight = (Fresnel terms of the light loops' BSDF samples + integrateGlossy's) / (their count):
    - light-sampling-only lights draw no BSDF samples: the hand-down scalars are those without lights, the AOVs grow;
    - a black lobe colour stops integrateGlossy from sampling (src/rlGgx.h:174-176) but not the light loop
      (167-170): without lights the mean is 1, with one BSDF-sampling light it is that loop's mean;
    - a light the shading point is inside of draws no samples at all"""
    n, spp_n, seed = 1024, 3, 4
    c = cases.skin_mixed(cases.SEED_PARITY, n)
    p = dict(c["params"])
    black = (np.arange(n) % 4) == 0
    p["specular_color"] = np.where(black[None, :], np.float32(0.0), p["specular_color"]).astype(np.float32)
    sc = O.make_scene(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    run = lambda lights: O.skin_integrate(c["wo"], c["N"], c["T"], p, c["N"], sc, spp_n, seed, nthreads=4, lights=lights)
    dark = run(None)
    light_only = run([O.make_light(center=(0.5, 0.5, 4.0), radius=1.0, radiance=(2.0, 1.5, 1.0), mis_mode=1)])
    for k in ("sheenFresnel", "specularFresnel", "sssWeight", "sss"):
        assert np.array_equal(light_only[k].view(np.uint32), dark[k].view(np.uint32)), k
    assert (light_only["sheen"].sum(0) > dark["sheen"].sum(0)).mean() > 0.2
    both = run([O.make_light(center=(0.5, 0.5, 4.0), radius=1.0, radiance=(2.0, 1.5, 1.0), mis_mode=0)])
    on = black & (p["specular_weight"] > 1e-4)
    assert np.array_equal(dark["specularFresnel"][on], p["specular_weight"][on])                  # mean = 1
    assert (both["specularFresnel"][on] < p["specular_weight"][on]).mean() > 0.9                  # mean = the loop's
    assert (both["specularFresnel"] != dark["specularFresnel"]).mean() > 0.9
    assert (both["specular"][:, on] == 0).all()                                                   # black colour: no light either
    inside = run([O.make_light(center=(0.0, 0.0, 0.0), radius=5.0, radiance=(7.0, 7.0, 7.0), mis_mode=0)])
    for k in dark:
        assert np.array_equal(inside[k].view(np.uint32), dark[k].view(np.uint32)), k
    for r in (dark, light_only, both):
        for k in r:
            assert np.isfinite(r[k]).all(), k


def test_whole_node_shading_closed_forms():
    """orc_batch_ggx_shade / orc_batch_disney_shade (shader_evaluate of the two nodes, src/rlGgx.cpp:248-327,
    src/rlDisney.cpp:685-727): a Lambertian diffuse closure under a uniform environment reflects diffuseColor x env
    (brdf x cos / pdf = 1 for every cosine-weighted sample); the untraced transmission is SQR(iorOut / iorIn) |N . dir|
    x env x KtColor x Kt; the output is the AOVs added in the reference's order; radiance scales linearly."""
    n, spp_n, seed = 512, 3, 8
    wo, N, T, P = _points(n, (0.3, 0.1, 0.9))
    g = O.Ggx(wo, N, T, KsColor=(0.9, 0.8, 0.7), roughness=0.4, ior=1.5, nthreads=4)
    A = O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0))
    env = (0.5, 0.25, 2.0)
    r = g.shade(P, [A], spp_n, seed, Kd_color=(0.8, 0.6, 0.4), Kd=0.5, Kd_roughness=0.0, Ks=0.5, Kt_color=(0.2, 0.4, 0.8), Kt=0.5,
                env=env, traced=False)
    for k in range(3):
        np.testing.assert_allclose(r["indirect_diffuse"][k], (0.8, 0.6, 0.4)[k] * 0.5 * env[k], rtol=2e-6)
    # untraced refraction about N: Snell, eta = 1 / 1.5 for an entering ray
    ci = float(wo[2, 0]); eta = 1.0 / 1.5
    ct = math.sqrt(1.0 - eta * eta * (1.0 - ci * ci))
    for k in range(3):
        np.testing.assert_allclose(r["refraction"][k], 1.5 ** 2 * ct * env[k] * (0.2, 0.4, 0.8)[k] * 0.5, rtol=1e-5)
    want = ((r["direct_diffuse"] + r["direct_specular"]) + r["refraction"]) + (r["indirect_diffuse"] + r["indirect_specular"])
    assert np.array_equal(want.view(np.uint32), r["out"].view(np.uint32))
    dd, ds = g.direct_lighting(P, [A], spp_n, seed, Kd_color=(0.8, 0.6, 0.4), Kd=0.5, Kd_roughness=0.0, Ks=0.5)
    assert np.array_equal(dd.view(np.uint32), r["direct_diffuse"].view(np.uint32))
    assert np.array_equal(ds.view(np.uint32), r["direct_specular"].view(np.uint32))
    r2 = g.shade(P, [A], spp_n, seed, Kd_color=(0.8, 0.6, 0.4), Kd=0.5, Kd_roughness=0.0, Ks=0.5, Kt_color=(0.2, 0.4, 0.8), Kt=0.5,
                 env=tuple(2 * e for e in env), traced=False)
    for q in ("refraction", "indirect_diffuse", "indirect_specular"):
        assert np.array_equal((2 * r[q]).view(np.uint32), r2[q].view(np.uint32)), q
    # Kd = 0: AiColorIsSmall(diffuseColor) switches both diffuse estimators off (280, 315)
    r0 = g.shade(P, [A], spp_n, seed, Kd=0.0, Kt=0.0, env=env)
    assert not r0["direct_diffuse"].any() and not r0["indirect_diffuse"].any() and not r0["refraction"].any()
    assert np.array_equal(r0["direct_specular"].view(np.uint32), r["direct_specular"].view(np.uint32))

    c = cases.disney_mixed(cases.SEED_PARITY, n)
    d = disney_oracle(O, c)
    Pd = np.stack([O.gen_uniform(seed, 0, n, 40 + j, 0.0, 4.0) for j in range(3)])
    s1 = d.shade(Pd, [A], spp_n, seed, env=(1.0, 1.0, 1.0))
    s2 = d.shade(Pd, [A], spp_n, seed, env=(2.0, 4.0, 0.5))
    for q in ("indirect_diffuse", "indirect_specular"):
        assert np.array_equal((s1[q] * np.array([[2.0], [4.0], [0.5]], np.float32)).view(np.uint32), s2[q].view(np.uint32)), q
        assert np.isfinite(s1[q]).all() and (s1[q] >= 0).all() and (s1[q] > 0).mean() > 0.5
    dd, ds = d.direct_lighting(Pd, [A], spp_n, seed)
    assert np.array_equal(dd.view(np.uint32), s1["direct_diffuse"].view(np.uint32))
    assert np.array_equal(ds.view(np.uint32), s1["direct_specular"].view(np.uint32))
    want = (s1["direct_diffuse"] + s1["direct_specular"]) + (s1["indirect_diffuse"] + s1["indirect_specular"])
    assert np.array_equal(want.view(np.uint32), s1["out"].view(np.uint32))
    # the indirect loops are the n^2-spp integrator's (other sample streams): the same integrals
    whole = d.integrate(8, seed)
    big = d.shade(Pd, None, 8, seed)
    np.testing.assert_allclose(big["indirect_diffuse"].astype(np.float64).mean(axis=1),
                               (whole["diffuse_sum"].astype(np.float64) / 64).mean(axis=1), rtol=0.02)
    np.testing.assert_allclose(big["indirect_specular"].astype(np.float64).mean(axis=1),
                               (whole["specular_sum"].astype(np.float64) / 64).mean(axis=1), rtol=0.1)


def test_the_two_summation_orders_of_the_estimator_agree():
    """The light loops' estimator is the builder's stand-in for the closed AiEvaluateLightSample loop, so its summation
    order is defined here, once: canonically ONE running sum per AOV (light sample's term, then the BSDF sample's, sample
    by sample).  The device kernels run the two strategies as separate passes and so produce the second form -- one sum
    per strategy, added at the end (orc_batch_*_two_sums) -- which the GPU tests match bit for bit.  Same terms, another
    order: the two forms must agree to 1e-6, for rlGgx and rlDisney, the light loops alone and the whole shader_evaluate,
    one light and three, every MIS mode; with one strategy switched off they are the same sum and must be bit-equal."""
    n = 1 << 13
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    dcase = cases.disney_mixed(cases.SEED_PARITY, n)
    P = np.stack([O.gen_uniform(cases.SEED_PARITY, 0, n, O.S_PARAM0 + 16 + j, 0.0, 4.0) for j in range(3)])
    th = O.hardware_threads()
    g, d = ggx_oracle(O, c, nthreads=th), disney_oracle(O, dcase, nthreads=th)
    assert g.two_sums_default is False and d.two_sums_default is False        # the canonical form is the default

    def worst(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        den = np.maximum(np.abs(b), 1e-30)
        return float((np.abs(a - b) / den).max())

    sets = {"one": [O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0))],
            "three": [O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
                      O.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0)),
                      O.make_light(center=(0.5, -3.0, 5.0), radius=2.0, radiance=(1.0, 1.0, 1.0))]}
    differing = 0
    for name, lts in sets.items():
        for spp_n in (3, 4):
            kw = dict(Kd_color=(0.8, 0.7, 0.6), Kd=0.5, Kd_roughness=0.3, Ks=0.5)
            a, b = g.direct_lighting(P, lts, spp_n, 321, **kw), g.direct_lighting(P, lts, spp_n, 321, two_sums=True, **kw)
            for x, y in zip(a, b):
                assert worst(y, x) <= 1e-6, (name, spp_n, "rlGgx light loop")
                differing += int((x.view(np.uint32) != y.view(np.uint32)).sum())
            a, b = d.direct_lighting(P, lts, spp_n, 17), d.direct_lighting(P, lts, spp_n, 17, two_sums=True)
            for x, y in zip(a, b):
                assert worst(y, x) <= 1e-6, (name, spp_n, "rlDisney light loop")
                differing += int((x.view(np.uint32) != y.view(np.uint32)).sum())
        a = g.shade(P, lts, 3, 5, Kt=0.5, env=(1.0, 0.9, 0.8))
        b = g.shade(P, lts, 3, 5, Kt=0.5, env=(1.0, 0.9, 0.8), two_sums=True)
        for k in a:
            assert worst(b[k], a[k]) <= 1e-6, (name, k, "rlGgx shader_evaluate")
        a, b = d.shade(P, lts, 3, 5), d.shade(P, lts, 3, 5, two_sums=True)
        for k in a:
            assert worst(b[k], a[k]) <= 1e-6, (name, k, "rlDisney shader_evaluate")
    assert differing > 0, "the two orders never differed: the second form is not being exercised"
    # one strategy only: a single sum either way
    for mode in (1, 2):
        lt = O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0), mis_mode=mode)
        for x, y in zip(g.direct_lighting(P, lt, 4, 321), g.direct_lighting(P, lt, 4, 321, two_sums=True)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        for x, y in zip(d.direct_lighting(P, lt, 4, 17), d.direct_lighting(P, lt, 4, 17, two_sums=True)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
