"""GPU: RLS_MATH_FAST (hardware rcp/sqrt/sin/cos/exp arithmetic, algebraic view analysis) against the
oracle.  FAST is opt-in; it is held to the north-star tolerance statistically, with the same
conditioning-aware protocol SURVEY.md 8(c) prescribes: medians at round-off, a bounded fraction beyond
1e-5, and the outliers on points where the oracle itself moves inside the 1-ulp box of its inputs (tests/conditioning.py)."""
import numpy as np
import pytest

import cases
import rlshaders_amd as R
from gpu_util import dev, disney_oracle, disney_sampler, ggx_oracle, ggx_sampler, host

pytestmark = pytest.mark.gpu

N = 1 << 16
TOL = 1e-5


@pytest.fixture(scope="module")
def fast():
    ctx = R.Context(0)
    ctx.set_math_mode(True)
    assert ctx.lib.rls_context_get_math_mode(ctx.handle) == 1
    yield ctx
    ctx.close()


def test_mode_argument_checked(fast):
    assert fast.lib.rls_context_set_math_mode(fast.handle, 7) == 1
    assert fast.lib.rls_context_get_math_mode(fast.handle) == 1


def test_ggx_eval_pdf_decoupled(fast, oracle):
    c = cases.ggx_mixed(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 2)
    og = ggx_oracle(oracle, c)
    wi, f_ref, pdf_ref, _ = og.sample_eval_pdf(x[0], x[1])
    s = ggx_sampler(fast, c)
    ef = cases.summarize(cases.rel_err(host(s.evalBrdf(dev(wi))), f_ref))
    ep = cases.summarize(cases.rel_err(host(s.evalPdf(dev(wi))), pdf_ref))
    print("fast ggx eval decoupled", ef)
    print("fast ggx pdf  decoupled", ep)
    # ~20 one-ulp operations per value, no cancellation: everything within the tolerance
    assert ef["nonfinite"] == 0 and ep["nonfinite"] == 0
    assert ef["p99"] <= TOL and ep["p99"] <= TOL
    assert ef["p999"] <= 3e-5 and ep["p999"] <= 3e-5
    assert ef["max"] <= 2e-4 and ep["max"] <= 2e-4


def test_ggx_chain_conditioning(fast, oracle):
    """SURVEY.md 8(c) protocol (3) on config 2's kernel: a FAST output may be beyond 1e-5 only where the oracle's own output
    moves inside the 1-ulp box of its 19 inputs (tests/conditioning.py: 38 axis nudges + 64 random corners), and then by no
    more than 8 x that movement -- the factor the 2^24-point measurement supports (profiles/r04_fast_conditioning.json:
    beyond 8 x on 1.2e-5 ... 4.2e-5 of the points per output, beyond 1024 x on 0 ... 4 points of 2^24 over three seeds -- no
    branch flips since cos(theta') of the stretched view is formed by the reference's own sequence; 98.8 ... 99.9 % of the outliers have the oracle moving by > 1e-5 / 4 itself)."""
    import conditioning as Q
    c = cases.ggx_mixed(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 4)
    mk = lambda cc: ggx_oracle(oracle, cc)
    ref = mk(c).reflect_refract(x[0], x[1], x[2], x[3])
    s = ggx_sampler(fast, c)
    got = [host(t) for t in s.reflectRefract(dev(x[0]), dev(x[1]), dev(x[2]), dev(x[3]))]
    err = Q.chain_errors(got, ref)
    out = np.zeros(N, bool)
    for e in err:
        out |= ~(e <= TOL)
    idx = np.union1d(np.nonzero(out)[0], np.arange(8))
    assert idx.size <= 0.05 * N                                   # ~2.7 % of the points have some output beyond 1e-5
    cs, xs = Q.subset(c, x, idx)
    sens, _ = Q.ggx_chain_sensitivity(mk, cs, xs, [np.ascontiguousarray(r[..., idx]) for r in ref], corners=64, seed=1)
    for k, nm in enumerate(Q.NAMES):
        st = cases.summarize(err[k])
        print("fast ggx chain", nm, st)
        assert st["nonfinite"] == 0
        assert st["median"] <= 2e-6, (nm, st)
        assert st["frac_gt_1e5"] <= 5e-2 and st["p99"] <= 1e-4 and st["p999"] <= 1e-3, (nm, st)
        e = err[k][idx]
        beyond8 = int((e > np.maximum(TOL, 8.0 * sens[k])).sum())
        beyond1024 = int((e > np.maximum(TOL, 1024.0 * sens[k])).sum())
        unexplained = int((~(e <= TOL) & (sens[k] < TOL / 4)).sum())
        print("   outliers", int((~(e <= TOL)).sum()), "beyond 8 x sens", beyond8, "beyond 1024 x", beyond1024, "unexplained", unexplained)
        # 2^24 points: <= 4.2e-5 of the points beyond 8 x; here 2^16 points, so a handful at most
        assert beyond8 <= 2e-4 * N, (nm, beyond8)
        assert beyond1024 == 0, (nm, beyond1024)
        assert unexplained <= 2e-4 * N, (nm, unexplained)
    # EXACT stays the default
    ex = R.Context(0)
    assert ex.lib.rls_context_get_math_mode(ex.handle) == 0
    ex.close()


@pytest.mark.parametrize("lobe", [R.RLS_RAY_DIFFUSE, R.RLS_RAY_GLOSSY])
def test_disney(fast, oracle, lobe):
    c = cases.disney_mixed(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 2)
    od = disney_oracle(oracle, c)
    ref = od.sample_eval_pdf(lobe, x[0], x[1])
    s = disney_sampler(fast, c)
    s.setSampleType(lobe)
    got = [host(t) for t in s.sampleEvalPdf(dev(x[0]), dev(x[1]))]
    z_ref = (ref[0] == 0).all(axis=0)
    z_got = (got[0] == 0).all(axis=0)
    assert (z_ref != z_got).mean() <= 1e-3
    both = ~z_ref & ~z_got
    for k, nm in enumerate(("wi", "f", "pdf")):
        st = cases.summarize(cases.rel_err(got[k][..., both], ref[k][..., both]))
        print("fast disney", lobe, nm, st)
        assert st["nonfinite"] == 0 and st["median"] <= 3e-6 and st["frac_gt_1e5"] <= 6e-2, (nm, st)
        assert st["p99"] <= 1e-4, (nm, st)
    f = host(s.evalBrdf(dev(ref[0])))
    st = cases.summarize(cases.rel_err(f, ref[1]))
    assert st["p99"] <= TOL and st["p999"] <= 1e-4, st


def test_sss_and_skin(fast, oracle):
    c = cases.sss_mixed(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 2)
    o = oracle.Sss(N, c["dist"], c["albedo"], N=c["N"], T=c["T"], has_dPdu=True)
    ref = o.probe(x[0], x[1])
    s = R.SssSampler(fast, dev(c["N"]), dev(c["T"]), dev(c["albedo"]), dev(c["dist"]))
    got = {k: host(v) for k, v in s.getProbeRay(dev(x[0]), dev(x[1])).items()}
    for k in ("r", "origin", "dir", "maxdist", "pdf", "profile"):
        st = cases.summarize(cases.rel_err(got[k], ref[k]))
        print("fast probe", k, st)
        assert st["nonfinite"] == 0 and st["median"] <= 3e-6 and st["frac_gt_1e5"] <= 5e-2, (k, st)
    ck = cases.skin_mixed(cases.SEED_PARITY, N)
    xi = cases.xi(cases.SEED_PARITY, N, 6)
    refk = oracle.skin(ck["wo"], ck["N"], ck["T"], ck["params"], xi, nthreads=4)
    sk = R.SkinShader(fast, dev(ck["wo"]), dev(ck["N"]), dev(ck["T"]), **{k: dev(v) for k, v in ck["params"].items()})
    gotk = {k: host(v) for k, v in sk.sampleEvalPdf(dev(xi)).items()}
    for k in ("sheen_wi", "sheen_f", "spec_f", "spec_pdf", "r", "r_pdf", "profile", "sssWeight"):
        st = cases.summarize(cases.rel_err(gotk[k], refk[k]))
        print("fast skin", k, st)
        assert st["nonfinite"] == 0 and st["median"] <= 3e-6 and st["frac_gt_1e5"] <= 5e-2, (k, st)


def test_integrators_in_fast_mode(fast, oracle):
    """the round-2 entry points run in FAST mode too (rlDisney n^2-spp loop incl. chunked streaming, rlSkin shader_evaluate,
    integrateRefract): Monte-Carlo sums of many samples, held to the sums' own scale rather than per point"""
    n, spp_n, seed = 1 << 12, 4, 17
    d = cases.disney_mixed(cases.SEED_PARITY, n)
    ref = disney_oracle(oracle, d).integrate(spp_n, seed)
    ds = disney_sampler(fast, d)
    got = {k: host(v) for k, v in ds.integrate(spp_n, seed).items()}
    sums, _ = ds.integrateChunked(spp_n, seed, 1000)
    for k in ("diffuse_sum", "specular_sum"):
        a, b = got[k].astype(np.float64), ref[k].astype(np.float64)
        assert np.isfinite(a).all()
        assert abs(a.mean() / b.mean() - 1) < 2e-3, (k, a.mean(), b.mean())
        assert np.quantile(cases.rel_err(got[k], ref[k]), 0.9) <= 1e-3, k
        assert np.allclose(host(sums[k]), got[k], rtol=1e-4, atol=1e-6), k            # chunked == unchunked, same mode
    for k in ("diffuse_count", "specular_count"):
        assert np.mean(got[k] != ref[k]) < 0.02, k
    sk = cases.skin_mixed(cases.SEED_PARITY, n)
    kw = dict(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    refk = oracle.skin_integrate(sk["wo"], sk["N"], sk["T"], sk["params"], sk["N"], oracle.make_scene(**kw), spp_n, seed, nthreads=4)
    shader = R.SkinShader(fast, dev(sk["wo"]), dev(sk["N"]), dev(sk["T"]), **{k: dev(v) for k, v in sk["params"].items()})
    gotk = {k: host(v) for k, v in shader.integrate(dev(sk["N"]), R.make_scene(**kw), spp_n, seed).items()}
    for k in ("sheenFresnel", "specularFresnel", "sssWeight"):
        st = cases.summarize(cases.rel_err(gotk[k], refk[k]))
        assert st["nonfinite"] == 0 and st["median"] <= 1e-5 and st["p99"] <= 1e-3, (k, st)
    for k in ("sheen", "specular", "sss", "out"):
        a, b = gotk[k].astype(np.float64), refk[k].astype(np.float64)
        assert np.isfinite(a).all() and abs(a.mean() / b.mean() - 1) < 5e-3, (k, a.mean(), b.mean())
    g = cases.ggx_mixed(cases.SEED_PARITY, n)
    res, tir = ggx_oracle(oracle, g).integrate_refract(spp_n, seed)
    gr, gt = [host(t) for t in ggx_sampler(fast, g).integrateRefract(spp_n, seed, want_tir=True)]
    assert np.array_equal(gt, tir)
    assert abs(gr.astype(np.float64).mean() / res.astype(np.float64).mean() - 1) < 2e-3


def test_light_loops_in_fast_mode(fast, oracle):
    """rls_disney_direct_lighting and rls_skin_integrate's light loops in FAST mode: Monte-Carlo sums in which a BSDF
    sample grazing a light's rim may flip in or out, so the AOVs are held at the batch's scale and per point in bulk"""
    n, spp_n, seed = 1 << 12, 4, 23
    P = (cases.xi(cases.SEED_PARITY, n, 3) * np.array([[4.0], [4.0], [1.0]], np.float32)).astype(np.float32)
    specs = (dict(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
             dict(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=1))
    lo = [oracle.make_light(**s) for s in specs]
    lg = [R._capi.SphereLight.from_buffer_copy(bytes(l)) for l in lo]
    d = cases.disney_mixed(cases.SEED_PARITY, n)
    ref = disney_oracle(oracle, d).direct_lighting(P, lo, spp_n, seed)
    got = [host(t) for t in disney_sampler(fast, d).directLighting(dev(P), lg, spp_n, seed)]
    for nm, a, b in zip(("diffuse", "specular"), got, ref):
        assert np.isfinite(a).all(), nm
        assert abs(a.astype(np.float64).mean() / b.astype(np.float64).mean() - 1) < 5e-3, nm
        assert np.quantile(cases.rel_err(a, b), 0.9) <= 1e-3, nm
    sk = cases.skin_mixed(cases.SEED_PARITY, n)
    kw = dict(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    skl = [oracle.make_light(center=(0.5, 0.5, 4.0), radius=1.0, radiance=(2.0, 1.5, 1.0))]
    skg = [R._capi.SphereLight.from_buffer_copy(bytes(l)) for l in skl]
    refk = oracle.skin_integrate(sk["wo"], sk["N"], sk["T"], sk["params"], sk["N"], oracle.make_scene(**kw), spp_n, seed,
                                 nthreads=4, lights=skl)
    shader = R.SkinShader(fast, dev(sk["wo"]), dev(sk["N"]), dev(sk["T"]), **{k: dev(v) for k, v in sk["params"].items()})
    gotk = {k: host(v) for k, v in shader.integrate(dev(sk["N"]), R.make_scene(**kw), spp_n, seed, lights=skg).items()}
    for k in ("sheenFresnel", "specularFresnel", "sssWeight"):
        st = cases.summarize(cases.rel_err(gotk[k], refk[k]))
        assert st["nonfinite"] == 0 and st["median"] <= 1e-5 and st["p99"] <= 1e-3, (k, st)
    for k in ("sheen", "specular", "sss", "out"):
        a, b = gotk[k].astype(np.float64), refk[k].astype(np.float64)
        assert np.isfinite(a).all() and abs(a.mean() / b.mean() - 1) < 5e-3, (k, a.mean(), b.mean())


def test_whole_node_shading_in_fast_mode(fast, oracle):
    """rls_ggx_shade / rls_disney_shade in FAST mode: every AOV at the batch's scale and per point in bulk"""
    n, spp_n, seed = 1 << 12, 4, 29
    P = (cases.xi(cases.SEED_PARITY, n, 3) * np.array([[4.0], [4.0], [1.0]], np.float32)).astype(np.float32)
    lo = [oracle.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
          oracle.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=1)]
    lg = [R._capi.SphereLight.from_buffer_copy(bytes(l)) for l in lo]
    g = cases.ggx_mixed(cases.SEED_PARITY, n)
    kw = dict(Kd=0.6, Ks=0.4, Kt=0.5)
    ref = ggx_oracle(oracle, g).shade(P, lo, spp_n, seed, Kd_color=(0.9, 0.5, 0.3), Kd_roughness=0.4, Kt_color=(0.2, 0.9, 0.9), **kw)
    got = {k: host(v) for k, v in ggx_sampler(fast, g).shade(dev(P), lg, spp_n, seed, KdColor=(0.9, 0.5, 0.3), diffuseRoughness=0.4,
                                                            KtColor=(0.2, 0.9, 0.9), **kw).items()}
    d = cases.disney_mixed(cases.SEED_PARITY, n)
    refd = disney_oracle(oracle, d).shade(P, lo, spp_n, seed)
    gotd = {k: host(v) for k, v in disney_sampler(fast, d).shade(dev(P), lg, spp_n, seed).items()}
    for name, r, q in (("ggx", ref, got), ("disney", refd, gotd)):
        for k in r:
            a, b = q[k].astype(np.float64), r[k].astype(np.float64)
            assert np.isfinite(a).all(), (name, k)
            assert abs(a.mean() / b.mean() - 1) < 5e-3, (name, k, a.mean(), b.mean())
            assert np.quantile(cases.rel_err(q[k], r[k]), 0.9) <= 2e-3, (name, k)


def test_fast_mode_specialisations_agree(fast):
    """FAST mode has the UNIFORM_* and by-reference specialisations too (the same templates compiled with RLS_FAST = 1): each
    must give what FAST's streamed kernels give on the same values as per-point planes -- same arithmetic, so the same bits."""
    import torch
    n, m = 1 << 14, 11
    wo, Nn, T = (dev(a) for a in cases.frame(cases.SEED_PARITY, n))
    xi = cases.xi(cases.SEED_PARITY, n, 6)
    dx = [dev(a) for a in xi]
    full = lambda v: dev(np.repeat(np.asarray(v, np.float32)[:, None], n, axis=1) if np.ndim(v) else np.full(n, v, np.float32))

    def same(a, b, what):
        a, b = host(a), host(b)
        ok = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert ok.all(), (what, int((~ok).sum()))

    # uniform against planes
    g = dict(specColor=(0.9, 0.8, 0.7), ior=1.5, roughness=0.35, anisotropic=0.25)
    a = R.GgxSampler(fast, wo, Nn, T, **g).reflectRefract(*dx[:4])
    b = R.GgxSampler(fast, wo, Nn, T, **{k: full(v) for k, v in g.items()}).reflectRefract(*dx[:4])
    for k, (p, q) in enumerate(zip(a, b)):
        same(p, q, ("ggx uniform", k))
    d = dict(base_color=(0.85, 0.7, 0.2), subsurface=0.2, metallic=0.3, specular=0.5, specular_tint=0.25, roughness=0.4,
             anisotropic=0.4, sheen=0.5, sheen_tint=0.5, clearcoat=0.6, clearcoat_gloss=0.7)
    for lobe in (R.RLS_RAY_DIFFUSE, R.RLS_RAY_GLOSSY):
        du, dp = R.DisneySampler(fast, wo, Nn, T, **d), R.DisneySampler(fast, wo, Nn, T, **{k: full(v) for k, v in d.items()})
        dc = R.DisneySampler(fast, wo, Nn, T, **dict(d, base_color=full(d["base_color"])))          # the colour-map kernel
        for s in (du, dp, dc):
            s.setSampleType(lobe)
        for k, (p, q, r) in enumerate(zip(du.sampleEvalPdf(dx[0], dx[1]), dp.sampleEvalPdf(dx[0], dx[1]), dc.sampleEvalPdf(dx[0], dx[1]))):
            same(p, q, ("disney uniform", lobe, k))
            same(r, q, ("disney colour map", lobe, k))
    sk = dict(cases.SKIN_DEFAULTS, sheen_weight=0.3, sss_scatter_dist=(1.0, 0.6, 0.35))
    a = R.SkinShader(fast, wo, Nn, T, **sk).sampleEvalPdf(dev(xi))
    b = R.SkinShader(fast, wo, Nn, T, **{k: full(v) for k, v in sk.items()}).sampleEvalPdf(dev(xi))
    for k in a:
        same(a[k], b[k], ("skin uniform", k))
    a = R.SssSampler(fast, Nn, T, (0.8, 0.5, 0.4), (1.0, 0.6, 0.35)).getProbeRay(dx[0], dx[1])
    b = R.SssSampler(fast, Nn, T, (0.8, 0.5, 0.4), full((1.0, 0.6, 0.35))).getProbeRay(dx[0], dx[1])
    for k in a:
        same(a[k], b[k], ("sss uniform", k))
    # by reference against planes
    ids = np.random.default_rng(2).integers(0, m, n).astype(np.int32)
    t = cases.ggx_mixed(cases.SEED_EDGE, m)
    cols = dict(specColor=t["KsColor"], ior=t["ior"], roughness=t["roughness"], anisotropic=t["anisotropic"])
    a = R.GgxSampler(fast, wo, Nn, T, materials=(dev(ids), m), **{k: dev(v) for k, v in cols.items()}).reflectRefract(*dx[:4])
    b = R.GgxSampler(fast, wo, Nn, T, **{k: dev(np.ascontiguousarray(v[..., ids])) for k, v in cols.items()}).reflectRefract(*dx[:4])
    for k, (p, q) in enumerate(zip(a, b)):
        same(p, q, ("ggx by reference", k))
    ts = cases.skin_mixed(cases.SEED_EDGE, m)["params"]
    a = R.SkinShader(fast, wo, Nn, T, materials=(dev(ids), m), **{k: dev(v) for k, v in ts.items()}).sampleEvalPdf(dev(xi))
    b = R.SkinShader(fast, wo, Nn, T, **{k: dev(np.ascontiguousarray(v[..., ids])) for k, v in ts.items()}).sampleEvalPdf(dev(xi))
    for k in a:
        same(a[k], b[k], ("skin by reference", k))
