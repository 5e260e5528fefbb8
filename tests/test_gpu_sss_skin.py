"""GPU parity of the rlSss (NDProfile, SssSampler hot parts) and rlSkin kernels against the oracle."""
import numpy as np
import pytest

import cases
import rlshaders_amd as R
from gpu_util import dev, host

pytestmark = pytest.mark.gpu

N = 1 << 16
TOL = 1e-5


@pytest.fixture(scope="module")
def mixed(oracle):
    return cases.sss_mixed(cases.SEED_PARITY, N), cases.xi(cases.SEED_PARITY, N, 2)


def test_nd_profile_sample_pdf_eval(gpu, oracle, mixed):
    c, x = mixed
    o = oracle.Sss(N, c["dist"], c["albedo"])
    r_ref, pdf_ref, prof_ref = o.nd_sample(x[0])
    p = R.NDProfile(gpu, N, dev(c["dist"]), dev(c["albedo"]))
    r, pdf, prof = (host(t) for t in p.sample(dev(x[0])))
    for nm, a, b in (("r", r, r_ref), ("pdf", pdf, pdf_ref), ("profile", prof, prof_ref)):
        st = cases.summarize(cases.rel_err(a, b))
        print("nd", nm, st)
        cases.assert_tight(st, nm)
    # decoupled on the oracle's r: only expf differs between the two libms
    st = cases.summarize(cases.rel_err(host(p.getPdf(dev(r_ref))), o.nd_pdf(r_ref)))
    sp = cases.summarize(cases.rel_err(host(p.evalProfile(dev(r_ref))), o.nd_profile(r_ref)))
    print("nd pdf decoupled", st)
    print("nd profile decoupled", sp)
    assert st["max"] <= TOL and sp["max"] <= TOL


def test_nd_uniform_kat(gpu, oracle):
    """the SURVEY 8(c) probe configuration: d = (1, .5, .25), uniform over the batch"""
    n = 4096
    rx = np.linspace(0, 1, n, endpoint=False).astype(np.float32)
    o = oracle.Sss(n, (1.0, 0.5, 0.25), (1.0, 0.84235, 0.5))
    p = R.NDProfile(gpu, n, (1.0, 0.5, 0.25), (1.0, 0.84235, 0.5))
    ref = o.nd_sample(rx)
    got = [host(t) for t in p.sample(dev(rx))]
    for a, b in zip(got, ref):
        fin = np.isfinite(b) if b.ndim == 1 else np.isfinite(b).all(axis=0)
        assert cases.summarize(cases.rel_err(a[..., fin], b[..., fin]))["p99"] <= 1e-5


def test_nd_degenerate_distances(gpu, oracle):
    """maxRadius < 1e-4 -> r = 0, pdf = 1, black profile; one tiny channel -> that channel 1, r = 0
    when it is selected (src/rlSss.cpp:38,46,70,88-91,101)"""
    n = 4096
    rx = cases.xi(cases.SEED_EDGE, n, 1)[0]
    for dist in ((0.0, 0.0, 0.0), (1e-5, 1e-5, 1e-5), (1.0, 1e-5, 0.5)):
        o = oracle.Sss(n, dist)
        ref = o.nd_sample(rx)
        got = [host(t) for t in R.NDProfile(gpu, n, dist).sample(dev(rx))]
        for a, b in zip(got, ref):
            both = np.isfinite(a) & np.isfinite(b)
            assert (np.isfinite(a) == np.isfinite(b)).all()
            assert cases.summarize(cases.rel_err(a[both], b[both]))["p99"] <= 1e-5, dist


@pytest.mark.parametrize("has_dPdu", [True, False])
def test_probe_ray(gpu, oracle, mixed, has_dPdu):
    c, x = mixed
    # dPdu: scaled, not orthogonal to N -> exercises the Gram-Schmidt frame (src/rlSss.h:151-154)
    T = (c["T"] * 2.5 + 0.3 * c["N"]).astype(np.float32) if has_dPdu else c["T"]
    o = oracle.Sss(N, c["dist"], c["albedo"], N=c["N"], T=T, has_dPdu=has_dPdu)
    ref = o.probe(x[0], x[1])
    s = R.SssSampler(gpu, dev(c["N"]), dev(T), dev(c["albedo"]), dev(c["dist"]), has_dPdu=has_dPdu)
    got = {k: host(v) for k, v in s.getProbeRay(dev(x[0]), dev(x[1])).items()}
    for k in ("r", "origin", "dir", "maxdist", "pdf", "profile"):
        st = cases.summarize(cases.rel_err(got[k], ref[k]))
        print("probe", has_dPdu, k, st)
        cases.assert_tight(st, k)
    assert np.array_equal(got["dir"], ref["dir"]) or cases.rel_err(got["dir"], ref["dir"]).max() <= 1e-6
    # with P the origin is P + offset
    P = cases.xi(cases.SEED_EDGE, N, 3)
    got2 = host(s.getProbeRay(dev(x[0]), dev(x[1]), P=dev(P))["origin"])
    assert np.allclose(got2, P + got["origin"], rtol=0, atol=1e-6)


def test_mis_pdf_and_cavity_fade(gpu, oracle, mixed):
    c, x = mixed
    o = oracle.Sss(N, c["dist"], c["albedo"], N=c["N"], T=c["T"], has_dPdu=True)
    s = R.SssSampler(gpu, dev(c["N"]), dev(c["T"]), dev(c["albedo"]), dev(c["dist"]))
    # displacement of a probe hit: a point near the surface within maxR; normal of the hit
    disp = (cases.xi(cases.SEED_EDGE, N, 3) - 0.5).astype(np.float32)
    sN, _, _ = cases.frame(cases.SEED_EDGE, N)
    for literal in (False, True):
        st = cases.summarize(cases.rel_err(host(s.misPdf(dev(disp), dev(sN), literal)), o.mis_pdf(disp, sN, literal)))
        print("mis pdf literal" if literal else "mis pdf projection", st)
        cases.assert_tight(st, "mis pdf")
    fade = host(R.SssSampler.cavityFade(gpu, dev(disp), dev(sN), dev(c["N"])))
    assert cases.summarize(cases.rel_err(fade, oracle.cavity_fade(disp, sN, c["N"])))["max"] <= TOL
    wi = host(R.SssSampler.sampleDiffuseDirection(gpu, dev(x[0]), dev(x[1]), dev(c["N"]), dev(c["T"])))
    st = cases.summarize(cases.rel_err(wi, oracle.sample_diffuse_direction(c["N"], c["T"], x[0], x[1])))
    print("sss diffuse direction", st)
    cases.assert_tight(st, "sss diffuse direction")


def test_util_directions(gpu, oracle, mixed):
    _, x = mixed
    sph, disk = (host(t) for t in R.util_directions(gpu, dev(x[0]), dev(x[1])))
    rs, rd = oracle.util_directions(x[0], x[1])
    cases.assert_tight(cases.summarize(cases.rel_err(sph, rs)), "sphericalDirection")
    cases.assert_tight(cases.summarize(cases.rel_err(disk, rd)), "concentricDiskSample")
    # the degenerate centre of the square maps to the centre of the disk
    h = dev(np.full(64, 0.5, np.float32))
    assert np.all(host(R.util_directions(gpu, h, h)[1]) == 0)


SKIN_KEYS = ("sheen_wi", "sheen_f", "sheen_pdf", "sheen_fresnel", "spec_wi", "spec_f", "spec_pdf", "spec_fresnel",
             "r", "r_pdf", "profile", "sheenFresnel", "specularFresnel", "sssWeight")


def _skin_check(gpu, oracle, wo, N, T, params, xi, tag):
    ref = oracle.skin(wo, N, T, params, xi, nthreads=4)
    sk = R.SkinShader(gpu, dev(wo), dev(N), dev(T), **{k: dev(v) for k, v in params.items()})
    got = {k: host(v) for k, v in sk.sampleEvalPdf(dev(xi)).items()}
    for k in SKIN_KEYS:
        st = cases.summarize(cases.rel_err(got[k], ref[k]))
        print("skin", tag, k, st)
        cases.assert_tight(st, (tag, k))
    return got, ref


def test_reflect_direction_and_luminance(gpu, oracle, mixed):
    """a3 / a4 through the ABI: arbitrary (not normalised, not front-facing) vectors and colours"""
    c, _ = mixed
    i = (cases.xi(cases.SEED_EDGE, N, 3) * 2 - 1).astype(np.float32)
    col = cases.xi(cases.SEED_PARITY, N, 3)
    r_ref, l_ref = oracle.reflect_luminance(i, c["N"], col)
    r, lum = (host(t) for t in R.util_reflect_luminance(gpu, dev(i), dev(c["N"]), dev(col)))
    assert np.array_equal(r.view(np.uint32), r_ref.view(np.uint32))
    assert np.array_equal(lum.view(np.uint32), l_ref.view(np.uint32))


def test_skin_mixed(gpu, oracle):
    c = cases.skin_mixed(cases.SEED_PARITY, N)
    xi = cases.xi(cases.SEED_PARITY, N, 6)
    got, ref = _skin_check(gpu, oracle, c["wo"], c["N"], c["T"], c["params"], xi, "mixed")
    # layer weights are exact functions of the sampled Fresnel terms (src/rlSkin.cpp:204,228,238)
    p = c["params"]
    sf = np.where(p["sheen_weight"] > 1e-4, got["sheen_fresnel"] * p["sheen_weight"], 0).astype(np.float32)
    assert np.array_equal(sf, got["sheenFresnel"])


@pytest.mark.parametrize("preset", sorted(cases.SKIN_PRESETS))
def test_skin_presets(gpu, oracle, preset):
    n = 1 << 14
    wo, N, T = cases.frame(cases.SEED_PARITY, n)
    xi = cases.xi(cases.SEED_PARITY, n, 6)
    p = cases.SKIN_PRESETS[preset]
    got, ref = _skin_check(gpu, oracle, wo, N, T, p, xi, preset)
    if p["sheen_weight"] <= 1e-4:      # lobe skipped: zero outputs (src/rlSkin.cpp:191)
        assert np.all(got["sheen_wi"] == 0) and np.all(got["sheen_pdf"] == 0) and np.all(got["sheenFresnel"] == 0)
    if p["specular_weight"] <= 1e-4:
        assert np.all(got["spec_f"] == 0) and np.all(got["specularFresnel"] == 0)
        assert np.array_equal(got["sssWeight"], np.full(n, p["sss_weight"], np.float32))


def test_gaussian_profile_alternate(gpu, oracle, mixed):
    """GaussianProfile (src/rlSss.h:63-97), not instantiated by the reference; fast_exp -> exp (unpinned)"""
    c, x = mixed
    d = np.ascontiguousarray(c["dist"][0])
    ref = oracle.gauss(d, x[0])
    got = [host(t) for t in R.GaussianProfile(gpu, N, dev(d)).sample(dev(x[0]))]
    for nm, a, b in zip(("r", "pdf", "profile"), got, ref):
        st = cases.summarize(cases.rel_err(a, b))
        print("gaussian", nm, st)
        cases.assert_tight(st, ("gaussian", nm))
    # uniform distance
    got_u = [host(t) for t in R.GaussianProfile(gpu, N, 1.5).sample(dev(x[0]))]
    ref_u = oracle.gauss(1.5, x[0])
    assert np.array_equal(got_u[0].view(np.uint32), ref_u[0].view(np.uint32))
    # the pdf integrates the radial density: int pdf(r) 2 pi r dr over [0, maxR] = 1 (numerically, dist 1)
    rr = np.linspace(1e-6, 1.0, 200001, dtype=np.float64)
    var = 1.0 / 12.46
    dens = np.exp(-rr * rr / 2 / var) / (2 * np.pi * var) / (1 - np.exp(-0.5 / var)) * 2 * np.pi * rr
    assert abs(np.trapezoid(dens, rr) - 1.0) < 1e-6


def test_nd_profile_across_the_reciprocal_window(gpu, oracle):
    """The one-sample kernels divide by d_i through its reciprocal while d_i and the radius sit inside rlm::div32_y's operand
    window (2^-13 .. 2^14 for d_i, 2^-40 .. 2^40 for r; nd_make / nd_pdf_profile, rls_device.hpp) and the IEEE way otherwise --
    decided per lane.  Scatter distances from 1e-7 to 1e7, mixed within a wavefront and across the three channels, the window's
    corner values included: radius, pdf and profile equal the oracle's bit for bit on either side of every border."""
    n = 1 << 16
    rng = np.random.default_rng(3)
    e = rng.uniform(-7.0, 7.0, (3, n))
    dist = (10.0 ** e).astype(np.float32)
    corners = np.array([2.0 ** -14, 2.0 ** -13, np.nextafter(np.float32(2.0 ** -13), np.float32(0)), 2.0 ** 14,
                        np.nextafter(np.float32(2.0 ** 14), np.float32(1e9)), 1e-4, np.nextafter(np.float32(1e-4), np.float32(0)),
                        0.0, np.inf, np.nan, 1.0], dtype=np.float32)
    k = np.arange(n)
    for c in range(3):                                       # every 7th lane of channel c takes a corner value
        sel = (k % 7) == c
        dist[c, sel] = corners[(k[sel] // 7 + c) % len(corners)]
    albedo = np.full((3, n), 0.5, np.float32)
    rx = oracle.gen_uniform(5, 0, n, oracle.S_XI0)
    p = R.NDProfile(gpu, n, dev(dist), albedo=dev(albedo))
    ref = oracle.Sss(n, dist, albedo, nthreads=4).nd_sample(rx)
    got = [host(t) for t in p.sample(dev(rx))]
    for name, a, b in zip(("r", "pdf", "profile"), got, ref):
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (name, int((~same).sum()))
    # the probe ray on the same profiles (frame from the seeded generator), radii driven to both ends by the random number
    _, N, T = cases.frame(5, n)
    x = np.stack([rx, oracle.gen_uniform(5, 0, n, oracle.S_XI0 + 1)])
    x[0, ::5] = np.float32(0.0)
    x[0, 1::5] = np.nextafter(np.float32(1.0), np.float32(0.0))
    s = R.SssSampler(gpu, dev(N), dev(T), dev(albedo), dev(dist))
    gp = s.getProbeRay(dev(x[0]), dev(x[1]))
    rp = oracle.Sss(n, dist, albedo, N=N, T=T, nthreads=4).probe(x[0], x[1])
    for name in ("r", "maxdist", "pdf", "profile"):
        a, b = host(gp[name]), rp[name]
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (name, int((~same).sum()))


UNIFORM_DISTANCES = [(1.0, 0.6, 0.35), (1.0, 0.5, 0.25), (0.0, 0.0, 0.0), (1e-5, 1e-5, 1e-5), (1.0, 1e-5, 0.5),
                     (2.0 ** -13, 1.0, 3.0), (float(np.nextafter(np.float32(2.0 ** -13), np.float32(0))), 1.0, 3.0),
                     (2.0 ** 14, 1.0, 0.01), (float(np.nextafter(np.float32(2.0 ** 14), np.float32(1e9))), 5.0, 7.0),
                     (1e-4, 2e-4, 1e-3), (1e7, 1e-7, 1.0), (float("inf"), 1.0, 1.0), (float("nan"), 1.0, 1.0), (37.5, 0.02, 911.0)]


@pytest.mark.parametrize("multiplier", [None, 1.7])
def test_uniform_scatter_distance_is_hoisted_and_bit_identical(gpu, oracle, multiplier):
    """With the scatter distance (and its multiplier) one value for the batch the kernels run setDistance once per thread ahead of
    the tile loop and keep NDProfile in scalar registers (sss.hip, uniform_profile) -- the same values as the per-point
    evaluation: radius, pdf, profile, probe ray and MIS pdf equal the oracle's AND the streamed kernel's (the same distances as
    per-point planes) bit for bit, for distances inside, on the borders of and outside the reciprocal window, degenerate ones
    included."""
    n = 1 << 14
    _, N, T = cases.frame(11, n)
    x = np.stack([oracle.gen_uniform(11, 0, n, oracle.S_XI0 + j) for j in range(2)])
    x[0, ::5] = np.float32(0.0)
    x[0, 1::5] = np.nextafter(np.float32(1.0), np.float32(0.0))
    disp = (cases.xi(cases.SEED_EDGE, n, 3) - 0.5).astype(np.float32)
    sN, _, _ = cases.frame(cases.SEED_EDGE, n)
    albedo = (0.8, 0.5, 0.4)

    def same(a, b, what):
        a, b = np.asarray(a), np.asarray(b)
        ok = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert ok.all(), (what, int((~ok).sum()))

    for dist in UNIFORM_DISTANCES:
        planes = np.repeat(np.asarray(dist, np.float32)[:, None], n, axis=1)
        mult_plane = None if multiplier is None else np.full(n, multiplier, np.float32)
        o = oracle.Sss(n, dist, albedo, multiplier=multiplier, N=N, T=T, nthreads=4)
        kw = {} if multiplier is None else dict(multiplier=multiplier)
        kws = {} if multiplier is None else dict(multiplier=dev(mult_plane))
        pu = R.NDProfile(gpu, n, dist, albedo, **kw)
        ps = R.NDProfile(gpu, n, dev(planes), albedo, **kws)
        ref = o.nd_sample(x[0])
        gu, gs = [host(t) for t in pu.sample(dev(x[0]))], [host(t) for t in ps.sample(dev(x[0]))]
        for name, a, b, c in zip(("r", "pdf", "profile"), gu, gs, ref):
            same(a, c, (dist, name, "uniform vs oracle"))
            same(a, b, (dist, name, "uniform vs streamed"))
        r = np.abs(ref[0]) + np.float32(0.01)
        same(host(pu.getPdf(dev(r))), o.nd_pdf(r), (dist, "getPdf"))
        su = R.SssSampler(gpu, dev(N), dev(T), albedo, dist, **kw)
        ss = R.SssSampler(gpu, dev(N), dev(T), albedo, dev(planes), **kws)
        gu, gs, rp = su.getProbeRay(dev(x[0]), dev(x[1])), ss.getProbeRay(dev(x[0]), dev(x[1])), o.probe(x[0], x[1])
        for name in ("r", "origin", "dir", "maxdist", "pdf", "profile"):
            same(host(gu[name]), rp[name], (dist, name, "probe, uniform vs oracle"))
            same(host(gu[name]), host(gs[name]), (dist, name, "probe, uniform vs streamed"))
        same(host(su.misPdf(dev(disp), dev(sN))), o.mis_pdf(disp, sN, False), (dist, "mis pdf"))


SKIN_UNIFORM_CASES = {
    "all_layers": dict(cases.SKIN_DEFAULTS, sheen_weight=0.3, sss_scatter_dist=(1.0, 0.6, 0.35), sss_color=(1.0, 0.84, 0.5)),
    "no_sheen": dict(cases.SKIN_DEFAULTS, sss_scatter_dist=(0.4, 0.2, 0.1), sss_dist_multiplier=2.5),
    "no_specular": dict(cases.SKIN_DEFAULTS, sheen_weight=0.9, specular_weight=0.0, sheen_roughness=0.05, sheen_ior=2.4),
    "no_sss": dict(cases.SKIN_DEFAULTS, sheen_weight=1.0, specular_weight=1.0, sss_weight=0.0),
    "weights_on_the_threshold": dict(cases.SKIN_DEFAULTS, sheen_weight=1e-4, specular_weight=float(np.nextafter(np.float32(1e-4), np.float32(1)))),
    "tiny_ior_and_roughness": dict(cases.SKIN_DEFAULTS, sheen_weight=0.5, sheen_ior=0.0, sheen_roughness=0.0, specular_ior=1e-5,
                                   specular_roughness=1.0),
    "distance_outside_the_window": dict(cases.SKIN_DEFAULTS, sheen_weight=0.5, sss_scatter_dist=(1e-5, 3e4, 1.0)),
    "degenerate_distance": dict(cases.SKIN_DEFAULTS, sheen_weight=0.5, sss_scatter_dist=(0.0, 0.0, 0.0)),
}


@pytest.mark.parametrize("maps", [False, True])
@pytest.mark.parametrize("case", sorted(SKIN_UNIFORM_CASES))
def test_skin_uniform_parameters_are_hoisted_and_bit_identical(gpu, oracle, case, maps):
    """Every parameter one value for the batch: skin_kernel<UNIFORM_ALL> evaluates the parameter-only arithmetic once per thread
    (the lobes' roughness / ior terms, NDProfile::setDistance).  Its 24 outputs equal the oracle's and those of the streamed
    kernel (the same values as per-point planes) bit for bit, with layers switched off, weights on the 1e-4 threshold,
    degenerate ior / roughness and scatter distances outside the reciprocal window."""
    n = 1 << 14
    wo, N, T = cases.frame(cases.SEED_PARITY, n)
    xi = cases.xi(cases.SEED_PARITY, n, 6)
    p = dict(SKIN_UNIFORM_CASES[case])
    if maps:      # the colours and layer weights textured (per-point planes, weights straddling the 1e-4 gates), the rest plain:
        # the MIXED kernel (per-parameter tests in the loop)
        m = cases.skin_mixed(cases.SEED_EDGE, n)["params"]
        for k in ("sss_color", "specular_color", "sheen_color", "sss_weight", "specular_weight", "sheen_weight"):
            p[k] = m[k].copy()
        for k in ("sss_weight", "specular_weight", "sheen_weight"):
            p[k][::11] = np.float32(1e-4)
            p[k][5::13] = np.float32(0.0)
    ref = oracle.skin(wo, N, T, p, xi, nthreads=4)
    planes = {k: (v if isinstance(v, np.ndarray) else
                  np.repeat(np.asarray(v, np.float32)[:, None], n, axis=1) if np.ndim(v) else np.full(n, v, np.float32))
              for k, v in p.items()}
    gu = {k: host(v) for k, v in R.SkinShader(gpu, dev(wo), dev(N), dev(T), **{k: dev(v) for k, v in p.items()})
          .sampleEvalPdf(dev(xi)).items()}
    gs = {k: host(v) for k, v in R.SkinShader(gpu, dev(wo), dev(N), dev(T), **{k: dev(v) for k, v in planes.items()})
          .sampleEvalPdf(dev(xi)).items()}
    for k in SKIN_KEYS:
        for other, what in ((ref[k], "oracle"), (gs[k], "streamed kernel")):
            a, b = gu[k], np.asarray(other)
            ok = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
            assert ok.all(), (case, k, what, int((~ok).sum()))
