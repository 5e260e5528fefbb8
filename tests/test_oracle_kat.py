"""CPU: the oracle against the only reference-derived numbers that exist (SURVEY.md 8(c) probe KATs,
tests/golden/survey_kat.json) and against closed-form identities of the published formulas."""
import ctypes as C
import json
import math
from pathlib import Path

import numpy as np

import cases
import oracle_lib as O

KAT = json.loads((Path(__file__).parent / "golden" / "survey_kat.json").read_text())


def _one(v):
    return np.asarray(v, np.float32).reshape(3, 1)


def test_ggx_survey_kat():
    k = KAT["ggx"]
    g = O.Ggx(_one(k["wo"]), _one(k["N"]), _one(k["T"]), KsColor=k["KsColor"], ior=k["ior"],
              roughness=float(np.sqrt(np.float32(k["roughness_squared"]))), anisotropic=k["anisotropic"])
    wi, f, pdf, F = g.sample_eval_pdf(np.float32([k["xi"][0]]), np.float32([k["xi"][1]]))
    # printed with 9 significant digits in the survey: agree to float32 round-off of the print
    assert np.allclose(wi[:, 0], k["L"], rtol=0, atol=2e-9 + 6e-8 * 0), wi[:, 0]
    assert abs(f[0, 0] - k["f"]) <= 1e-9 * 2 and f[0, 0] == f[1, 0] == f[2, 0]
    assert abs(pdf[0] - k["pdf"]) <= 3e-8
    assert 0 < F[0] < 1


def test_nd_survey_kat():
    k = KAT["nd"]
    n = 1
    s = O.Sss(n, k["dist"], k["albedo"])
    lib = O.lib()

    class ND(C.Structure):
        _fields_ = [("d", C.c_float * 3), ("c1", C.c_float * 3), ("c2", C.c_float * 3), ("maxR", C.c_float)]
    p = ND()
    lib.orc_nd_set_distance.argtypes = [C.POINTER(ND), O.V3, O.RGB]
    lib.orc_nd_set_distance(p, O.V3(*k["dist"]), O.RGB(*k["albedo"]))
    assert p.maxR == k["maxR"]
    for x, r in k["radius"]:
        got = s.nd_sample(np.float32([x]))[0][0]
        assert abs(got - r) <= 2e-6 * r, (x, got, r)      # double- vs float-overload build: ~1e-7 relative
    assert abs(s.nd_pdf(np.float32([0.5]))[0] - k["pdf_at_0.5"]) <= 2e-7
    assert abs(s.nd_profile(np.float32([0.5]))[0, 0] - k["profile_r_at_0.5"]) <= 2e-7


def test_ggx_isotropic_closed_forms():
    """Walter et al. EGSR'07: D (eq. 33), G1 (eq. 34) and the dielectric Fresnel (eq. 22) at known points"""
    n = 4096
    wo, N, T = cases.frame(5, n)
    rough = np.float32(0.6)
    g = O.Ggx(wo, N, T, ior=1.5, roughness=rough)
    # mirror direction of the normal itself: h = N... use wi = reflect(wo, N) so that h == N
    d = (wo * N).sum(axis=0)
    wi = (2 * d * N - wo).astype(np.float32)
    pdf = g.pdf(wi).astype(np.float64)
    a = float(rough) ** 2
    D = 1.0 / (math.pi * a * a)                      # D(h = n)
    cos = d.astype(np.float64)
    G1 = 2.0 / (1.0 + np.sqrt(1.0 + a * a * (1.0 / cos ** 2 - 1.0)))
    want = np.maximum(D * G1 / cos * 0.25, 1e-4)
    assert np.allclose(pdf, want, rtol=2e-5)
    f = g.eval(wi).astype(np.float64)[0]
    c = cos
    gg = np.sqrt(1.5 ** 2 - 1 + c * c)
    F = 0.5 * ((gg - c) / (gg + c)) ** 2 * (1 + ((c * (gg + c) - 1) / (c * (gg - c) + 1)) ** 2)
    want_f = F * G1 * G1 * D * 0.25 / (c * c) * c
    assert np.allclose(f, want_f, rtol=5e-5)


def test_sample_pdf_consistency_monte_carlo():
    """integral of f/pdf over samples = directional albedo <= 1 for white Ks, and the VNDF weight
    f/pdf = F * G1(L) is bounded by 1 per sample (Heitz & d'Eon EGSR'14)"""
    n = 1 << 14
    wo, N, T = cases.frame(11, n)
    x = cases.xi(11, n, 2)
    g = O.Ggx(wo, N, T, ior=1.5, roughness=0.5)
    wi, f, pdf, F = g.sample_eval_pdf(x[0], x[1])
    w = f[0] / pdf
    ok = pdf > 1.0001e-4                                   # away from the pdf floor
    assert np.all(w[ok] <= 1.0 + 1e-4) and np.all(w[ok] >= 0)
    assert np.all((F >= 0) & (F <= 1))


def test_disney_lobes_basic():
    n = 1 << 12
    c = cases.disney_mixed(3, n)
    sc = {k: c[k] for k in O.DISNEY_SCALARS}
    d = O.Disney(c["wo"], c["N"], c["T"], base_color=c["base_color"], **sc)
    x = cases.xi(3, n, 2)
    wi, f, pdf = d.sample_eval_pdf(O.RAY_DIFFUSE, x[0], x[1])
    cosl = (wi * c["N"]).sum(axis=0)
    assert np.all(cosl >= -1e-6)                                 # cosine hemisphere
    assert np.allclose(pdf, np.maximum(1e-4, cosl * np.float32(0.31830988)), rtol=1e-6, atol=1e-9)
    assert np.all(f >= 0) and np.all(np.isfinite(f))
    wi, f, pdf = d.sample_eval_pdf(O.RAY_GLOSSY, x[0], x[1])
    zero = (wi == 0).all(axis=0)
    assert np.all(f[:, zero] == 0) and np.all(pdf[zero] == 0)
    assert np.all(np.isfinite(f)) and np.all(pdf >= 0)
    # default preset (all scalars 0): specular lobe is black except sheen/clearcoat, both 0
    wo, N, T = cases.frame(4, n)
    d0 = O.Disney(wo, N, T, base_color=(0.85, 0.7047, 0.2057))
    wi0, f0, _ = d0.sample_eval_pdf(O.RAY_GLOSSY, x[0], x[1])
    assert np.all(f0 >= 0)


def test_nd_profile_properties():
    """the sampled radius stays inside maxR = 3 max(d); pdf integrates the radial density"""
    n = 1 << 14
    c = cases.sss_mixed(9, n)
    s = O.Sss(n, c["dist"], c["albedo"])
    rx = cases.xi(9, n, 1)[0]
    r, pdf, prof = s.nd_sample(rx)
    maxR = 3 * c["dist"].max(axis=0)
    assert np.all(r >= 0) and np.all(r <= maxR * (1 + 1e-5))
    assert np.all(pdf[r > 1e-4] > 0) and np.all(prof >= 0)
    # Christensen-Burley: R(r) * 2 pi r integrates to 1 per channel over [0, inf) -> check numerically for d = 1
    rr = np.linspace(1e-4, 60, 600001, dtype=np.float64).astype(np.float32)
    s1 = O.Sss(rr.size, (1.0, 1.0, 1.0))
    R = s1.nd_profile(rr)[0].astype(np.float64)
    integral = np.trapezoid(R * 2 * np.pi * rr.astype(np.float64), rr.astype(np.float64))
    assert abs(integral - 1.0) < 2e-3


def test_probe_ray_geometry():
    n = 1 << 12
    c = cases.sss_mixed(21, n)
    x = cases.xi(21, n, 2)
    s = O.Sss(n, c["dist"], c["albedo"], N=c["N"], T=c["T"], has_dPdu=True)
    o = s.probe(x[0], x[1])
    maxR = (3 * c["dist"].max(axis=0)).astype(np.float64)
    # the probe origin lies on the sphere of radius maxR about P, the ray points back through the disk
    assert np.allclose(np.linalg.norm(o["origin"].astype(np.float64), axis=0), maxR, rtol=2e-5)
    assert np.allclose(np.linalg.norm(o["dir"].astype(np.float64), axis=0), 1.0, atol=1e-5)
    along = -(o["origin"] * o["dir"]).sum(axis=0)
    assert np.allclose(along * 2, o["maxdist"], rtol=1e-4, atol=1e-5)
    # axis choice by rx: < .5 -> -N, < .75 -> +U, else +V (src/rlSss.h:491-500,519-529)
    dn = (o["dir"] * c["N"]).sum(axis=0)
    assert np.all(dn[x[0] < 0.5] < -0.999)
    assert np.all(np.abs(dn[x[0] >= 0.5]) < 1e-5)


def test_skin_layer_arithmetic():
    n = 1 << 12
    c = cases.skin_mixed(17, n)
    xi = cases.xi(17, n, 6)
    out = O.skin(c["wo"], c["N"], c["T"], c["params"], xi)
    p = c["params"]
    on_sheen = p["sheen_weight"] > 1e-4
    on_spec = p["specular_weight"] > 1e-4
    sf = np.where(on_sheen, out["sheen_fresnel"] * p["sheen_weight"], 0).astype(np.float32)
    pf = np.where(on_spec, out["spec_fresnel"] * p["specular_weight"], 0).astype(np.float32)
    assert np.array_equal(sf, out["sheenFresnel"]) and np.array_equal(pf, out["specularFresnel"])
    w = (p["sss_weight"] * (np.float32(1) - pf * (np.float32(1) - sf))).astype(np.float32)
    assert np.array_equal(w, out["sssWeight"])
    off = out["sssWeight"] < 1e-4
    assert np.all(out["r"][off] == 0) and np.all(out["profile"][:, off] == 0)


def test_generator_statistics():
    n = 1 << 16
    wo, N, T = O.gen_frame(1234, 0, n)
    for v in (wo, N, T):
        assert np.abs(np.linalg.norm(v.astype(np.float64), axis=0) - 1).max() < 1e-6
    assert np.abs((N * T).sum(axis=0)).max() < 1e-6
    cosv = (wo * N).sum(axis=0)
    assert cosv.min() > 0.0199 and abs(cosv.mean() - 0.51) < 0.01
    assert np.abs(N.mean(axis=1)).max() < 0.02
    u = O.gen_uniform(1234, 0, n, 11)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    a = O.gen_aniso(1234, 0, n)
    assert np.all(a[0::2] == 0) and a[1::2].mean() > 0.4
    # counter-based: a shard generated with an index offset equals the slice of the whole
    w2, _, _ = O.gen_frame(1234, 1000, 500)
    assert np.array_equal(w2, wo[:, 1000:1500])


def test_reflect_direction_and_luminance_closed_forms():
    """a3 reflectDirection = 2 |i.n| n - i (src/rlUtil.h:31-34: with the ABS a back-facing normal does NOT give the
    mirror direction), a4 colorToLuminance (src/rlUtil.h:36-39: Rec.709 weights as written)"""
    import numpy as np
    import oracle_lib as O
    i = np.array([[0.6, 0.0, 0.3], [0.0, 0.6, -0.4], [0.8, 0.8, 0.5]], np.float32)
    n = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, -1.0, 1.0]], np.float32)
    c = np.array([[1.0, 0.0, 0.25], [0.0, 1.0, 0.5], [0.0, 0.0, 0.75]], np.float32)
    r, lum = O.reflect_luminance(i, n, c)
    # column 0: mirror about +z; column 1: normal flipped -> 2 * 0.8 * (0,0,-1) - i, not the mirror direction
    assert np.allclose(r[:, 0], [-0.6, 0.0, 0.8]) and np.allclose(r[:, 1], [0.0, -0.6, -2.4])
    assert np.allclose(r[:, 2], [-0.3, 0.4, 0.5])
    w = np.array([0.212671, 0.715160, 0.072169])
    assert abs(lum[0] - w[0]) < 1e-7 and abs(lum[1] - w[1]) < 1e-7
    assert abs(lum[2] - float(w @ [0.25, 0.5, 0.75])) < 1e-7
