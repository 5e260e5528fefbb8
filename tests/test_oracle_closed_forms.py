"""CPU: known answers the PUBLISHED formulas fix, per function -- so that a typo in the oracle cannot hide behind
"GPU == oracle".  Sources: Walter et al. EGSR'07 (eqs. 22, 33, 34, 40), Burley, "Physically Based Shading at Disney"
(SIGGRAPH 2012 course notes, GTR1 / GTR2 / diffuse), Christensen & Burley 2015 (normalized diffusion), each as cited
by the reference's own source comments (src/rlDisney.cpp:1-10, src/rlSss.cpp)."""
import ctypes as C
import math

import numpy as np

import cases
import oracle_lib as O


def _hemisphere(nt=1500, nph=720):
    """midpoint quadrature over the upper hemisphere: directions [3, nt*nph], weights sin(theta) dtheta dphi"""
    th = (np.arange(nt) + 0.5) * (0.5 * math.pi / nt)
    ph = (np.arange(nph) + 0.5) * (2 * math.pi / nph)
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)]).reshape(3, -1)
    w = (np.sin(T) * (0.5 * math.pi / nt) * (2 * math.pi / nph)).reshape(-1)
    return d, w


def _z_frame(n):
    N = np.tile(np.array([[0.0], [0.0], [1.0]], np.float32), (1, n))
    T = np.tile(np.array([[1.0], [0.0], [0.0]], np.float32), (1, n))
    return N, T


def test_ggx_at_normal_incidence_walter_eq_22_33_34():
    """view = light = normal: h = n, tan(theta) = 0 -> D = 1 / (pi ax ay) (eq. 33), G1 = 1 (eq. 34),
    F = ((eta - 1) / (eta + 1))^2 (eq. 22 at c = 1)"""
    N, T = _z_frame(4)
    rough = np.float32([0.3, 0.6, 0.9, 0.5])
    aniso = np.float32([0.0, 0.0, 0.5, 1.0])
    ior = np.float32([1.5, 1.33, 2.4, 1.5])
    g = O.Ggx(N.copy(), N, T, KsColor=(1.0, 1.0, 1.0), ior=ior, roughness=rough, anisotropic=aniso)
    f, pdf = g.eval(N.copy())[0].astype(np.float64), g.pdf(N.copy()).astype(np.float64)
    aspect = np.sqrt(1 - 0.9 * aniso.astype(np.float64))
    ax, ay = rough.astype(np.float64) ** 2 / aspect, rough.astype(np.float64) ** 2 * aspect
    D = 1 / (math.pi * ax * ay)
    F0 = ((ior.astype(np.float64) - 1) / (ior.astype(np.float64) + 1)) ** 2
    assert np.allclose(pdf, D * 0.25, rtol=2e-6)
    assert np.allclose(f, F0 * D * 0.25, rtol=5e-6)


def test_ggx_distribution_is_normalised():
    """eq. 33's defining property: integral of D(m) (m.n) over the hemisphere = 1, isotropic and anisotropic.  At
    normal incidence evalPdf(wi) = D(h) / 4 with h = normalize(wi + n) (G1 = 1), so D comes out of the pdf entry point."""
    h, w = _hemisphere()
    n = h.shape[1]
    N, T = _z_frame(n)
    wi = (2 * h[2] * h - N).astype(np.float32)                   # mirror of the normal about h
    for rough, aniso in ((0.5, 0.0), (0.7, 0.0), (0.6, 0.8)):
        g = O.Ggx(N.copy(), N, T, ior=1.5, roughness=rough, anisotropic=aniso, nthreads=8)
        D = 4.0 * g.pdf(wi).astype(np.float64)
        total = float((D * h[2] * w).sum())
        assert abs(total - 1.0) < 2e-3, (rough, aniso, total)


def test_burley_gtr1_gtr2_normalisation_and_smith_g():
    lib = O.lib()
    lib.orc_kat_D_GTR1.restype = C.c_float
    lib.orc_kat_D_GTR1.argtypes = [C.c_float, C.c_float]
    lib.orc_kat_D_GTR2Aniso.restype = C.c_float
    lib.orc_kat_D_GTR2Aniso.argtypes = [C.c_float, C.c_float, O.V3]
    lib.orc_kat_smithG_GGX.restype = C.c_float
    lib.orc_kat_smithG_GGX.argtypes = [C.c_float, C.c_float]
    # GTR1 (Burley eq. 4 with gamma = 1): D = (a^2 - 1) / (pi ln a^2 (1 + (a^2 - 1) cos^2)); integral D cos dw = 1
    # analytically: 2 pi * integral_0^1 D(c) c dc = (a^2-1)/(ln a^2) * ln(1 + (a^2-1) c^2)/(a^2-1) |_0^1 = 1
    # pointwise against the formula, then the normalisation numerically from the oracle's own values: in u = cos^2 the
    # integral is pi * int_0^1 D(u) du; the peak at u = 1 is a^2 wide (1e-6 for gloss 1), so the grid is logarithmic in 1 - u
    t = np.concatenate([[0.0], np.exp(np.linspace(math.log(1e-10), 0.0, 6000))])        # t = 1 - u
    for gloss in (0.0, 0.3, 1.0):
        alpha = (1 - gloss) * 0.1 + gloss * 0.001                  # LERP(gloss, .1, .001), src/rlDisney.cpp:547
        a2 = alpha * alpha
        D = np.array([lib.orc_kat_D_GTR1(gloss, float(1.0 - x)) for x in t], np.float64)
        want = (a2 - 1) / (math.pi * math.log(a2) * (1 + (a2 - 1) * (1.0 - t)))
        # fp32 evaluation of 1 + (a^2 - 1) u cancels near the peak (relative error ~ 6e-8 / (t + a^2)): compare pointwise
        # where that is below 1e-5, and the integral of the oracle's values with the formula's over the same range
        far = t > 1e-2
        assert np.allclose(D[far], want[far], rtol=1e-4), gloss
        assert abs(math.pi * float(np.trapezoid(want, t)) - 1.0) < 2e-3, gloss          # the formula is normalised
        mid = t > 1e-4
        io, iw = math.pi * float(np.trapezoid(D[mid], t[mid])), math.pi * float(np.trapezoid(want[mid], t[mid]))
        assert abs(io - iw) < 2e-3 * iw, (gloss, io, iw)
    # GTR2 anisotropic (Burley eq. 13 / Walter eq. 33): at m = n it is 1 / (pi ax ay); normalised over the hemisphere
    assert abs(lib.orc_kat_D_GTR2Aniso(0.2, 0.4, O.V3(0.0, 0.0, 1.0)) - 1 / (math.pi * 0.2 * 0.4)) < 1e-5
    h, w = _hemisphere(600, 360)
    D = np.array([lib.orc_kat_D_GTR2Aniso(0.3, 0.5, O.V3(float(x), float(y), float(z))) for x, y, z in h.T[::7]], np.float64)
    assert abs(float((D * h[2, ::7] * w[::7]).sum()) * 7 - 1.0) < 5e-3
    # smithG_GGX (src/rlDisney.cpp:570-577) is G1 / (2 N.V) of Walter eq. 34: 1 / (c + sqrt(a^2 + c^2 - a^2 c^2))
    for cth, a in ((1.0, 0.5), (0.5, 0.25), (0.1, 0.8)):
        g1 = 2 / (1 + math.sqrt(1 + a * a * (1 / cth ** 2 - 1)))
        assert abs(lib.orc_kat_smithG_GGX(cth, a) - g1 / (2 * cth)) < 2e-6 * g1 / (2 * cth) + 1e-7


def test_disney_diffuse_at_normal_incidence():
    """view = light = normal: Schlick weights FL = FV = 0 -> Burley's Fd = 1, f = base / pi * (1 - metallic) * cos;
    the Hanrahan-Krueger branch gives 1.25 * (Fss * (1 / (NL + NV) - .5) + .5) = 0.625 with Fss = 1
    (src/rlDisney.cpp:214-233)"""
    N, T = _z_frame(3)
    base = (0.8, 0.5, 0.2)
    d = O.Disney(N.copy(), N, T, base_color=base, subsurface=np.float32([0, 1, 0.5]), metallic=np.float32([0, 0, 0.25]),
                 roughness=np.float32([0.7, 0.7, 0.7]))
    f = d.eval(O.RAY_DIFFUSE, N.copy()).astype(np.float64)
    mix = np.array([1.0, 0.625, 0.5 * 1.0 + 0.5 * 0.625])
    om = np.array([1.0, 1.0, 0.75])
    for k in range(3):
        assert np.allclose(f[k], base[k] / math.pi * mix * om, rtol=3e-6)
    assert np.allclose(d.pdf(O.RAY_DIFFUSE, N.copy()), 1 / math.pi, rtol=1e-6)          # cosine lobe at the pole


def test_normalized_diffusion_mass_and_sampling():
    """Christensen-Burley profile R(r) = (e^(-r/d) + e^(-r/3d)) / (8 pi d r): the mass of channel i inside
    maxR = 3 max(d) is (C1_i + 3 C2_i) / 4 with C1 = 1 - e^(-maxR/d), C2 = 1 - e^(-maxR/3d) (src/rlSss.cpp:31-32);
    getPdf integrates to 1 over the disc of radius maxR; getRadius inverts each lobe's CDF"""
    dist = (1.0, 0.5, 0.25)
    maxR = 3.0 * max(dist)
    m = 1200001
    r = (np.arange(m) + 0.5) * (maxR / m)
    s = O.Sss(m, dist, nthreads=8)
    R = s.nd_profile(r.astype(np.float32)).astype(np.float64)
    dr = maxR / m
    for k, d in enumerate(dist):
        c1, c2 = 1 - math.exp(-maxR / d), 1 - math.exp(-maxR / (3 * d))
        mass = float((R[k] * 2 * math.pi * r).sum() * dr)
        assert abs(mass - (c1 + 3 * c2) / 4) < 2e-4, (k, mass, (c1 + 3 * c2) / 4)
    pdf = s.nd_pdf(r.astype(np.float32)).astype(np.float64)
    assert abs(float((pdf * 2 * math.pi * r).sum() * dr) - 1.0) < 2e-4
    # sampling: the lobe is picked by thirds of xi (thresholds .3333 / .6666, src/rlSss.h:30-42), inside it the
    # two-exponential CDF  F(r) = (1 - e^(-r/d) + 3 (1 - e^(-r/3d))) / (C1 + 3 C2)  is inverted piecewise
    n = 30000
    xi = ((np.arange(n) + 0.5) / n).astype(np.float32)
    s2 = O.Sss(n, dist)
    rr = s2.nd_sample(xi)[0].astype(np.float64)
    for k, (lo, hi, d) in enumerate(((0.0, 0.3333, dist[0]), (0.3333, 0.6666, dist[1]), (0.6666, 1.0, dist[2]))):
        sel = (xi > lo + 1e-3) & (xi < hi - 1e-3)
        u = (xi[sel].astype(np.float64) - lo) / (hi - lo)
        c1, c2 = 1 - math.exp(-maxR / d), 1 - math.exp(-maxR / (3 * d))
        w = c1 / (c1 + 3 * c2)
        x = rr[sel]
        # the reference samples e^(-r/d) with probability w and e^(-r/3d) otherwise: the radii of the first branch follow
        # the truncated exponential CDF (1 - e^(-r/d)) / C1 in u / w, the others (1 - e^(-r/3d)) / C2 in (u - w) / (1 - w)
        first = u <= w
        assert np.allclose((1 - np.exp(-x[first] / d)) / c1, u[first] / w, atol=2e-5)
        assert np.allclose((1 - np.exp(-x[~first] / (3 * d))) / c2, (u[~first] - w) / (1 - w), atol=2e-5)
        assert (x <= maxR * (1 + 1e-6)).all()


def test_refraction_obeys_snell_walter_eq_40():
    """the stand-in for AiRefractRay, about the sampled microfacet normal m: sin(theta_t) = (eta_i / eta_t) sin(theta_i),
    the refracted direction lies in the plane of (view, m), on the other side of m, with unit length"""
    n = 4096
    wo, N, T = cases.frame(21, n)
    g = O.Ggx(wo, N, T, ior=1.5, roughness=0.3)
    x = cases.xi(21, n, 2)
    m = g.microfacet(x[0], x[1]).astype(np.float64)
    wt, w, ok = g.refract(x[0], x[1])
    wt, v = wt.astype(np.float64), wo.astype(np.float64)
    assert ok.all()                                                  # entering the denser medium: never total reflection
    ci = (v * m).sum(axis=0)
    ct = -(wt * m).sum(axis=0)
    use = ci > 0.05
    si, st = np.sqrt(1 - ci ** 2), np.sqrt(np.maximum(0, 1 - ct ** 2))
    assert (ct[use] > 0).all()
    assert np.abs(st[use] - si[use] / 1.5).max() < 3e-5            # fp32 cosines through sqrt(1 - c^2)
    assert np.abs(np.linalg.norm(wt, axis=0) - 1)[use].max() < 2e-6
    plane = np.cross(v.T, m.T)
    assert np.abs((plane * wt.T).sum(axis=1))[use].max() < 2e-6
