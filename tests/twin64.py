"""An independent float64 restatement (numpy) of the rlGgx closure and NDProfile, written from the reference's lines -- NOT
from oracle/rls_oracle.c -- as a second anchor for the oracle: a transcription error would have to be made twice, in two
languages, to go unnoticed.  Each function cites the lines it follows.  The twin evaluates every formula in float64 from the
float32 inputs, with the reference's branch structure; two quantities are formed in float32 as the reference forms them,
because their fp32 value IS the value the reference's branches and results depend on: the rational fit of the second slope
(its denominator cancels to 4.9e-4 at u = 1) and the comparisons against AI_EPSILON thresholds.
Test infrastructure only (tests/test_oracle_float64_twin.py, tools/studies/)."""
import numpy as np

F = np.float32
EPS = float(F(1e-4))


def _norm(v):
    n = np.linalg.norm(v, axis=0)
    return v / np.where(n > 0, n, 1.0)


class Ggx64:
    """GgxSamplerT's constructor, src/rlGgx.h:130-156: iors, frame (U = tangent, V = N x U), alphas, mRoughness"""

    def __init__(self, c):
        self.wo, self.N, self.T = (np.asarray(c[k], np.float64) for k in ("wo", "N", "T"))
        n = self.wo.shape[1]
        full = lambda v: np.broadcast_to(np.asarray(v, np.float64), (n,))
        rough, aniso, ior = full(c["roughness"]), full(c["anisotropic"]), full(c["ior"])
        ks = np.asarray(c["KsColor"], np.float64)
        self.ks = ks if ks.ndim == 2 else np.repeat(ks[:, None], n, axis=1)
        self.iorIn, self.iorOut = np.ones(n), np.maximum(ior, 1e-4)
        self.U, self.V = self.T, np.cross(self.N.T, self.T.T).T
        aspect = np.sqrt(1.0 - aniso * 0.9)
        self.ax = np.maximum(1e-4, rough * rough / aspect)
        self.ay = np.maximum(1e-4, rough * rough * aspect)
        self.rough = np.maximum(1e-5, rough * rough)

    # src/rlGgx.cpp:14-61
    def _slope(self, theta, rx, ry, ry32):
        with np.errstate(all="ignore"):
            r = np.sqrt(rx / (1.0 - rx))
            ux, uy = r * np.cos(2 * np.pi * ry), r * np.sin(2 * np.pi * ry)
            B = np.tan(theta)
            B2 = B * B
            G1 = 2.0 / (1.0 + np.sqrt(1.0 + B2))
            A = 2.0 * rx / G1 - 1.0
            A2 = A * A
            tmp = 1.0 / (A2 - 1.0)
            D = np.sqrt(np.maximum(0.0, B2 * tmp * tmp - (A2 - B2) * tmp))
            x1, x2 = B * tmp - D, B * tmp + D
            sx = np.where((A < 0) | (x2 > 1.0 / B), x1, x2)
            up = ry32 > F(0.5)
            u = np.where(up, (F(2) * (ry32 - F(0.5)).astype(F)).astype(F), (F(2) * (F(0.5) - ry32).astype(F)).astype(F)).astype(F)
            m = lambda a, b: (a * b).astype(F)
            num = m(u, (m(u, (m(u, F(0.27385)) - F(0.73369)).astype(F)) + F(0.46341)).astype(F))
            den = (m(u, (m(u, (m(u, F(0.093073)) + F(0.309420)).astype(F)) - F(1.0)).astype(F)) + F(0.597999)).astype(F)
            z = num.astype(np.float64) / den.astype(np.float64)
            sy = np.where(up, 1.0, -1.0) * z * np.sqrt(1.0 + sx * sx)
            uniform = (theta < EPS) | (np.abs(A2 - 1.0) < EPS)
        return np.where(uniform, ux, sx), np.where(uniform, uy, sy)

    # VNDFKernel::evalSample, src/rlGgx.cpp:63-99
    def microfacet(self, rx32, ry32):
        rx, ry = rx32.astype(np.float64), ry32.astype(np.float64)
        V = self.wo
        cosv = np.clip((self.N * V).sum(0), -1.0, 1.0)
        phiv = np.arctan2((self.V * V).sum(0), (self.U * V).sum(0))
        sinv = np.sqrt(np.maximum(0.0, 1.0 - cosv * cosv))                  # sphericalDirection, src/rlUtil.h:21-29
        l = _norm(np.stack([sinv * np.cos(phiv) * self.ax, sinv * np.sin(phiv) * self.ay, cosv]))
        flat = ~(l[2] < 1.0 - EPS)
        theta = np.where(flat, 0.0, np.arccos(np.clip(l[2], -1, 1)))
        phi = np.where(flat, 0.0, np.arctan2(l[1], l[0]))
        sx, sy = self._slope(theta, rx, ry, ry32)
        c, s = np.cos(phi), np.sin(phi)
        ox, oy = -(c * sx - s * sy) * self.ax, -(s * sx + c * sy) * self.ay
        return _norm(ox * self.U + oy * self.V + self.N)

    # rlUtil reflectDirection, src/rlUtil.h:31-34
    def reflect(self, m):
        return m * (2.0 * np.abs((self.wo * m).sum(0))) - self.wo

    # src/rlGgx.h:249-270
    def fresnel(self, i, m):
        c = np.abs((i * m).sum(0))
        g2 = (self.iorOut / self.iorIn) ** 2 - 1.0 + c * c
        with np.errstate(all="ignore"):
            g = np.sqrt(np.maximum(g2, 0.0))
            gmc, gpc = g - c, g + c
            f = 0.5 * (gmc / gpc) ** 2 * (1.0 + ((c * gpc - 1.0) / (c * gmc + 1.0)) ** 2)
        return np.where(g2 < 0, 1.0, f)

    # src/rlGgx.h:332-340
    def D(self, m):
        mu, mv, mn = (m * self.U).sum(0), (m * self.V).sum(0), (m * self.N).sum(0)
        return (1.0 / np.pi) / (self.ax * self.ay * ((mu / self.ax) ** 2 + (mv / self.ay) ** 2 + mn * mn) ** 2)

    # src/rlGgx.h:343-357
    def G1(self, v, m):
        vm, vn = (v * m).sum(0), (v * self.N).sum(0)
        with np.errstate(all="ignore"):
            g = 2.0 / (1.0 + np.sqrt(1.0 + self.rough ** 2 * (1.0 / (vn * vn) - 1.0)))
        return np.where(vm * vn < 0, 0.0, g)

    # evalPdf, src/rlGgx.h:121-127 with VNDFKernel::evalPdf, 72-80
    def pdf(self, wi):
        H = _norm(self.wo + wi)
        with np.errstate(all="ignore"):
            p = self.D(H) * self.G1(self.wo, H) / np.abs((self.wo * self.N).sum(0)) * 0.25
        return np.maximum(p, EPS)

    # evalBrdf, src/rlGgx.h:110-119, evalReflectance 158-165, reflection 304-313
    def eval(self, wi):
        vn = (self.wo * self.N).sum(0)
        hr = np.where(vn < 0, -1.0, 1.0) * _norm(wi + self.wo)
        ln = (wi * self.N).sum(0)
        with np.errstate(all="ignore"):
            refl = self.fresnel(self.wo, hr) * self.G1(self.wo, hr) * self.G1(wi, hr) * self.D(hr) * 0.25 / (np.abs(ln) * np.abs(vn))
        f = self.ks * (refl * ln)
        small = (np.abs(self.ks) < EPS).all(axis=0) | (wi == 0).all(axis=0)
        return np.where(small, 0.0, f)

    # getSampleWeight, src/rlGgx.h:294-301
    def sample_weight(self, o, m):
        ih = (self.wo * m).sum(0)
        mn, iN = np.abs((m * self.N).sum(0)), np.abs((self.wo * self.N).sum(0))
        with np.errstate(all="ignore"):
            return self.G1(self.wo, m) * self.G1(o, m) * np.abs(ih / (iN * mn))


class NdProfile64:
    """NDProfile, src/rlSss.h:27-61, src/rlSss.cpp:20-106 (scatter distances already multiplied)"""

    def __init__(self, dist):
        self.d = np.asarray(dist, np.float64)
        self.maxR = self.d.max(axis=0) * 3.0
        with np.errstate(all="ignore"):
            self.c1 = 1.0 - np.exp(-self.maxR / self.d)
            self.c2 = 1.0 - np.exp(-self.maxR / self.d / 3.0)

    def radius(self, rx32):
        rx = rx32.astype(np.float64)
        lo = np.where(rx32 < F(0.3333), 0.0, np.where(rx32 > F(0.6666), float(F(0.6666)), float(F(0.3333))))
        hi = np.where(rx32 < F(0.3333), float(F(0.3333)), np.where(rx32 > F(0.6666), 1.0, float(F(0.6666))))
        k = np.where(rx32 < F(0.3333), 0, np.where(rx32 > F(0.6666), 2, 1))
        t = np.clip((rx - lo) / (hi - lo), 0.0, 1.0)
        pick = lambda a: np.take_along_axis(a, k[None, :], axis=0)[0]
        d, w1, w2 = pick(self.d), pick(self.c1), pick(self.c2)
        w = w1 / (w1 + w2 * 3.0)
        with np.errstate(all="ignore"):
            tail = t > w
            tt = np.where(tail, np.clip((t - w) / (1.0 - w), 0, 1), np.clip(t / w, 0, 1))
            r = np.where(tail, np.log(1.0 - tt * w2) * (-d * 3.0), np.log(1.0 - tt * w1) * (-d))
        return np.where((self.maxR < EPS) | (d < EPS), 0.0, r)

    def pdf(self, r):
        d = np.maximum(self.d, EPS)
        with np.errstate(all="ignore"):
            s = ((np.exp(-r / d) + np.exp(-r / d / 3.0)) / d / (self.c1 + self.c2 * 3.0)).sum(axis=0)
            p = s / (2.0 * np.pi * r * 3.0)
        return np.where(self.maxR < EPS, 1.0, p)

    def profile(self, r):
        with np.errstate(all="ignore"):
            v = (np.exp(-r / self.d) + np.exp(-r / (3.0 * self.d))) / (8.0 * np.pi * r * self.d)
        v = np.where(self.d < EPS, 1.0, v)
        return np.where(self.maxR < EPS, 0.0, np.where(r < EPS, 1.0, v))


def _lerp(t, a, b):
    return (1.0 - t) * a + t * b


class Disney64:
    """DisneySampler, src/rlDisney.cpp:155-192 (constructor), 199-236 (evalDiffuse), 318-357 (evalSpecular), 359-404
    (samplers), 515-577 (pdfs, D_GTR1, D_GTR2Aniso, smithG_GGX); tangent = mAxisU is an input (AiBuildLocalFramePolar is closed)"""

    def __init__(self, c):
        self.wo, self.N, self.U = (np.asarray(c[k], np.float64) for k in ("wo", "N", "T"))
        n = self.wo.shape[1]
        g = lambda k, d=0.0: np.broadcast_to(np.asarray(c.get(k, d), np.float64), (n,))
        base = np.asarray(c.get("base_color", (1.0, 1.0, 1.0)), np.float64)
        self.base = base if base.ndim == 2 else np.repeat(base[:, None], n, axis=1)
        self.V = np.cross(self.N.T, self.U.T).T
        self.rough, self.subsurface, self.metallic = g("roughness"), g("subsurface"), g("metallic")
        self.specular = g("specular") * float(F(0.08))
        self.clearcoat, self.gloss = g("clearcoat") * 0.25, g("clearcoat_gloss")
        aspect = np.sqrt(1.0 - g("anisotropic") * float(F(0.9)))
        self.ax = np.maximum(float(F(1e-2)), self.rough ** 2 / aspect)
        self.ay = np.maximum(float(F(1e-2)), self.rough ** 2 * aspect)
        self.srough = self.rough ** 2
        lum = self.base[0] * float(F(0.212671)) + self.base[1] * float(F(0.715160)) + self.base[2] * float(F(0.072169))
        tint = np.where(lum > 0, self.base / np.where(lum > 0, lum, 1.0), 1.0)
        metallic_color = self.specular * _lerp(g("specular_tint"), 1.0, tint)
        self.f0 = _lerp(self.metallic, metallic_color, self.base)
        self.sheen_color = _lerp(g("sheen_tint"), 1.0, tint) * g("sheen")

    def _d_gtr2(self, m, mn2):
        hu, hv = (m * self.U).sum(0), (m * self.V).sum(0)
        return (1.0 / np.pi) / (self.ax * self.ay * ((hu / self.ax) ** 2 + (hv / self.ay) ** 2 + mn2) ** 2)

    def _d_gtr1(self, mn2):
        a2 = _lerp(self.gloss, float(F(0.1)), float(F(0.001))) ** 2
        return (a2 - 1.0) / np.pi / (np.log(a2) * (1.0 + (a2 - 1.0) * mn2))

    @staticmethod
    def _smith(nv, a):
        return 1.0 / (nv + np.sqrt(a * a + nv * nv - a * a * nv * nv))

    def eval_diffuse(self, L):
        ln, vn = (L * self.N).sum(0), (self.wo * self.N).sum(0)
        H = _norm(L + self.wo)
        lh, vh = (L * H).sum(0), (self.wo * H).sum(0)                    # the reference names V.H "NdotH" (src/rlDisney.cpp:210)
        black = (ln < EPS) | (vn < EPS) | (vh < EPS) | (lh < EPS)
        fl, fv = np.clip(1.0 - ln, 0, 1) ** 5, np.clip(1.0 - vn, 0, 1) ** 5
        f90 = 0.5 + 2.0 * self.rough * lh * lh
        diffuse = _lerp(fl, 1.0, f90) * _lerp(fv, 1.0, f90)
        fss90 = self.rough * lh * lh
        fss = _lerp(fl, 1.0, fss90) * _lerp(fv, 1.0, fss90)
        with np.errstate(all="ignore"):
            ss = 1.25 * (fss * (1.0 / (ln + vn) - 0.5) + 0.5)
        out = self.base / np.pi * _lerp(self.subsurface, diffuse, ss) * (1.0 - self.metallic)
        return np.where(black, 0.0, out) * ln                           # evalBrdf multiplies by N.L (src/rlDisney.cpp:130-134)

    def eval_specular(self, L):
        ln, vn = (L * self.N).sum(0), (self.wo * self.N).sum(0)
        M = _norm(L + self.wo)
        lm, nm = (L * M).sum(0), (self.N * M).sum(0)
        black = (ln < EPS) | (vn < EPS) | (nm < EPS) | (lm < EPS)
        nm2 = nm * nm
        fh = np.clip(1.0 - lm, 0, 1) ** 5
        with np.errstate(all="ignore"):
            ds, fs = self._d_gtr2(M, nm2), _lerp(fh, self.f0, 1.0)
            gs = self._smith(ln, self.srough) * self._smith(vn, self.srough)
            dr, fr = self._d_gtr1(nm2), _lerp(fh, 0.04, 1.0)
            gr = self._smith(ln, 0.25) * self._smith(vn, 0.25)
            out = ds * fs * gs + self.clearcoat * dr * fr * gr + fh * self.sheen_color * (1.0 - self.metallic)
        return np.where(black, 0.0, out) * ln

    def pdf_diffuse(self, L):
        return np.maximum(EPS, (L * self.N).sum(0) / np.pi)

    def pdf_specular(self, L):
        m = _norm(L + self.wo)
        im, mn = np.abs((L * m).sum(0)), (m * self.N).sum(0)
        cw = self.clearcoat / (self.clearcoat + 1.0)
        vn = np.maximum(1e-4, (self.wo * self.N).sum(0))
        with np.errstate(all="ignore"):
            dw = self._smith(im, self.srough) * self._d_gtr2(m, mn * mn) * 2.0 * im / vn
            d = _lerp(cw, dw, self._d_gtr1(mn * mn) * np.abs(mn) / im)
        return np.where(mn < 0, 0.0, d * 0.25)
