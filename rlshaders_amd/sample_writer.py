"""Lat-long sample / radiance images of a closure: the host-side mirror of the reference's debug
``SampleWriter`` (src/rlUtil.h:43-171), minus the EXR writer -- images are numpy arrays, written as
binary PPM / .npy.  The closure under test is evaluated on the GPU through the C ABI like everything
else; torch does the binning.

On top of what the reference draws, ``compare()`` turns the two images into a number: the sample
histogram against the pdf integrated over each lat-long bin (a chi-square statistic per degree of
freedom), i.e. a check that ``evalSample`` really draws from ``evalPdf``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .closures import Context


class SampleWriter:
    """``SampleWriter(w, h)``: w azimuth bins over [0, 2 pi), h polar bins over [0, pi/2) about +z
    (src/rlUtil.h:105-110,139-140).  The closure must be built on the canonical frame N = +z, T = +x."""

    def __init__(self, ctx: Context, w: int = 128, h: int = 32):
        self.ctx, self.w, self.h = ctx, int(w), int(h)

    # ---- the canonical frame, replicated n times ------------------------------------------------
    def frame(self, n: int, wo):
        dev = self.ctx.torch_device
        v = torch.tensor(wo, dtype=torch.float32, device=dev)
        v = v / v.norm()
        WO = v.reshape(3, 1).repeat(1, n).contiguous()
        N = torch.zeros(3, n, device=dev); N[2] = 1
        T = torch.zeros(3, n, device=dev); T[0] = 1
        return WO, N, T

    # ---- writeRadiance (src/rlUtil.h:98-113): evalBrdf over the grid -----------------------------
    def grid_directions(self, sub: int = 1) -> torch.Tensor:
        """bin-centre directions of a (h*sub) x (w*sub) grid -> [3, h*sub*w*sub]"""
        dev = self.ctx.torch_device
        H, W = self.h * sub, self.w * sub
        th = (torch.arange(H, device=dev, dtype=torch.float32) + 0.5) * (0.5 * math.pi / H)
        ph = (torch.arange(W, device=dev, dtype=torch.float32) + 0.5) * (2.0 * math.pi / W)
        st, ct = torch.sin(th)[:, None], torch.cos(th)[:, None]
        d = torch.stack([st * torch.cos(ph)[None, :], st * torch.sin(ph)[None, :], ct.expand(H, W)])
        return d.reshape(3, -1).contiguous()

    def writeRadiance(self, make_closure, sub: int = 1) -> np.ndarray:
        """``make_closure(n)`` -> an object with evalBrdf([3,n]); returns the [h*sub, w*sub, 3] image of f."""
        d = self.grid_directions(sub)
        f = make_closure(d.shape[1]).evalBrdf(d)
        return f.reshape(3, self.h * sub, self.w * sub).permute(1, 2, 0).cpu().numpy()

    def pdf_image(self, make_closure, sub: int = 16) -> np.ndarray:
        """probability of each (h x w) bin: evalPdf on a sub x sub grid per bin times the solid angle"""
        d = self.grid_directions(sub)
        p = make_closure(d.shape[1]).evalPdf(d).reshape(self.h * sub, self.w * sub).double()
        H, W = self.h * sub, self.w * sub
        th = (torch.arange(H, device=p.device, dtype=torch.float64) + 0.5) * (0.5 * math.pi / H)
        domega = torch.sin(th)[:, None] * (0.5 * math.pi / H) * (2.0 * math.pi / W)
        cell = (p * domega).reshape(self.h, sub, self.w, sub).sum(dim=(1, 3))
        return cell.cpu().numpy()

    # ---- writeSample (src/rlUtil.h:115-156): histogram of evalSample ------------------------------
    def writeSample(self, make_closure, count: int, seed: int = 1):
        """-> (histogram [h, w] of samples in the upper hemisphere, number below it ("missing"),
        number of invalid (zero-vector) samples)"""
        g = torch.Generator(device=self.ctx.torch_device)
        g.manual_seed(seed)
        rx = torch.rand(count, device=self.ctx.torch_device, generator=g)
        ry = torch.rand(count, device=self.ctx.torch_device, generator=g)
        out = make_closure(count).evalSample(rx, ry)
        wi = out[0] if isinstance(out, tuple) else out
        zero = (wi == 0).all(dim=0)
        z = wi[2].clamp(-1.0, 1.0)
        theta = torch.acos(z)
        phi = torch.atan2(wi[1], wi[0])
        phi = torch.where(phi < 0, phi + 2.0 * math.pi, phi)
        below = (theta > 0.5 * math.pi) & ~zero
        ok = ~zero & ~below
        i = (phi * (self.w / (2.0 * math.pi))).long().clamp(0, self.w - 1)
        j = (theta * (self.h / (0.5 * math.pi))).long().clamp(0, self.h - 1)
        hist = torch.bincount((j * self.w + i)[ok], minlength=self.w * self.h).reshape(self.h, self.w)
        return hist.cpu().numpy(), int(below.sum()), int(zero.sum())

    # ---- sample density against pdf -------------------------------------------------------------
    def compare(self, make_closure, count: int, seed: int = 1, min_expected: float = 50.0, sub: int = 16) -> dict:
        hist, below, zero = self.writeSample(make_closure, count, seed)
        prob = self.pdf_image(make_closure, sub)
        expected = prob * count
        use = expected >= min_expected
        chi2 = float((((hist - expected) ** 2) / np.maximum(expected, 1e-30))[use].sum())
        dof = int(use.sum())
        big = expected >= 1.0e4            # bins whose counting noise is below 1 %
        dev = float(np.abs(hist[big] / expected[big] - 1.0).max()) if big.any() else 0.0
        return dict(chi2_per_dof=chi2 / max(dof, 1), dof=dof, max_rel_dev=dev, big_bins=int(big.sum()), pdf_mass=float(prob.sum()),
                    sampled_mass=float(hist.sum()) / count, covered_mass=float(prob[use].sum()),
                    below_horizon=below, invalid=zero, count=count)


def write_ppm(path: str, img: np.ndarray, scale: float | None = None) -> None:
    """[h, w] or [h, w, 3] float image -> binary PPM, linear scale to the 99.5th percentile"""
    a = np.asarray(img, dtype=np.float64)
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, axis=2)
    s = scale if scale is not None else max(np.quantile(a, 0.995), 1e-30)
    b = (np.clip(a / s, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (b.shape[1], b.shape[0]))
        f.write(b.tobytes())
