"""Oracle checks for the rlGgx direct-lighting loop (src/rlGgx.cpp:274-299) with its documented stand-ins
for the closed light services (parity unpinned): a spherical light, Oren-Nayar diffuse, power-heuristic
MIS.  Anchors: the Lambertian irradiance of a small distant light in closed form, and the classical MIS
consistency test -- light sampling only, BSDF sampling only and their MIS combination estimate the same
integral, which only holds if GgxSampler's sample / eval / pdf triple is self-consistent."""
import math

import numpy as np
import pytest

import oracle_lib as O


def _points(n, wo, rng=None):
    N = np.zeros((3, n), np.float32); N[2] = 1
    T = np.zeros((3, n), np.float32); T[0] = 1
    if rng is not None:
        a = rng.uniform(0, 2 * math.pi, n)
        T[0], T[1] = np.cos(a), np.sin(a)
    w = np.asarray(wo, np.float64); w /= np.linalg.norm(w)
    WO = np.repeat(w.astype(np.float32)[:, None], n, axis=1)
    return np.ascontiguousarray(WO), N, T, np.zeros((3, n), np.float32)


def test_lambert_small_distant_light_closed_form():
    n = 256
    wo, N, T, P = _points(n, (0.3, 0.1, 0.9))
    g = O.Ggx(wo, N, T, roughness=0.5, ior=1.5, nthreads=4)
    c = np.array([3.0, 4.0, 12.0]); R = 0.05                      # |c| = 13
    lt = O.make_light(center=tuple(c), radius=R, radiance=(2.0, 1.0, 0.5), mis_mode=1)
    dd, _ = g.direct_lighting(P, lt, 4, 1, Kd_color=(0.8, 0.6, 0.4), Kd=0.5, Kd_roughness=0.0)
    dist = 13.0
    omega = 2 * math.pi * (1 - math.sqrt(1 - (R / dist) ** 2))
    want = [rad * kdc * 0.5 / math.pi * (c[2] / dist) * omega for rad, kdc in zip((2.0, 1.0, 0.5), (0.8, 0.6, 0.4))]
    np.testing.assert_allclose(dd.astype(np.float64).mean(axis=1), want, rtol=2e-3)


def test_oren_nayar_reduces_to_lambert_and_is_reciprocal():
    rng = np.random.default_rng(1)
    lib = O.lib()
    import ctypes as C
    class ON(C.Structure):
        _fields_ = [("N", O.V3), ("T", O.V3), ("A", C.c_float), ("B", C.c_float)]
    lib.orc_oren_nayar_brdf.restype = C.c_float
    lib.orc_oren_nayar_brdf.argtypes = [C.POINTER(ON), O.V3, O.V3]
    lib.orc_oren_nayar_init.argtypes = [C.POINTER(ON), O.V3, O.V3, C.c_float]
    for sigma in (0.0, 0.3, 1.0):
        on = ON()
        lib.orc_oren_nayar_init(C.byref(on), O.V3(0, 0, 1), O.V3(1, 0, 0), sigma)
        for _ in range(50):
            a, b = rng.normal(size=3), rng.normal(size=3)
            a[2], b[2] = abs(a[2]) + 0.05, abs(b[2]) + 0.05
            a /= np.linalg.norm(a); b /= np.linalg.norm(b)
            fab = lib.orc_oren_nayar_brdf(C.byref(on), O.V3(*a), O.V3(*b))     # x cos(b)
            fba = lib.orc_oren_nayar_brdf(C.byref(on), O.V3(*b), O.V3(*a))     # x cos(a)
            assert abs(fab / b[2] - fba / a[2]) < 1e-6                          # Helmholtz reciprocity
            if sigma == 0.0:
                assert abs(fab - b[2] / math.pi) < 1e-7
            assert fab >= 0
        assert lib.orc_oren_nayar_brdf(C.byref(on), O.V3(0, 0, 1), O.V3(0.6, 0, -0.8)) == 0.0


@pytest.mark.parametrize("rough,radius", [(0.15, 1.5), (0.4, 0.6), (0.8, 3.0)])
def test_mis_consistency_of_the_ggx_triple(rough, radius):
    rng = np.random.default_rng(3)
    n = 8192
    wo, N, T, P = _points(n, (0.5, 0.0, 0.85), rng)
    # the light sits around the mirror direction, where the specular lobe is
    centre = (-2.0, 0.3, 3.5)
    g = O.Ggx(wo, N, T, KsColor=(0.9, 0.8, 0.7), roughness=rough, ior=1.6, anisotropic=0.4, nthreads=O.hardware_threads())
    out = {}
    for mode in (0, 1, 2):
        lt = O.make_light(center=centre, radius=radius, radiance=(1.0, 1.0, 1.0), mis_mode=mode)
        dd, ds = g.direct_lighting(P, lt, 4, 9, Kd_color=(1, 1, 1), Kd=1.0, Kd_roughness=0.5, Ks=1.0)
        out[mode] = (dd.astype(np.float64).mean(axis=1), ds.astype(np.float64).mean(axis=1))
        assert np.isfinite(dd).all() and np.isfinite(ds).all() and (dd >= 0).all() and (ds >= 0).all()
    for mode in (1, 2):
        np.testing.assert_allclose(out[mode][0], out[0][0], rtol=0.03)      # Oren-Nayar lobe
        np.testing.assert_allclose(out[mode][1], out[0][1], rtol=0.04)      # GGX lobe
    assert out[0][1].min() > 0


def test_inside_the_light_and_below_the_horizon_are_black():
    n = 64
    wo, N, T, P = _points(n, (0.0, 0.0, 1.0))
    g = O.Ggx(wo, N, T, roughness=0.5, ior=1.5)
    dd, ds = g.direct_lighting(P, O.make_light(center=(0, 0, 0.5), radius=1.0), 2, 1)
    assert not dd.any() and not ds.any()
    dd, ds = g.direct_lighting(P, O.make_light(center=(0, 0, -5.0), radius=1.0), 2, 1)
    assert not dd.any() and not ds.any()
