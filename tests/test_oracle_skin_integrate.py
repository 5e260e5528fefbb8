"""Oracle (CPU): rlSkin's shader_evaluate over n^2 samples per layer (src/rlSkin.cpp:174-254) and integrateRefract
(src/rlGgx.h:205-245): identities that hold whatever the closed renderer services are replaced by."""
import numpy as np

import cases
import oracle_lib as O
from gpu_util_cpu import ggx_oracle

SCENE = dict(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)


def test_skin_hand_down_identities():
    n, spp_n, seed = 512, 4, 11
    c = cases.skin_mixed(cases.SEED_PARITY, n)
    p = c["params"]
    sc = O.make_scene(**SCENE)
    r = O.skin_integrate(c["wo"], c["N"], c["T"], p, c["N"], sc, spp_n, seed, env=(1.0, 0.9, 0.8))
    # the sheen lobe is the first closure of the point: its mean Fresnel is getAvgReflectWeight of the same n^2 samples
    # (sampler dimension pair 0, the one rlGgx's own integrator uses), src/rlSkin.cpp:204
    g = O.Ggx(c["wo"], c["N"], c["T"], KsColor=p["sheen_color"], ior=p["sheen_ior"], roughness=p["sheen_roughness"],
              anisotropic=0.0)
    s, avg = g.integrate(spp_n, seed)
    on = p["sheen_weight"] > 1e-4
    assert np.array_equal((avg * p["sheen_weight"])[on].view(np.uint32), r["sheenFresnel"][on].view(np.uint32))
    inv = np.float32(1.0) / np.float32(spp_n * spp_n)
    env = np.array([1.0, 0.9, 0.8], np.float32)[:, None]
    want = (s * inv * env) * p["sheen_weight"]
    assert np.array_equal(want[:, on].view(np.uint32), r["sheen"][:, on].view(np.uint32))
    assert (r["sheenFresnel"][~on] == 0).all() and (r["sheen"][:, ~on] == 0).all()
    # :238 and :254
    one = np.float32(1.0)
    w = p["sss_weight"] * (one - r["specularFresnel"] * (one - r["sheenFresnel"]))
    assert np.array_equal(w.view(np.uint32), r["sssWeight"].view(np.uint32))
    assert np.array_equal(((r["sheen"] + r["specular"]) + r["sss"]).view(np.uint32), r["out"].view(np.uint32))
    # Fresnel means are means of values in [0, 1]; the layer below never gains energy from the one above
    assert (r["sheenFresnel"] >= 0).all() and (r["sheenFresnel"] <= p["sheen_weight"] + 1e-6).all()
    assert (r["sssWeight"] <= p["sss_weight"] + 1e-6).all()


def test_skin_layers_switch_off():
    n, spp_n, seed = 256, 4, 5
    c = cases.skin_mixed(cases.SEED_PARITY, n)
    sc = O.make_scene(**SCENE)
    p = dict(c["params"], sheen_weight=0.0, specular_weight=0.0)
    r = O.skin_integrate(c["wo"], c["N"], c["T"], p, c["N"], sc, spp_n, seed)
    assert (r["sheen"] == 0).all() and (r["specular"] == 0).all()
    assert (r["sheenFresnel"] == 0).all() and (r["specularFresnel"] == 0).all()
    assert np.array_equal(r["sssWeight"], O._full(p["sss_weight"], n))          # nothing reflected above the SSS layer
    assert np.array_equal(r["out"].view(np.uint32), r["sss"].view(np.uint32))
    # a black specular colour: integrateGlossy returns without sampling (src/rlGgx.h:174-176), the closure has drawn no
    # sample and getAvgReflectWeight() is 1 (181-184): the layer below gets sss_weight * (1 - specular_weight)
    p = dict(c["params"], sheen_weight=0.0, specular_color=(0.0, 0.0, 0.0))
    r = O.skin_integrate(c["wo"], c["N"], c["T"], p, c["N"], sc, spp_n, seed)
    on = p["specular_weight"] > 1e-4
    assert np.array_equal(r["specularFresnel"][on], p["specular_weight"][on]) and (r["specular"] == 0).all()
    # sss_weight 0: black, the probe rays are never traced (:244)
    p = dict(c["params"], sss_weight=0.0)
    r = O.skin_integrate(c["wo"], c["N"], c["T"], p, c["N"], sc, spp_n, seed)
    assert (r["sss"] == 0).all()


def test_integrate_refract_closed_forms():
    n, spp_n, seed = 512, 4, 3
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    # ior 1: no bending.  Untraced branch (src/rlGgx.h:213-222): dir = -view, weight = SQR(1) * |N . view|
    c1 = dict(c, ior=1.0)
    res, tir = ggx_oracle(O, c1).integrate_refract(spp_n, seed, traced=False, env=(2.0, 1.0, 0.5))
    nv = np.abs((c["N"] * c["wo"]).sum(axis=0))
    assert (tir == 0).all()
    assert np.abs(res[1] - nv).max() < 1e-6 and np.abs(res[0] - 2 * res[1]).max() < 1e-6
    assert np.abs(res[2] - 0.5 * res[1]).max() < 1e-6
    # traced branch: the mean of getSampleWeight over the same n^2 samples the per-sample entry point sees
    g = ggx_oracle(O, c)
    res, tir = g.integrate_refract(spp_n, seed, traced=True)
    acc = np.zeros(n, np.float32)
    for s in range(spp_n * spp_n):
        xi = np.array([O.sample_02(seed, i, 0, s) for i in range(n)], np.float32).T
        _, w, _ = g.refract(np.ascontiguousarray(xi[0]), np.ascontiguousarray(xi[1]))
        acc += w
    assert np.array_equal((acc * (np.float32(1.0) / np.float32(spp_n * spp_n))).view(np.uint32), res[0].view(np.uint32))
    assert (tir == 0).all()                                       # entering a denser medium never reflects totally
    # leaving a dense medium at a grazing angle does
    ex = np.ones(n, np.uint8)
    _, tir = ggx_oracle(O, dict(c, ior=1.5), exiting=ex).integrate_refract(spp_n, seed, traced=True)
    assert 0.0 < tir.mean() < 1.0 and (tir >= 0).all() and (tir <= 1).all()
    _, tir1 = ggx_oracle(O, dict(c, ior=1.5), exiting=ex).integrate_refract(spp_n, seed, traced=False)
    cos_crit = np.sqrt(1 - (1 / 1.5) ** 2)
    assert np.array_equal(tir1 > 0, np.abs((c["N"] * c["wo"]).sum(axis=0)) < cos_crit) or \
        np.mean((tir1 > 0) != (np.abs((c["N"] * c["wo"]).sum(axis=0)) < cos_crit)) < 0.01
