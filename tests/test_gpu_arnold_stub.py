"""GPU: all three node paths of the generated Arnold-side stub (rl_arnold_stub.hpp: GgxNode, DisneyNode, SkinNode) --
per shading point `add(globals, evaluator, xi)` with a table-lookup evaluator in the shape of AiShaderEvalParam*, one
`flush()` per node through the C++ mirror and the C ABI -- against the oracle on the same shading points."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases
from gpu_util import disney_oracle, ggx_oracle

ROOT = Path(__file__).resolve().parent.parent
FIX = json.loads((ROOT / "tests" / "golden" / "param_surface.json").read_text())["nodes"]


def _table(node, n, values):
    """3 rows per positional parameter id, in the reference's enum order (what AiShaderEvalParam*(pid) would return)"""
    t = np.zeros((3 * len(FIX[node]["parameters"]), n), np.float32)
    for pid, p in enumerate(FIX[node]["parameters"]):
        if p["type"] == "STR":
            continue
        v = values.get(p["name"], p["default"])                      # parameters the test does not vary: node defaults
        v = np.asarray(v, np.float32)
        if v.ndim == 1 and v.shape[0] == n and p["type"] in ("FLT", "BOOL"):
            t[3 * pid] = v
        elif v.ndim == 2:
            t[3 * pid:3 * pid + 3] = v
        else:
            for c in range(v.size):
                t[3 * pid + c] = v.flat[c]
    return t


@pytest.mark.gpu
@pytest.mark.parametrize("parameters", ["textured", "constant", "instances"])
def test_three_nodes_through_the_stub(oracle, tmp_path, parameters):
    """textured: every parameter differs from point to point (planes).  constant: every parameter evaluates to the same value
    at every point, as on a node without linked textures -- the stub notices (detail::Par1 / Par3), uploads nothing for them
    and the library runs its UNIFORM_ALL kernels; instances: the batch mixes the hits of 23 node instances, each with its own
    constant parameters -- the stub finds the distinct parameter rows and sends them by reference (detail::Materials,
    rls_material_index: one id per point, one table entry per instance).  The results must be the oracle's all the same."""
    from rlshaders_amd import build
    build.build_library()
    build.build_host_examples()
    exe = build.OBJDIR / "test_arnold_stub"
    n = 3000
    g = cases.ggx_mixed(cases.SEED_PARITY, n)
    d = cases.disney_mixed(cases.SEED_PARITY, n)
    s = cases.skin_mixed(cases.SEED_PARITY, n)
    if parameters == "instances":
        inst = np.random.default_rng(4).integers(0, 23, n)
        inst[:64] = np.repeat(np.arange(23), 3)[:64]                      # runs of equal ids, as consecutive hits on a node are
        gi, di, si = cases.ggx_mixed(cases.SEED_EDGE, 23), cases.disney_mixed(cases.SEED_EDGE, 23), cases.skin_mixed(cases.SEED_EDGE, 23)
        for k in ("KsColor", "ior", "roughness", "anisotropic"):
            g[k] = np.ascontiguousarray(gi[k][..., inst])
        for k in ("base_color",) + tuple(oracle.DISNEY_SCALARS):
            d[k] = np.ascontiguousarray(di[k][..., inst])
        s["params"] = {k: np.ascontiguousarray(v[..., inst]) for k, v in si["params"].items()}
    if parameters == "constant":
        full = lambda v: (np.repeat(np.asarray(v, np.float32)[:, None], n, axis=1) if np.ndim(v) else np.full(n, v, np.float32))
        g.update({k: full(v) for k, v in dict(KsColor=(0.9, 0.8, 0.7), ior=1.5, roughness=0.3, anisotropic=0.6).items()})
        d.update({k: full(v) for k, v in dict(base_color=(0.85, 0.7047, 0.2057), subsurface=0.2, metallic=0.3, specular=0.5,
                                              specular_tint=0.25, roughness=0.4, anisotropic=0.4, sheen=0.5, sheen_tint=0.5,
                                              clearcoat=0.6, clearcoat_gloss=0.7).items()})
        s["params"] = {k: full(v) for k, v in dict(cases.SKIN_DEFAULTS, sheen_weight=0.3, sss_scatter_dist=(1.0, 0.6, 0.35),
                                                   sss_color=(1.0, 0.84, 0.5)).items()}
    xi = cases.xi(cases.SEED_PARITY, n, 6)
    wo, N, T = g["wo"], g["N"], g["T"]                  # the three case sets share their frames (same seed)
    assert np.array_equal(wo, d["wo"]) and np.array_equal(wo, s["wo"])
    geo = np.concatenate([-wo, N, N, T])                # sg->Rd = -wo, sg->N = sg->Nf = N, U = T
    tg = _table("rlGgx", n, dict(KsColor=g["KsColor"], ior=g["ior"], specularRoughness=g["roughness"],
                                 anisotropic=g["anisotropic"]))
    td = _table("rlDisney", n, {k: d[k] for k in ("base_color",) + tuple(oracle.DISNEY_SCALARS)})
    ts = _table("rlSkin", n, s["params"])
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        np.array([n], np.float32).tofile(f)
        for a in (geo, xi, tg, td, ts):
            np.ascontiguousarray(a, np.float32).tofile(f)
    outp = tmp_path / "out.bin"
    p = subprocess.run([str(exe), "run", str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    info = json.loads(p.stdout.strip().splitlines()[-1])
    # rlGgx: KsColor + 3 scalars, rlDisney: base_color + 10, rlSkin: 3 colours + the scatter distance + 8 scalars
    assert info == {"n": n, "ggx_planes": 8, "disney_planes": 14, "skin_planes": 24,
                    "uniform_parameters": 4 + 11 + 12 if parameters == "constant" else 0,
                    "reference_batches": 3 if parameters == "instances" else 0}
    out = np.fromfile(outp, np.float32).reshape(-1, n)
    og, od, osk = out[:8], out[8:22], out[22:]
    # rlGgx: evalSample -> evalBrdf -> evalPdf (+ the Fresnel side effect)
    rwi, rf, rpdf, rF = ggx_oracle(oracle, g).sample_eval_pdf(xi[0], xi[1])
    for name, a, b in (("wi", og[0:3], rwi), ("f", og[3:6], rf), ("pdf", og[6], rpdf), ("fresnel", og[7], rF)):
        cases.assert_tight(cases.summarize(cases.rel_err(a, b)), ("rlGgx", name))
    # rlDisney: both lobes
    o = disney_oracle(oracle, d)
    for l, lobe in enumerate((oracle.RAY_DIFFUSE, oracle.RAY_GLOSSY)):
        rwi, rf, rpdf = o.sample_eval_pdf(lobe, xi[0], xi[1])
        blk = od[7 * l:7 * l + 7]
        for name, a, b in (("wi", blk[0:3], rwi), ("f", blk[3:6], rf), ("pdf", blk[6], rpdf)):
            cases.assert_tight(cases.summarize(cases.rel_err(a, b)), ("rlDisney", lobe, name))
    # rlSkin: the 24 planes of the composite
    ref = oracle.skin(wo, N, T, s["params"], xi, nthreads=4)
    order = ("sheen_wi", "sheen_f", "sheen_pdf", "sheen_fresnel", "spec_wi", "spec_f", "spec_pdf", "spec_fresnel",
             "r", "r_pdf", "profile", "sheenFresnel", "specularFresnel", "sssWeight")
    row = 0
    for k in order:
        rows = 3 if ref[k].ndim == 2 else 1
        a = osk[row:row + rows] if rows == 3 else osk[row]
        cases.assert_tight(cases.summarize(cases.rel_err(a, ref[k])), ("rlSkin", k))
        row += rows
    assert row == 24


@pytest.mark.gpu
@pytest.mark.parametrize("parameters", ["textured", "instances"])
def test_whole_shader_evaluate_through_the_stub(oracle, tmp_path, parameters):
    """`addShade(globals, evaluator)` per shading point and one `shade()` per node: the whole shader_evaluate of rlGgx,
    rlDisney and rlSkin (rls_ggx_shade, rls_disney_shade, rls_skin_integrate) through the generated stub, against the
    oracle's shader_evaluate on the same points, lights, environment and sample seeds"""
    import os
    from rlshaders_amd import build
    build.build_library()
    build.build_host_examples()
    exe = build.OBJDIR / "test_arnold_stub"
    n = 2000
    g = cases.ggx_mixed(cases.SEED_PARITY, n)
    d = cases.disney_mixed(cases.SEED_PARITY, n)
    s = cases.skin_mixed(cases.SEED_PARITY, n)
    wo, N, T = g["wo"], g["N"], g["T"]
    P = N.copy()                                            # shading points on the unit sphere (the skin scene's surface)
    geo = np.concatenate([-wo, N, N, T, P])
    u = lambda j: oracle.gen_uniform(cases.SEED_PARITY, 0, n, oracle.S_PARAM0 + j)
    shader = dict(KdColor=np.stack([u(0), u(1), u(2)]), Kd=u(3), diffuseRoughness=u(4), Ks=u(5),
                  KtColor=np.stack([u(6), u(7), u(8)]), Kt=u(9))
    if parameters == "instances":             # the hits of 19 node instances: every parameter constant per instance
        inst = np.random.default_rng(8).integers(0, 19, n)
        pick = lambda v: np.ascontiguousarray(v[..., inst])
        gi, di, si = cases.ggx_mixed(cases.SEED_EDGE, 19), cases.disney_mixed(cases.SEED_EDGE, 19), cases.skin_mixed(cases.SEED_EDGE, 19)
        for k in ("KsColor", "ior", "roughness", "anisotropic"):
            g[k] = pick(gi[k])
        for k in ("base_color",) + tuple(oracle.DISNEY_SCALARS):
            d[k] = pick(di[k])
        s["params"] = {k: pick(v) for k, v in si["params"].items()}
        shader = {k: pick(v[..., :19]) for k, v in shader.items()}
    tg = _table("rlGgx", n, dict(KsColor=g["KsColor"], ior=g["ior"], specularRoughness=g["roughness"],
                                 anisotropic=g["anisotropic"], **shader))
    td = _table("rlDisney", n, {k: d[k] for k in ("base_color",) + tuple(oracle.DISNEY_SCALARS)})
    sp = dict(s["params"], sss_scatter_dist=(s["params"]["sss_scatter_dist"] * np.float32(0.1)).astype(np.float32))
    ts = _table("rlSkin", n, sp)
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        np.array([n], np.float32).tofile(f)
        for a in (geo, tg, td, ts):
            np.ascontiguousarray(a, np.float32).tofile(f)
    outp = tmp_path / "out.bin"
    env = dict(os.environ, RLS_INTEGRATE_GROUP="1")         # one lane per point: the reference's summation order
    p = subprocess.run([str(exe), "shade", str(inp), str(outp)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr
    info = json.loads(p.stdout.strip().splitlines()[-1])
    assert info == {"n": n, "ggx_planes": 18, "disney_planes": 15, "skin_planes": 15, "uniform_parameters": 0,
                    "reference_batches": 3 if parameters == "instances" else 0}
    out = np.fromfile(outp, np.float32).reshape(-1, n)
    og, od, osk = out[:18], out[18:33], out[33:]
    lights = [oracle.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0), mis_mode=0),
              oracle.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=2)]
    envc = (0.7, 0.8, 0.9)
    ref = ggx_oracle(oracle, g).shade(P, lights, 3, 77, Kd_color=shader["KdColor"], Kd=shader["Kd"],
                                      Kd_roughness=shader["diffuseRoughness"], Ks=shader["Ks"], Kt_color=shader["KtColor"],
                                      Kt=shader["Kt"], env=envc, traced=True)
    for j, k in enumerate(("direct_diffuse", "direct_specular", "refraction", "indirect_diffuse", "indirect_specular", "out")):
        cases.assert_tight(cases.summarize(cases.rel_err(og[3 * j:3 * j + 3], ref[k])), ("rlGgx", k))
    ref = disney_oracle(oracle, d).shade(P, lights, 3, 77, env=envc)
    for j, k in enumerate(("direct_diffuse", "direct_specular", "indirect_diffuse", "indirect_specular", "out")):
        cases.assert_tight(cases.summarize(cases.rel_err(od[3 * j:3 * j + 3], ref[k])), ("rlDisney", k))
    scene = oracle.make_scene(geometry="sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    ref = oracle.skin_integrate(wo, N, T, sp, P, scene, 3, 77, env=envc, nthreads=4, lights=lights[:1])
    for j, k in enumerate(("sheen", "specular", "sss", "out")):
        cases.assert_tight(cases.summarize(cases.rel_err(osk[3 * j:3 * j + 3], ref[k])), ("rlSkin", k))
    for j, k in enumerate(("sheenFresnel", "specularFresnel", "sssWeight")):
        cases.assert_tight(cases.summarize(cases.rel_err(osk[12 + j], ref[k])), ("rlSkin", k))
    assert (ref["sss"] > 0).mean() > 0.2
