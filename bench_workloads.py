"""bench_workloads.py -- the workload table of bench.py: which closure calls one "step" of each named workload makes, on
which planes, with how many algorithmic bytes per shading point.  bench.py is the driver (arguments, timing, roofline and
CPU-baseline records, the JSON lines); tools/ import `make_workload` from here as well.

Every workload generates its inputs on the device (rls_gen_*: the counter-based hash of SURVEY.md 8(d), the same numbers the
CPU oracle generates for the CPU leg) into ONE arena (rlshaders_amd.Arena) and returns a `Workload` whose `launch()` issues
the kernel launches of one step.  `kernel` is the name rocprofv3 --kernel-trace gives the dominant kernel ({m}: 0 EXACT, 1 FAST).
"""
from __future__ import annotations

import os
import time

SEED = 1234               # throughput seed, SURVEY.md 8(d)

# stream ids (shared with oracle/rls_oracle.h)
S_ROUGH, S_IOR = 5, 6
S_KS = 8
S_XI0 = 11
S_PARAM0 = 32


WORKLOADS = ["ggx_reflect_refract", "ggx_reflect_refract_uniform", "ggx_reflect_refract_materials", "ggx_reflect", "ggx_eval", "ggx_pdf", "ggx_direct",
             "ggx_shade", "disney_direct", "disney_shade", "disney_integrate", "disney_stream", "disney_triple_diffuse",
             "disney_triple_glossy", "disney_triple_glossy_uniform", "disney_triple_glossy_colour_map", "sss_probe", "sss_probe_uniform", "nd_sample", "sss_scatter", "skin", "skin_uniform", "skin_integrate",
             "ggx_reflect_refract_host", "ggx_reflect_refract_host_materials", "ggx_shade_host_materials",
             "disney_shade_host_materials", "skin_integrate_host_materials"]

# every lobe of rlDisney switched on: the parameters of disney_triple_glossy_uniform, one value each for the whole batch
DISNEY_UNIFORM = dict(base_color=(0.850000024, 0.704699695, 0.205699995), subsurface=0.2, metallic=0.3, specular=0.5, specular_tint=0.25,
                      roughness=0.4, anisotropic=0.4, sheen=0.5, sheen_tint=0.5, clearcoat=0.6, clearcoat_gloss=0.7)

# rlSkin's node defaults (src/rlSkin.cpp:109-128) with the sheen layer switched on and a skin-like scatter distance: the
# parameters of the *_uniform skin workload, one value each for the whole batch
SKIN_UNIFORM = dict(sss_color=(1.0, 0.842350006, 0.5), sss_weight=1.0, sss_dist_multiplier=1.0, sss_scatter_dist=(1.0, 0.6, 0.35),
                    specular_color=(1.0, 1.0, 1.0), specular_weight=0.6, specular_roughness=0.5, specular_ior=1.44,
                    sheen_color=(1.0, 1.0, 1.0), sheen_weight=0.3, sheen_roughness=0.35, sheen_ior=1.44)


class Workload:
    """name, samples per point, algorithmic bytes per point, a launch() closure"""

    def __init__(self, name, samples_per_point, bytes_per_point, launch, kernel, desc, bound="hbm", launches_per_step=1,
                 survey_bytes=None, config=None):
        self.name, self.samples_per_point, self.bytes_per_point = name, samples_per_point, bytes_per_point
        self.launch, self.kernel, self.desc = launch, kernel, desc
        # bytes_per_point: the planes the verb's arithmetic needs (= what the kernel moves); survey_bytes: SURVEY.md 8(d)'s
        # figure where that also counts planes the reference reads and never uses
        self.survey_bytes = survey_bytes
        self.config = config                       # BASELINE.json configuration this workload is the kernel of
        # "hbm": the pointwise streaming kernels; "valu": the n^2-spp integrators, which read ~100 B per point for
        # tens of triples of arithmetic ("VALU-bound, not HBM-bound ... must be stated as such")
        self.bound = bound
        self.launches_per_step = launches_per_step     # kernel launches one step() issues (chunked streaming)


# planes (n floats each) a workload reads and writes: sizes its arena
PLANES = {"ggx_reflect_refract": 19 + 12, "ggx_reflect_refract_host": 19, "ggx_reflect_refract_host_materials": 15, "ggx_shade_host_materials": 13, "disney_shade_host_materials": 13, "skin_integrate_host_materials": 13, "ggx_reflect_refract_uniform": 13 + 12, "ggx_reflect_refract_materials": 13 + 12, "ggx_reflect": 17 + 8,
          "ggx_eval": 17 + 8 + 3, "ggx_pdf": 17 + 8 + 1, "disney_triple_diffuse": 24 + 7, "disney_triple_glossy": 24 + 7, "disney_triple_glossy_uniform": 11 + 7, "disney_triple_glossy_colour_map": 14 + 7,
          "nd_sample": 9 + 7 + 5, "disney_integrate": 22 + 8, "disney_stream": 22 + 8,
          "sss_probe": 17 + 12, "sss_probe_uniform": 11 + 12,
          "sss_scatter": 15 + 3, "skin": 35 + 24, "skin_uniform": 15 + 24, "skin_integrate": 29 + 3 + 15, "ggx_direct": 15 + 3 + 6 + 6,
          "disney_direct": 22 + 3 + 6, "ggx_shade": 15 + 3 + 6 + 4 + 18, "disney_shade": 22 + 3 + 15}     # (the generator's wo planes included where the closure ignores them)


def _as_planes(params: dict, n: int, colours=()):
    """Experiment switch RLS_BENCH_UNIFORM_AS_PLANES for the *_uniform workloads: "1" hands every parameter over as a constant
    per-point plane (the streamed kernel on the same values), "colours" only the named colour / weight parameters."""
    mode = os.environ.get("RLS_BENCH_UNIFORM_AS_PLANES", "")
    if mode not in ("1", "colours"):
        return params
    import torch
    const = lambda v: torch.full((n,), float(v), dtype=torch.float32, device="cuda")
    return {k: ((torch.stack([const(x) for x in v]) if isinstance(v, tuple) else const(v)) if mode == "1" or k in colours else v)
            for k, v in params.items()}


def make_workload(R, ctx, name: str, n: int, first: int, candidates: int = 1, chunk_log2: int = 20, depth: int = 3):
    """All planes of the workload live in one arena (R.Arena): one allocation, and with candidates > 1 the
    fastest of that many equally sized blocks ("Placement")."""
    A = R.Arena(ctx, n, PLANES[name], candidates)
    u = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(ctx, SEED, first, n, stream, lo, hi, out=A.plane())

    def u3(stream, lo=0.0, hi=1.0):
        t = A.planes(3)
        for j in range(3):
            R.gen_uniform(ctx, SEED, first, n, stream + j, lo, hi, out=t[j])
        return t

    wo, N, T = R.gen_frame(ctx, SEED, first, n, out=(A.planes(3), A.planes(3), A.planes(3)))
    if name in ("ggx_reflect_refract", "ggx_reflect", "ggx_eval", "ggx_pdf"):
        g = R.GgxSampler(ctx, wo, N, T, specColor=u3(S_KS), ior=u(S_IOR, 1.05, 2.55),
                         roughness=u(S_ROUGH, 0.05, 1.0), anisotropic=R.gen_aniso(ctx, SEED, first, n, out=A.plane()))
        xi = [u(S_XI0 + j) for j in range(4 if name == "ggx_reflect_refract" else 2)]
        if name == "ggx_reflect_refract":
            out = (A.planes(3), A.planes(3), A.plane(), A.plane(), A.planes(3), A.plane())
            # in: wo3 N3 T3 Ks3 rough ior aniso xi4 = 19 f; out: wi3 f3 pdf F wt3 weight = 12 f
            wl = Workload(name, 2, (19 + 12) * 4, lambda: g.reflectRefract(xi[0], xi[1], xi[2], xi[3], out=out),
                          "ggx_kernel<5, {m}, 1>",
                          "rlGgx reflect+refract VNDF sampling, mixed params (SURVEY 8d config 2)", config=2)
            wl.outputs = out
        else:
            out = (A.planes(3), A.planes(3), A.plane(), A.plane())
            if name == "ggx_reflect":
                wl = Workload(name, 1, (17 + 8) * 4, lambda: g.sampleEvalPdf(xi[0], xi[1], out=out),
                              "ggx_kernel<3, {m}, 1>", "rlGgx reflect triple, mixed params")
            else:
                # the verbs alone, as Arnold's integrators call them (src/rlGgx.h:110-127), on the directions evalSample drew
                g.sampleEvalPdf(xi[0], xi[1], out=out)
                wi = out[0]
                if name == "ggx_eval":
                    f = A.planes(3)
                    # evalBrdf reads wo3 N3 T3 Ks3 rough ior aniso wi3 = 18 f, writes f3
                    wl = Workload(name, 1, (18 + 3) * 4, lambda: g.evalBrdf(wi, out=f), "ggx_kernel<1, {m}, 1>",
                                  "rlGgx evalBrdf alone on sampled directions, mixed params (src/rlGgx.h:110-119)")
                else:
                    pdf = A.plane()
                    # evalPdf needs no colour and no ior: wo3 N3 T3 rough aniso wi3 = 14 f, writes pdf
                    wl = Workload(name, 1, (14 + 1) * 4, lambda: g.evalPdf(wi, out=pdf), "ggx_kernel<2, {m}, 1>",
                                  "rlGgx evalPdf alone on sampled directions, mixed params (src/rlGgx.h:121-127)",
                                  survey_bytes=(18 + 1) * 4)
    elif name in ("ggx_reflect_refract_host", "ggx_reflect_refract_host_materials"):
        by_ref = name.endswith("materials")
        # config 2 with the batch in page-locked HOST memory, where an Arnold-side stub's render threads gather it (the
        # reference evaluates per hit on those threads, src/rlGgx.cpp:248-261): 19 planes up, the same kernel per chunk,
        # 12 planes down, overlapped on `depth` streams (rlshaders_amd.Pipeline = rls_pipeline_*).  PCIe-bound.
        import torch
        if by_ref:
            # the same batch as the hits of 256 node instances: the six parameters travel as per-MATERIAL columns, uploaded
            # once, and every point carries its material id (rls_material_index): 14 planes up instead of 19
            M = 256
            table = [R.gen_uniform(ctx, SEED, 0, M, S_KS + j) for j in range(3)] + \
                    [R.gen_uniform(ctx, SEED, 0, M, S_ROUGH, 0.05, 1.0), R.gen_uniform(ctx, SEED, 0, M, S_IOR, 1.05, 2.55),
                     R.gen_aniso(ctx, SEED, 0, M)]
            ids = (R.gen_uniform(ctx, SEED, first, n, S_PARAM0 + 30) * M).to(torch.int32).clamp_(0, M - 1)
            dev_in = [wo[0], wo[1], wo[2], N[0], N[1], N[2], T[0], T[1], T[2]] + [u(S_XI0 + j) for j in range(4)] + \
                     [ids.view(torch.float32)]
        else:
            dev_in = [wo[0], wo[1], wo[2], N[0], N[1], N[2], T[0], T[1], T[2]] + list(u3(S_KS)) + \
                     [u(S_ROUGH, 0.05, 1.0), u(S_IOR, 1.05, 2.55), R.gen_aniso(ctx, SEED, first, n, out=A.plane())] + \
                     [u(S_XI0 + j) for j in range(4)]
        nin = len(dev_in)
        # ONE page-locked [planes, n] array per direction, as a stub's batch buffers are: equally spaced planes travel as one
        # strided copy per chunk and direction (rls_pipeline_run)
        hin_all = torch.empty(nin, n, dtype=torch.float32, pin_memory=True)
        hout_all = torch.empty(12, n, dtype=torch.float32, pin_memory=True)
        hin, hout = [hin_all[k] for k in range(nin)], [hout_all[k] for k in range(12)]
        torch.cuda.synchronize()
        for h, d in zip(hin, dev_in):
            h.copy_(d)
        cp = min(n, 1 << chunk_log2)
        pipe = R.Pipeline(ctx, cp, nin, 12, depth)
        # Freed device memory is cleared by the driver in the background -- on the copy engines: for a few seconds after a
        # multi-GB arena has been released (the previous workload of this process) the two copy directions no longer run at
        # once (tools/diag_copy_rates.py: both-directions rate 96 -> 57-64 GB/s right after a free, back after an idle second
        # or two).  Wait for that to pass before measuring a PCIe-bound pipeline: poll until both directions overlap again.
        settle = []
        for _ in range(40):
            r = pipe.copy_rates(1 << 27)
            settle.append(round(r["both"], 1))
            if r["both"] >= 1.4 * max(r["h2d"], r["d2h"]):
                break
            time.sleep(0.25)

        def chunk(slot, _first, count, i, o):
            # raw device addresses of the chunk's planes straight into the C ABI (no tensor object per plane and chunk: on a
            # slow host core 31 of them per chunk cost more than the chunk's copies)
            c = R._capi.GgxClosure()
            c.wo = R._capi.CVec3(i.ptr(0), i.ptr(1), i.ptr(2))
            c.N = R._capi.CVec3(i.ptr(3), i.ptr(4), i.ptr(5))
            c.T = R._capi.CVec3(i.ptr(6), i.ptr(7), i.ptr(8))
            c.KsColor = R._capi.ParamRgb(i.ptr(9), i.ptr(10), i.ptr(11), 0.0, 0.0, 0.0)
            c.specularRoughness = R._capi.Param(i.ptr(12), 0.0)
            c.ior = R._capi.Param(i.ptr(13), 0.0)
            c.anisotropic = R._capi.Param(i.ptr(14), 0.0)
            R._capi.check(slot.lib.rls_ggx_reflect_refract(
                slot.handle, count, R.closures.C.byref(c), i.ptr(15), i.ptr(16), i.ptr(17), i.ptr(18),
                R._capi.Vec3(o.ptr(0), o.ptr(1), o.ptr(2)), R._capi.Rgb(o.ptr(3), o.ptr(4), o.ptr(5)), o.ptr(6), o.ptr(7),
                R._capi.Vec3(o.ptr(8), o.ptr(9), o.ptr(10)), o.ptr(11)))

        def chunk_by_reference(slot, _first, count, i, o):
            c = R._capi.GgxClosure()
            c.wo = R._capi.CVec3(i.ptr(0), i.ptr(1), i.ptr(2))
            c.N = R._capi.CVec3(i.ptr(3), i.ptr(4), i.ptr(5))
            c.T = R._capi.CVec3(i.ptr(6), i.ptr(7), i.ptr(8))
            c.KsColor = R._capi.ParamRgb(table[0].data_ptr(), table[1].data_ptr(), table[2].data_ptr(), 0.0, 0.0, 0.0)
            c.specularRoughness = R._capi.Param(table[3].data_ptr(), 0.0)
            c.ior = R._capi.Param(table[4].data_ptr(), 0.0)
            c.anisotropic = R._capi.Param(table[5].data_ptr(), 0.0)
            c.materials = R._capi.MaterialIndex(i.ptr(13), M)
            R._capi.check(slot.lib.rls_ggx_reflect_refract(
                slot.handle, count, R.closures.C.byref(c), i.ptr(9), i.ptr(10), i.ptr(11), i.ptr(12),
                R._capi.Vec3(o.ptr(0), o.ptr(1), o.ptr(2)), R._capi.Rgb(o.ptr(3), o.ptr(4), o.ptr(5)), o.ptr(6), o.ptr(7),
                R._capi.Vec3(o.ptr(8), o.ptr(9), o.ptr(10)), o.ptr(11)))

        fn = chunk_by_reference if by_ref else chunk
        wl = Workload(name, 2, (nin + 12) * 4, lambda: pipe.run(n, hin, hout, fn), "ggx_kernel<5, {m}, %d>" % (0 if by_ref else 1),
                      f"rlGgx reflect+refract, batch resident in page-locked HOST memory: chunks of {cp} points uploaded, "
                      f"sampled and downloaded on {depth} streams (rls_pipeline_*); PCIe-bound, " +
                      ("parameters by reference (256 node instances: a material id per point, the six parameters as per-material "
                       "columns uploaded once): 56 B up + 48 B down per point" if by_ref else "76 B up + 48 B down per point"),
                      bound="pcie", launches_per_step=(n + cp - 1) // cp)
        wl.pipe, wl.host, wl.settle, wl.up_planes = pipe, (hin, hout), settle, nin
        if by_ref:
            wl.table = (table, ids)
    elif name in ("ggx_shade_host_materials", "disney_shade_host_materials", "skin_integrate_host_materials"):
        # the whole shader_evaluate of a node on a batch in page-locked HOST memory: what crosses the bus per shading point is
        # its geometry (wo3 N3 T3 P3) and a material id up, sg->out.RGB down -- 52 B + 12 B for the node's 128-144 (rlSkin: 48)
        # samples, instead of 104 B for the 2 samples of config 2's verbs.  256 node instances' parameters go by reference
        # (rls_material_index); the AOVs stay on the device (an AOV nobody enabled is not downloaded).
        import torch
        node = name.split("_")[0]
        M = 256
        K, Cc = R._capi, R.closures.C
        gu = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(ctx, SEED, 0, M, stream, lo, hi)
        gu3 = lambda stream, lo=0.0, hi=1.0: [gu(stream + j, lo, hi) for j in range(3)]
        col = lambda t: K.Param(t.data_ptr(), 0.0)
        col3 = lambda t: K.ParamRgb(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), 0.0, 0.0, 0.0)
        ids = (R.gen_uniform(ctx, SEED, first, n, S_PARAM0 + 30) * M).to(torch.int32).clamp_(0, M - 1)
        P = N if node == "skin" else u3(S_PARAM0 + 8 if node == "ggx" else S_PARAM0 + 16, 0.0, 4.0)      # rlSkin: points on the unit sphere
        dev_in = [wo[0], wo[1], wo[2], N[0], N[1], N[2], T[0], T[1], T[2], P[0], P[1], P[2], ids.view(torch.float32)]
        nin = len(dev_in)
        lights = [R.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
                  R.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0))]
        la, nl = R.closures.light_array(lights)
        env = (Cc.c_float * 3)(1.0, 0.9, 0.8)
        geometry = lambda c, i: (setattr(c, "wo", K.CVec3(i.ptr(0), i.ptr(1), i.ptr(2))), setattr(c, "N", K.CVec3(i.ptr(3), i.ptr(4), i.ptr(5))),
                                 setattr(c, "T", K.CVec3(i.ptr(6), i.ptr(7), i.ptr(8))), setattr(c, "materials", K.MaterialIndex(i.ptr(12), M)))
        rgbs = lambda o, count: [K.Rgb(o.device_ptr(3 * j), o.device_ptr(3 * j + 1), o.device_ptr(3 * j + 2)) for j in range(count)]
        if node == "ggx":
            tab = dict(Ks=gu3(S_KS), rough=gu(S_ROUGH, 0.05, 1.0), ior=gu(S_IOR, 1.05, 2.55), aniso=R.gen_aniso(ctx, SEED, 0, M),
                       KdColor=gu3(S_PARAM0), Kd=gu(S_PARAM0 + 3), KdRough=gu(S_PARAM0 + 4), KsW=gu(S_PARAM0 + 5),
                       KtColor=gu3(S_PARAM0 + 11), Kt=gu(S_PARAM0 + 14))
            nout, spp, kernel, samples = 18, 4, "ggx_shade_kernel<1, {m}>", 144

            def chunk_node(slot, cfirst, count, i, o):
                c = K.GgxClosure()
                geometry(c, i)
                c.KsColor, c.specularRoughness, c.ior, c.anisotropic = col3(tab["Ks"]), col(tab["rough"]), col(tab["ior"]), col(tab["aniso"])
                sh = K.GgxShader(col3(tab["KdColor"]), col(tab["Kd"]), col(tab["KdRough"]), col(tab["KsW"]), col3(tab["KtColor"]), col(tab["Kt"]))
                out = K.GgxShadeOut(*rgbs(o, 6))
                K.check(slot.lib.rls_ggx_shade(slot.handle, count, Cc.byref(c), Cc.byref(sh), K.CVec3(i.ptr(9), i.ptr(10), i.ptr(11)), la, nl,
                                               env, 1, spp, SEED, first + cfirst, Cc.byref(out)))
        elif node == "disney":
            tab = dict(base=gu3(S_KS), **{k: gu(S_PARAM0 + j) for j, k in enumerate(K.DISNEY_SCALARS)})
            nout, spp, kernel, samples = 15, 4, "disney_shade_kernel<1, {m}>", 128

            def chunk_node(slot, cfirst, count, i, o):
                c = K.DisneyClosure()
                geometry(c, i)
                c.base_color = col3(tab["base"])
                for k in K.DISNEY_SCALARS:
                    setattr(c, k, col(tab[k]))
                out = K.DisneyShadeOut(*rgbs(o, 5))
                K.check(slot.lib.rls_disney_shade(slot.handle, count, Cc.byref(c), K.CVec3(i.ptr(9), i.ptr(10), i.ptr(11)), la, nl, env, spp,
                                                  SEED, first + cfirst, Cc.byref(out)))
        else:
            tab = dict(sss_color=gu3(S_PARAM0), sss_weight=gu(S_PARAM0 + 3), sss_dist_multiplier=gu(S_PARAM0 + 4, 0.5, 1.5),
                       sss_scatter_dist=gu3(S_PARAM0 + 5, 0.1, 2.1), specular_color=gu3(S_PARAM0 + 8), specular_weight=gu(S_PARAM0 + 11),
                       specular_roughness=gu(S_PARAM0 + 12, 0.05, 1.0), specular_ior=gu(S_PARAM0 + 13, 1.05, 2.55),
                       sheen_color=gu3(S_PARAM0 + 14), sheen_weight=gu(S_PARAM0 + 17), sheen_roughness=gu(S_PARAM0 + 18, 0.05, 1.0),
                       sheen_ior=gu(S_PARAM0 + 19, 1.05, 2.55))
            scene = R.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
            nout, spp, kernel, samples = 15, 4, "skin_integrate_kernel<1, {m}>", 48

            def chunk_node(slot, cfirst, count, i, o):
                c = K.SkinClosure()
                geometry(c, i)
                for k in ("sss_color", "specular_color", "sheen_color"):
                    setattr(c, k, col3(tab[k]))
                for k in ("sss_weight", "sss_dist_multiplier", "specular_weight", "specular_roughness", "specular_ior", "sheen_weight",
                          "sheen_roughness", "sheen_ior"):
                    setattr(c, k, col(tab[k]))
                for k in range(3):
                    c.sss_scatter_dist[k] = col(tab["sss_scatter_dist"][k])
                # device planes 0..8 the three layer AOVs, 9..11 their scalars, 12..14 sg->out.RGB (the one that is downloaded)
                a = rgbs(o, 3)
                out = K.SkinIntegrateOut(a[0], a[1], a[2], K.Rgb(o.device_ptr(12), o.device_ptr(13), o.device_ptr(14)),
                                         o.device_ptr(9), o.device_ptr(10), o.device_ptr(11))
                K.check(slot.lib.rls_skin_integrate(slot.handle, count, Cc.byref(c), K.CVec3(i.ptr(9), i.ptr(10), i.ptr(11)), Cc.byref(scene),
                                                    env, None, 0, spp, SEED, first + cfirst, Cc.byref(out)))
        hin_all = torch.empty(nin, n, dtype=torch.float32, pin_memory=True)
        hout_all = torch.empty(3, n, dtype=torch.float32, pin_memory=True)
        hin = [hin_all[k] for k in range(nin)]
        hout = [None] * (nout - 3) + [hout_all[k] for k in range(3)]          # the AOVs are not downloaded, sg->out.RGB is
        torch.cuda.synchronize()
        for h, d in zip(hin, dev_in):
            h.copy_(d)
        cp = min(n, 1 << chunk_log2)
        pipe = R.Pipeline(ctx, cp, nin, nout, depth)
        settle = []
        for _ in range(40):                                              # see ggx_reflect_refract_host
            r = pipe.copy_rates(1 << 27)
            settle.append(round(r["both"], 1))
            if r["both"] >= 1.4 * max(r["h2d"], r["d2h"]):
                break
            time.sleep(0.25)
        what = {"ggx": "rlGgx shader_evaluate, whole (two lights x 48 + 3 x 16 samples per point)",
                "disney": "rlDisney shader_evaluate, whole (two lights x 48 + 2 x 16 samples per point)",
                "skin": "rlSkin shader_evaluate (16 samples per layer, probe rays on an analytic sphere)"}[node]
        wl = Workload(name, samples, (nin + 3) * 4, lambda: pipe.run(n, hin, hout, chunk_node), kernel,
                      f"{what}, batch resident in page-locked HOST memory: chunks of {cp} points on {depth} streams, parameters by "
                      "reference (256 node instances); per shading point 52 B up (wo3 N3 T3 P3 + material id) and 12 B down "
                      "(sg->out.RGB; the AOVs stay on the device)",
                      bound="pcie", launches_per_step=(n + cp - 1) // cp)
        wl.pipe, wl.host, wl.settle, wl.up_planes, wl.down_planes = pipe, (hin, hout), settle, nin, 3
        wl.table = (tab, ids, P)
    elif name == "ggx_reflect_refract_materials":
        # config 2's batch as the hits of 256 node instances, device-resident: a material id per point, the six parameters as
        # per-instance columns (rls_material_index) -- the MIXED kernel with the parameters gathered from the table
        import torch
        M = 256
        table = dict(specColor=torch.stack([R.gen_uniform(ctx, SEED, 0, M, S_KS + j) for j in range(3)]),
                     roughness=R.gen_uniform(ctx, SEED, 0, M, S_ROUGH, 0.05, 1.0), ior=R.gen_uniform(ctx, SEED, 0, M, S_IOR, 1.05, 2.55),
                     anisotropic=R.gen_aniso(ctx, SEED, 0, M))
        ids = (R.gen_uniform(ctx, SEED, first, n, S_PARAM0 + 30) * M).to(torch.int32).clamp_(0, M - 1)
        g = R.GgxSampler(ctx, wo, N, T, materials=(ids, M), **table)
        xi = [u(S_XI0 + j) for j in range(4)]
        out = (A.planes(3), A.planes(3), A.plane(), A.plane(), A.planes(3), A.plane())
        wl = Workload(name, 2, (14 + 12) * 4, lambda: g.reflectRefract(xi[0], xi[1], xi[2], xi[3], out=out),
                      "ggx_kernel<5, {m}, 0>",
                      "rlGgx reflect+refract VNDF sampling, parameters by reference (256 node instances: a material id per point, "
                      "six per-instance columns): wo3 N3 T3 xi4 id in, 12 f out")
    elif name == "ggx_reflect_refract_uniform":
        # config 2's kernel as a stub without linked textures runs it: every node parameter one value for the batch
        # (Arnold parameters are constants unless textured), geometry and random numbers streamed
        params = _as_planes(dict(specColor=(0.9, 0.8, 0.7), ior=1.5, roughness=0.35, anisotropic=0.25), n, ("specColor",))
        g = R.GgxSampler(ctx, wo, N, T, **params)
        xi = [u(S_XI0 + j) for j in range(4)]
        out = (A.planes(3), A.planes(3), A.plane(), A.plane(), A.planes(3), A.plane())
        wl = Workload(name, 2, (13 + 12) * 4, lambda: g.reflectRefract(xi[0], xi[1], xi[2], xi[3], out=out),
                      "ggx_kernel<5, {m}, 2>",
                      "rlGgx reflect+refract VNDF sampling, uniform node parameters (KsColor, roughness 0.35, ior 1.5, "
                      "anisotropic 0.25): wo3 N3 T3 xi4 in, 12 f out")
    elif name == "ggx_direct":
        # the light loop of rlGgx (direct diffuse + direct specular): 16 light samples + 16 BSDF samples per lobe
        g = R.GgxSampler(ctx, wo, N, T, specColor=u3(S_KS), ior=u(S_IOR, 1.05, 2.55),
                         roughness=u(S_ROUGH, 0.05, 1.0), anisotropic=R.gen_aniso(ctx, SEED, first, n, out=A.plane()))
        P = u3(S_PARAM0 + 8, 0.0, 4.0)
        kdc, kd, kdr, ks = u3(S_PARAM0), u(S_PARAM0 + 3), u(S_PARAM0 + 4), u(S_PARAM0 + 5)
        light = R.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0))
        out = (A.planes(3), A.planes(3))
        wl = Workload(name, 48, (15 + 3 + 6 + 6) * 4,
                      lambda: g.directLighting(P, light, 4, SEED, KdColor=kdc, Kd=kd, diffuseRoughness=kdr, Ks=ks, out=out,
                                               first_index=first),
                      "ggx_direct_kernel<1, {m}>",
                      "rlGgx light loop: Oren-Nayar + GGX under a spherical light, 16 light + 2 x 16 BSDF samples per "
                      "point, power-heuristic MIS (SURVEY 8f rank 2; VALU-bound)", bound="valu")
    elif name == "ggx_shade":
        # shader_evaluate of rlGgx, whole: the light loop under two lights + transmission + indirect diffuse + indirect
        # glossy, 16 samples per loop
        g = R.GgxSampler(ctx, wo, N, T, specColor=u3(S_KS), ior=u(S_IOR, 1.05, 2.55),
                         roughness=u(S_ROUGH, 0.05, 1.0), anisotropic=R.gen_aniso(ctx, SEED, first, n, out=A.plane()))
        P = u3(S_PARAM0 + 8, 0.0, 4.0)
        kdc, kd, kdr, ks = u3(S_PARAM0), u(S_PARAM0 + 3), u(S_PARAM0 + 4), u(S_PARAM0 + 5)
        ktc, kt = u3(S_PARAM0 + 11), u(S_PARAM0 + 14)
        lights = [R.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
                  R.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0))]
        out = {k: A.planes(3) for k in R.GgxSampler.SHADE_AOVS + ("out",)}
        wl = Workload(name, 144, (15 + 3 + 6 + 4 + 18) * 4,
                      lambda: g.shade(P, lights, 4, SEED, KdColor=kdc, Kd=kd, diffuseRoughness=kdr, Ks=ks, KtColor=ktc, Kt=kt,
                                      env=(1.0, 0.9, 0.8), out=out, first_index=first),
                      "ggx_shade_kernel<1, {m}>",
                      "rlGgx shader_evaluate, whole: light loop under two spherical lights (2 x 48 samples) + integrateRefract + "
                      "indirect diffuse + integrateGlossy (3 x 16 samples) per point (src/rlGgx.cpp:248-327; VALU-bound)",
                      bound="valu")
    elif name == "disney_shade":
        base = u3(S_KS)
        sc = {k: u(S_PARAM0 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)}
        d = R.DisneySampler(ctx, wo, N, T, base_color=base, **sc)
        P = u3(S_PARAM0 + 16, 0.0, 4.0)
        lights = [R.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
                  R.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0))]
        out = {k: A.planes(3) for k in R.DisneySampler.SHADE_AOVS + ("out",)}
        wl = Workload(name, 128, (22 + 3 + 15) * 4,
                      lambda: d.shade(P, lights, 4, SEED, env=(1.0, 0.9, 0.8), out=out, first_index=first),
                      "disney_shade_kernel<1, {m}>",
                      "rlDisney shader_evaluate, whole: light loop under two spherical lights (2 x 48 samples) + integrateDiffuse + "
                      "integrateGlossy (2 x 16 samples) per point (src/rlDisney.cpp:685-727; VALU-bound)", bound="valu")
    elif name == "disney_direct":
        # the light loop of rlDisney (direct diffuse + direct specular) under two spherical lights: per light 16 light
        # samples (evaluated by both lobes) + 16 BSDF samples per lobe
        base = u3(S_KS)
        sc = {k: u(S_PARAM0 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)}
        d = R.DisneySampler(ctx, wo, N, T, base_color=base, **sc)
        P = u3(S_PARAM0 + 16, 0.0, 4.0)
        lights = [R.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
                  R.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0))]
        out = (A.planes(3), A.planes(3))
        wl = Workload(name, 96, (22 + 3 + 6) * 4,
                      lambda: d.directLighting(P, lights, 4, SEED, out=out, first_index=first),
                      "disney_direct_kernel<1, {m}>",
                      "rlDisney light loop: both lobes under two spherical lights, per light 16 light + 2 x 16 BSDF samples "
                      "per point, power-heuristic MIS (src/rlDisney.cpp:695-705; VALU-bound)", bound="valu")
    elif name in ("disney_triple_glossy_uniform", "disney_triple_glossy_colour_map"):
        params = _as_planes(dict(DISNEY_UNIFORM), n, ("base_color",))      # experiment switch, as in skin_uniform
        cmap = name.endswith("colour_map")
        if cmap:                # a colour map on an otherwise plain node: base_color per point, the ten scalars one value each
            params["base_color"] = u3(S_KS)
        d = R.DisneySampler(ctx, wo, N, T, **params)
        d.setSampleType(R.RLS_RAY_GLOSSY)
        xi = [u(S_XI0 + j) for j in range(2)]
        out = (A.planes(3), A.planes(3), A.plane())
        wl = Workload(name, 1, ((14 if cmap else 11) + 7) * 4, lambda: d.sampleEvalPdf(xi[0], xi[1], out=out),
                      "disney_kernel<3, false, {m}, %d>" % (3 if cmap else 2),
                      "rlDisney one-sample triple, glossy (GTR2 + clearcoat + sheen) lobe, " +
                      ("base_color textured, the ten scalars uniform (every lobe on): wo3 N3 T3 base3 xi2 in, wi3 f3 pdf out; "
                       "scalar-only arithmetic once per thread" if cmap else
                       "uniform node parameters (every lobe on): wo3 N3 T3 xi2 in, wi3 f3 pdf out; parameter-only arithmetic "
                       "once per thread"))
    elif name in ("disney_triple_diffuse", "disney_triple_glossy"):
        # the static triple of one lobe, one sample per point (src/rlDisney.cpp:109-152): evalSample -> evalBrdf -> evalPdf
        base = u3(S_KS)
        sc = {k: u(S_PARAM0 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)}
        d = R.DisneySampler(ctx, wo, N, T, base_color=base, **sc)
        d.setSampleType(R.RLS_RAY_DIFFUSE if name.endswith("diffuse") else R.RLS_RAY_GLOSSY)
        xi = [u(S_XI0 + j) for j in range(2)]
        out = (A.planes(3), A.planes(3), A.plane())
        lobe = 0 if name.endswith("diffuse") else 1
        # SURVEY's triple reads the whole closure: 22 f + xi2 in, wi3 f3 pdf out = 124 B.  The diffuse lobe's arithmetic needs
        # only subsurface, metallic and roughness of the ten scalars (src/rlDisney.cpp:199-236, 359-365, 515-518): 17 f in = 96 B;
        # the glossy lobe everything but subsurface: 23 f in = 120 B -- what the kernels move (profiles/r03_disney_triple_*_traffic)
        wl = Workload(name, 1, ((17 if lobe == 0 else 23) + 7) * 4, lambda: d.sampleEvalPdf(xi[0], xi[1], out=out),
                      "disney_kernel<3, %s, {m}, 1>" % ("true" if lobe == 0 else "false"),
                      f"rlDisney one-sample triple, {'diffuse' if lobe == 0 else 'glossy (GTR2 + clearcoat + sheen)'} lobe, "
                      "mixed params: wo3 N3 T3 base3 + the lobe's scalars + xi2 in, wi3 f3 pdf out (src/rlDisney.cpp:109-152)",
                      survey_bytes=(24 + 7) * 4)
    elif name in ("disney_integrate", "disney_stream"):
        base = u3(S_KS)
        sc = {k: u(S_PARAM0 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)}
        d = R.DisneySampler(ctx, wo, N, T, base_color=base, **sc)
        out = {"diffuse_sum": A.planes(3), "diffuse_count": A.plane(),
               "specular_sum": A.planes(3), "specular_count": A.plane()}
        if name == "disney_integrate":
            wl = Workload(name, 128, (22 + 8) * 4, lambda: d.integrate(8, SEED, out=out, first_index=first),
                          "disney_integrate_kernel<1, {m}>",
                          "rlDisney both lobes x 64 spp, reduced mode (SURVEY 8d config 3, mode R; VALU-bound)", bound="valu",
                          config=3)
            wl.outputs = out
        else:
            # mode S: every sample's (wi, f, pdf) = 28 B per triple goes to HBM; the 241 GB of a whole 2^26-point batch
            # are produced chunk by chunk into one chunk-sized set of sample-major planes (what a consumer would read
            # before the next chunk overwrites them: rls_disney_integrate_chunked)
            cp = min(n, 1 << chunk_log2)
            m = 2 * 64 * cp
            chunk = dict(wi=ctx.empty(3, m), f=ctx.empty(3, m), pdf=ctx.empty(m))
            wl = Workload(name, 128, (22 + 8) * 4 + 128 * 28,
                          lambda: d.integrateChunked(8, SEED, cp, out=out, chunk=chunk, first_index=first),
                          "disney_integrate_kernel<1, {m}>",
                          f"rlDisney both lobes x 64 spp, streamed mode in chunks of {cp} points (SURVEY 8d config 3, "
                          "mode S: 88 B in + 32 B sums + 128 x 28 B samples per point; VALU-bound: it runs at the speed of "
                          "mode R's arithmetic, 0.35 of the HBM peak)",
                          bound="valu", launches_per_step=(n + cp - 1) // cp, config=3)
    elif name in ("sss_probe", "sss_probe_uniform"):
        uniform = name.endswith("uniform")
        # _uniform: scatter distance and albedo one value for the batch, as a node without linked textures has them
        dist = (1.0, 0.6, 0.35)
        if uniform and os.environ.get("RLS_BENCH_UNIFORM_AS_PLANES") == "1":      # experiment switch, as in skin_uniform
            import torch
            dist = torch.stack([torch.full((n,), v, dtype=torch.float32, device="cuda") for v in dist])
        s = R.SssSampler(ctx, N, T, albedo=(0.8, 0.5, 0.4) if uniform else u3(S_KS),
                         dist=dist if uniform else u3(S_PARAM0, 0.1, 2.1))
        xi = [u(S_XI0 + j) for j in range(2)]
        out = {"r": A.plane(), "origin": A.planes(3), "dir": A.planes(3), "maxdist": A.plane(),
               "pdf": A.plane(), "profile": A.planes(3)}
        # SURVEY 8(d) config 4 counts 14 f in (dist3 albedo3 N3 T3 xi2) + 12 f out = 104 B; the reference computes `s` from the
        # albedo and never uses it (src/rlSss.cpp:22-23), so the verb needs -- and the kernel moves -- 11 f in: 92 B
        if uniform:
            wl = Workload(name, 1, (8 + 12) * 4, lambda: s.getProbeRay(xi[0], xi[1], out=out), "sss_kernel<3, 1, {m}>",
                          "rlSss ND probe ray + pdf + profile, uniform scatter distance (1, 0.6, 0.35): N3 T3 xi2 in, 12 f out; "
                          "setDistance once per thread")
        else:
            wl = Workload(name, 1, (11 + 12) * 4, lambda: s.getProbeRay(xi[0], xi[1], out=out),
                          "sss_kernel<3, 0, {m}>", "rlSss ND probe ray + pdf + profile (SURVEY 8d config 4)",
                          survey_bytes=(14 + 12) * 4, config=4)
            wl.outputs = out
    elif name == "nd_sample":
        # NDProfile alone: setDistance + getRadius + getPdf + evalProfile (src/rlSss.cpp:20-106); SURVEY 8(d) "profile-only":
        # 8 f in (dist3 albedo3 multiplier xi) + 5 f out = 52 B, of which the arithmetic needs dist3 xi: 4 f in
        p = R.NDProfile(ctx, n, u3(S_PARAM0, 0.1, 2.1), albedo=u3(S_KS))
        rx = u(S_XI0)
        out = (A.plane(), A.plane(), A.planes(3))
        wl = Workload(name, 1, (4 + 5) * 4, lambda: p.sample(rx, out=out), "sss_kernel<0, 0, {m}>",
                      "rlSss NDProfile alone: setDistance + getRadius + getPdf + evalProfile (SURVEY 8d config 4, profile-only)",
                      survey_bytes=(8 + 5) * 4)
    elif name == "sss_scatter":
        # shading points on the unit sphere (P = geometric normal), 16 probe rays each
        s = R.SssSampler(ctx, N, T, albedo=u3(S_KS), dist=u3(S_PARAM0, 0.02, 0.3))
        scene = R.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
        out = A.planes(3)
        wl = Workload(name, 16, (15 + 3) * 4, lambda: s.integrateScatter(N, scene, 4, SEED, out=out, first_index=first),
                      "sss_scatter_kernel<1, {m}>",
                      "rlSss integrateScatter, 16 probe rays per point on an analytic sphere (SURVEY 8f rank 3; "
                      "VALU-bound)", bound="valu")
    elif name == "skin_uniform":
        # experiment switch RLS_BENCH_UNIFORM_AS_PLANES: "1" the same values as per-point planes through the streamed kernel
        # (what the hoisting is worth), "colours" only the colours and layer weights (the colour-map case: the MIXED kernel)
        params = _as_planes(dict(SKIN_UNIFORM), n, ("sss_color", "specular_color", "sheen_color", "sss_weight", "specular_weight",
                                                    "sheen_weight"))
        sk = R.SkinShader(ctx, wo, N, T, **params)
        xi = A.planes(6)
        for j in range(6):
            R.gen_uniform(ctx, SEED, first, n, S_XI0 + j, out=xi[j])
        out = sk.alloc_out(arena=A)
        wl = Workload(name, 3, (15 + 24) * 4, lambda: sk.sampleEvalPdf(xi, out=out), "skin_kernel<{m}, 2>",
                      "rlSkin sheen GGX + specular GGX + SSS, uniform node parameters (node defaults, sheen 0.3, scatter distance "
                      "(1, 0.6, 0.35)): wo3 N3 T3 xi6 in, 24 f out; parameter-only arithmetic once per thread")
    elif name in ("skin", "skin_integrate"):
        p = dict(sss_color=u3(S_PARAM0), sss_weight=u(S_PARAM0 + 3), sss_dist_multiplier=u(S_PARAM0 + 4, 0.5, 1.5),
                 sss_scatter_dist=u3(S_PARAM0 + 5, 0.1, 2.1),
                 specular_color=u3(S_PARAM0 + 8), specular_weight=u(S_PARAM0 + 11),
                 specular_roughness=u(S_PARAM0 + 12, 0.05, 1.0), specular_ior=u(S_PARAM0 + 13, 1.05, 2.55),
                 sheen_color=u3(S_PARAM0 + 14), sheen_weight=u(S_PARAM0 + 17),
                 sheen_roughness=u(S_PARAM0 + 18, 0.05, 1.0), sheen_ior=u(S_PARAM0 + 19, 1.05, 2.55))
        sk = R.SkinShader(ctx, wo, N, T, **p)
        if name == "skin_integrate":
            # shader_evaluate with 16 samples per layer: 2 x 16 GGX triples + 16 probe rays per point, shading points on
            # the unit sphere (P = N); in 29 parameter planes + P3, out 4 AOVs x 3 + 3 layer scalars
            scene = R.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
            out = {k: A.planes(3) for k in ("sheen", "specular", "sss", "out")}
            out.update({k: A.plane() for k in ("sheenFresnel", "specularFresnel", "sssWeight")})
            wl = Workload(name, 48, (29 + 3 + 15) * 4,
                          lambda: sk.integrate(N, scene, 4, SEED, env=(1.0, 0.9, 0.8), out=out, first_index=first),
                          "skin_integrate_kernel<1, {m}>",
                          "rlSkin shader_evaluate, 16 samples per layer: sheen + specular integrateGlossy with the mean-"
                          "Fresnel hand-down, integrateScatter on an analytic sphere (src/rlSkin.cpp:174-254; VALU-bound)",
                          bound="valu")
            wl.arena = A
            return wl
        xi = A.planes(6)
        for j in range(6):
            R.gen_uniform(ctx, SEED, first, n, S_XI0 + j, out=xi[j])
        out = sk.alloc_out(arena=A)
        # 35 f in + 24 f out = 236 B; sss_color enters no arithmetic of the three samples (the albedo of
        # NDProfile::setDistance, unused: src/rlSss.cpp:22-23), so 32 f in: 224 B
        wl = Workload(name, 3, (32 + 24) * 4, lambda: sk.sampleEvalPdf(xi, out=out),
                      "skin_kernel<{m}, 1>", "rlSkin sheen GGX + specular GGX + SSS (SURVEY 8d config 5)",
                      survey_bytes=(35 + 24) * 4, config=5)
        wl.outputs = out
    else:
        raise ValueError(name)
    wl.arena = A
    return wl
