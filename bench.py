#!/usr/bin/env python3
"""bench.py -- BSDF Gsamples/s of the batched closure hot path on MI355X.

Default workload = BASELINE.json configs[1]: rlGgx reflect + refract visible-normal sampling over
2^26 SoA shading points per GPU, fp32, mixed per-point parameters.  One
"step" is one pass of the fused kernel (`rls_ggx_reflect_refract`) over the batch = 2 samples per
point (a reflect sample->eval->pdf triple and a refract sample->weight).  Inputs are generated on
the device before the timed region; nothing crosses PCIe inside it.  All planes of the workload live
in one arena, the fastest of `--arena-candidates` equally sized HBM blocks ("Placement");
the JSON line records the probe results under config.placement.

Contract: `python bench.py --gpus N --steps K --warmup W` ends with ONE compact JSON line on rank 0 (the headline: < 1800
bytes, the LAST line of stdout -- `headline()` below; a reader that keeps only a tail of stdout still gets all of it).
Everything longer -- the headline's full record (placement probe, device identities, counter sources) and one record per
workload of the `workloads` block -- is printed as its own JSON line BEFORE the headline, each tagged `"record": ...`, and
the same records are written to `gpurun_out/bench_workloads.json` (`--records-file`).  For N > 1
it runs under `python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU); the
batch shards by index range with no collective on the data path (weak scaling: every GPU gets its
own 2^26 points), the only communication being a barrier + device synchronise either side of the K timed steps and the
max over ranks of each rank's own elapsed time (taken after its synchronise, before the closing barrier).

`roofline`: algorithmic bytes per launch / average kernel duration (HIP events on the launch
stream) against the 8 TB/s HBM3E peak.  `cpu_baseline`: the CPU oracle (a port of the reference's
scalar closure code) timed on this host's cores on a bounded sample of the same workload.

`workloads` block (default run: `configs`): BASELINE configurations 3 (rlDisney 64 spp, reduced), 4 (the rlSss probe, 2^25
points per GPU) and 5 (rlSkin, 2^27 per GPU), each measured in this same process with >= 10 warm-up launches of its own, its
own `roofline` and -- on one GPU -- a short `cpu_baseline`; the default run finishes in about half a minute.  `--workloads all`
adds the verbs an Arnold-side stub calls one by one and the host-resident pipelines (minutes; the builder's profiling
sessions use it).  `--config {2,3,4,5}` makes one configuration the headline (workload + points per GPU of that BASELINE
configuration) and switches the block off.

Roofline accounting, per record: `frac` is on the bytes the verb's arithmetic needs (`algorithmic_bytes_per_point`;
where SURVEY.md 8(d) counts planes the reference reads and never uses -- the albedo of NDProfile::setDistance --
`survey_bytes_per_point` / `frac_survey_bytes` carry that figure too), `frac_counter_bytes` is on the HBM bytes the
PMC counters saw (profiles/*_traffic.json, separate rocprofv3 --pmc passes of the same command), and
`issue_slot_frac` prices the VALU wave-instructions the counters saw (profiles/*_flops.json) against one
wave-instruction per SIMD every two cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1.2288e12 per second.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = 157.3  # fp32 vector peak of the same guide (packed-fp32 FMA rate; the integrators' bound)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2   # wave64 VALU instructions per second: one per SIMD every two cycles
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

# BASELINE.json configs -> (workload, log2 of the points ONE GPU holds): config 4 is 2^28 points over 8 GPUs, config 5 2^30
CONFIG_PRESETS = {2: ("ggx_reflect_refract", 26), 3: ("disney_integrate", 26), 4: ("sss_probe", 25), 5: ("skin", 27)}

# the `workloads` block of the default line: (workload, log2 points per GPU, timed steps)
BLOCK_ALL = [("ggx_reflect_refract_uniform", 26, 40), ("ggx_reflect", 26, 40), ("ggx_eval", 26, 40), ("ggx_pdf", 26, 40),
             ("disney_triple_diffuse", 26, 40), ("disney_triple_glossy", 26, 40), ("disney_triple_glossy_uniform", 26, 40),
             ("disney_triple_glossy_colour_map", 26, 40),
             ("disney_integrate", 26, 12), ("disney_stream", 26, 10),
             ("sss_probe", 25, 40), ("sss_probe", 26, 40), ("sss_probe_uniform", 26, 40), ("nd_sample", 26, 40),
             ("skin", 26, 30), ("skin", 27, 20), ("skin_uniform", 26, 30),
             ("ggx_reflect_refract_host", 24, 4), ("ggx_reflect_refract_host_materials", 24, 4),
             ("ggx_shade_host_materials", 24, 4), ("disney_shade_host_materials", 24, 4), ("skin_integrate_host_materials", 24, 4)]
BLOCK_CONFIGS = [("disney_integrate", 26, 12), ("sss_probe", 25, 40), ("skin", 27, 20)]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the first ~10 launches of the VALU-bound EXACT kernel run up to 40 % slower while the GPU leaves its idle power state
    # (3.3, 2.9, 2.7, 2.6, 2.55, 2.5 ... 2.38 ms; tools/exp_warmup.py), hence ten warm-up steps by default
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2-points", type=int, default=None, help="points per GPU = 2^this (default 26, or the --config's)")
    ap.add_argument("--workload", default=None, choices=WORKLOADS, help="headline workload (default ggx_reflect_refract)")
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIG_PRESETS),
                    help="BASELINE.json configuration as the headline: 2 ggx_reflect_refract 2^26/GPU, 3 disney_integrate "
                         "2^26/GPU, 4 sss_probe 2^25/GPU (2^28 over 8), 5 skin 2^27/GPU (2^30 over 8)")
    ap.add_argument("--workloads", default="auto", choices=["auto", "all", "configs", "none"],
                    help="the `workloads` block: all = every other config and verb (minutes), configs = BASELINE configs 3-5 "
                         "only; auto = configs when the headline is the default, none otherwise")
    ap.add_argument("--records-file", default=None,
                    help="where the long records go besides stdout (default gpurun_out/bench_workloads.json; '-' = nowhere)")
    ap.add_argument("--chunk-log2", type=int, default=20,
                    help="disney_stream, ggx_reflect_refract_host: points per chunk = 2^this")
    ap.add_argument("--host-resident", action="store_true",
                    help="config 2 with the batch in page-locked HOST memory: = --workload ggx_reflect_refract_host "
                         "--log2-points 24 (a 2 GB batch), through rlshaders_amd.Pipeline (chunked upload / kernel / download "
                         "on --pipeline-depth streams); PCIe-bound")
    ap.add_argument("--pipeline-depth", type=int, default=3)
    ap.add_argument("--math", default="exact", choices=["exact", "fast"],
                    help="arithmetic of the measured kernels: exact (default, bit-faithful to the CPU closures) "
                         "or fast (RLS_MATH_FAST)")
    ap.add_argument("--arena-candidates", type=int, default=2,
                    help="equally sized blocks probed for the workload's plane arena (the fastest is kept); 1 = one arena, no "
                         "probing; 0 = no arena, every plane group its own allocation (for comparison).  Placement moves the "
                         "vector-issue-bound EXACT kernels by < 1 % (profiles/r02_placement.txt): two candidates are a check, "
                         "not a search; --math fast gains up to 15 % from 16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--checksum", action="store_true",
                    help="after the timed region: the order-independent 64-bit checksum (rls_checksum) of every output plane "
                         "of this rank's shard, gathered over the ranks; the shard sums of a G-rank run add up (mod 2^64) "
                         "to the sum of one process over the same G * n points (validation, BASELINE configs 2-5 only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target seconds of CPU-baseline work (headline)")
    ap.add_argument("--block-cpu-seconds", type=float, default=2.5, help="the same for each record of the workloads block")
    ap.add_argument("--block-log2-points", type=int, default=None,
                    help="every record of the workloads block at 2^this points per GPU instead of its configuration's size (tests)")
    a = ap.parse_args()
    if a.host_resident:
        a.workload = "ggx_reflect_refract_host"
        if a.log2_points is None:
            a.log2_points = 24
    explicit = a.workload is not None or a.config is not None or a.log2_points is not None
    if a.config is not None:
        w, l2 = CONFIG_PRESETS[a.config]
        if a.workload is not None and a.workload != w:
            ap.error(f"--config {a.config} is the workload {w}")
        a.workload = w
        if a.log2_points is None:
            a.log2_points = l2
    if a.workload is None:
        a.workload = "ggx_reflect_refract"
    if a.log2_points is None:
        a.log2_points = 26
    if a.workloads == "auto":
        a.workloads = "none" if explicit else "configs"
    return a


# ------------------------------------------------------------------------------------------------
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


# ------------------------------------------------------------------------------------------------
def _cpu_leg(workload: str, n: int, threads: int):
    """(run, samples_per_point, what): one pass of the CPU oracle over n points of the same seeded workload."""
    import numpy as np
    import oracle_lib as O
    import cases
    u = lambda stream, lo=0.0, hi=1.0: O.gen_uniform(SEED, 0, n, stream, lo, hi)
    u3 = lambda stream, lo=0.0, hi=1.0: np.stack([u(stream + j, lo, hi) for j in range(3)])
    if workload == "ggx_reflect_refract_uniform":
        wo, N, T = cases.frame(SEED, n)
        g = O.Ggx(wo, N, T, KsColor=(0.9, 0.8, 0.7), ior=1.5, roughness=0.35, anisotropic=0.25, nthreads=threads)
        x = cases.xi(SEED, n, 4)
        out = g.reflect_refract(x[0], x[1], x[2], x[3])
        return (lambda: g.reflect_refract(x[0], x[1], x[2], x[3], out=out)), 2, "orc_batch_ggx_reflect_refract"
    if workload in ("ggx_reflect_refract_host", "ggx_reflect_refract_host_materials", "ggx_reflect_refract_materials"):
        workload = "ggx_reflect_refract"            # the CPU closures' batch is host-resident by nature
    if workload.endswith("_host_materials") and not workload.startswith("ggx_reflect"):
        workload = workload[:-len("_host_materials")]            # ggx_shade, disney_shade, skin_integrate
    if workload in ("ggx_reflect_refract", "ggx_reflect", "ggx_eval", "ggx_pdf", "ggx_direct", "ggx_shade"):
        c = cases.ggx_mixed(SEED, n)
        g = O.Ggx(c["wo"], c["N"], c["T"], KsColor=c["KsColor"], ior=c["ior"], roughness=c["roughness"],
                  anisotropic=c["anisotropic"], nthreads=threads)
        x = cases.xi(SEED, n, 4)
        if workload in ("ggx_eval", "ggx_pdf"):
            wi = g.sample(x[0], x[1])[0]
            if workload == "ggx_eval":
                return (lambda: g.eval(wi)), 1, "orc_batch_ggx_eval"
            return (lambda: g.pdf(wi)), 1, "orc_batch_ggx_pdf"
        if workload == "ggx_reflect_refract":
            out = g.reflect_refract(x[0], x[1], x[2], x[3])
            return (lambda: g.reflect_refract(x[0], x[1], x[2], x[3], out=out)), 2, "orc_batch_ggx_reflect_refract"
        if workload == "ggx_reflect":
            return (lambda: g.sample_eval_pdf(x[0], x[1])), 1, "orc_batch_ggx_sample_eval_pdf"
        P = u3(S_PARAM0 + 8, 0.0, 4.0)
        lt = O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0))
        kdc, kd, kdr, ks = u3(S_PARAM0), u(S_PARAM0 + 3), u(S_PARAM0 + 4), u(S_PARAM0 + 5)
        if workload == "ggx_shade":
            ktc, kt = u3(S_PARAM0 + 11), u(S_PARAM0 + 14)
            lts = [lt, O.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0))]
            return (lambda: g.shade(P, lts, 4, SEED, Kd_color=kdc, Kd=kd, Kd_roughness=kdr, Ks=ks, Kt_color=ktc, Kt=kt,
                                    env=(1.0, 0.9, 0.8))), 144, "orc_batch_ggx_shade"
        return (lambda: g.direct_lighting(P, lt, 4, SEED, Kd_color=kdc, Kd=kd, Kd_roughness=kdr, Ks=ks)), 48, \
            "orc_batch_ggx_direct_lighting"
    if workload in ("disney_triple_glossy_uniform", "disney_triple_glossy_colour_map"):
        wo, N, T = cases.frame(SEED, n)
        params = dict(DISNEY_UNIFORM)
        if workload.endswith("colour_map"):
            params["base_color"] = u3(S_KS)
        d = O.Disney(wo, N, T, nthreads=threads, **params)
        x = cases.xi(SEED, n, 2)
        return (lambda: d.sample_eval_pdf(0x10, x[0], x[1])), 1, "orc_batch_disney_sample_eval_pdf"
    if workload in ("disney_integrate", "disney_stream", "disney_direct", "disney_shade", "disney_triple_diffuse",
                    "disney_triple_glossy"):
        c = cases.disney_mixed(SEED, n)
        sc = {k: c[k] for k in O.DISNEY_SCALARS}
        d = O.Disney(c["wo"], c["N"], c["T"], base_color=c["base_color"], nthreads=threads, **sc)
        if workload.startswith("disney_triple"):
            lobe = 0x08 if workload.endswith("diffuse") else 0x10        # AI_RAY_DIFFUSE / AI_RAY_GLOSSY
            x = cases.xi(SEED, n, 2)
            return (lambda: d.sample_eval_pdf(lobe, x[0], x[1])), 1, "orc_batch_disney_sample_eval_pdf"
        if workload == "disney_direct":
            P = u3(S_PARAM0 + 16, 0.0, 4.0)
            lts = [O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
                   O.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0))]
            return (lambda: d.direct_lighting(P, lts, 4, SEED)), 96, "orc_batch_disney_direct_lighting"
        if workload == "disney_shade":
            P = u3(S_PARAM0 + 16, 0.0, 4.0)
            lts = [O.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
                   O.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0))]
            return (lambda: d.shade(P, lts, 4, SEED, env=(1.0, 0.9, 0.8))), 128, "orc_batch_disney_shade"
        streamed = workload == "disney_stream"
        return (lambda: d.integrate(8, SEED, streamed=streamed)), 128, "orc_batch_disney_integrate"
    if workload == "nd_sample":
        s = O.Sss(n, u3(S_PARAM0, 0.1, 2.1), u3(S_KS), nthreads=threads)
        x = cases.xi(SEED, n, 1)
        return (lambda: s.nd_sample(x[0])), 1, "orc_batch_nd_sample_pdf_profile"
    if workload in ("sss_probe", "sss_probe_uniform", "sss_scatter"):
        _, N, T = cases.frame(SEED, n)
        if workload == "sss_probe_uniform":
            s = O.Sss(n, (1.0, 0.6, 0.35), (0.8, 0.5, 0.4), N=N, T=T, nthreads=threads)
            x = cases.xi(SEED, n, 2)
            return (lambda: s.probe(x[0], x[1])), 1, "orc_batch_sss_probe"
        if workload == "sss_probe":
            s = O.Sss(n, u3(S_PARAM0, 0.1, 2.1), u3(S_KS), N=N, T=T, nthreads=threads)
            x = cases.xi(SEED, n, 2)
            return (lambda: s.probe(x[0], x[1])), 1, "orc_batch_sss_probe"
        s = O.Sss(n, u3(S_PARAM0, 0.02, 0.3), u3(S_KS), N=N, T=T, nthreads=threads)
        scene = O.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
        return (lambda: O.integrate_scatter(s, N, scene, 4, SEED)), 16, "orc_batch_sss_integrate_scatter"
    if workload == "skin_uniform":
        wo, N, T = cases.frame(SEED, n)
        x = cases.xi(SEED, n, 6)
        return (lambda: O.skin(wo, N, T, SKIN_UNIFORM, x, nthreads=threads)), 3, "orc_batch_skin"
    if workload in ("skin", "skin_integrate"):
        wo, N, T = cases.frame(SEED, n)
        p = dict(sss_color=u3(S_PARAM0), sss_weight=u(S_PARAM0 + 3), sss_dist_multiplier=u(S_PARAM0 + 4, 0.5, 1.5),
                 sss_scatter_dist=u3(S_PARAM0 + 5, 0.1, 2.1),
                 specular_color=u3(S_PARAM0 + 8), specular_weight=u(S_PARAM0 + 11),
                 specular_roughness=u(S_PARAM0 + 12, 0.05, 1.0), specular_ior=u(S_PARAM0 + 13, 1.05, 2.55),
                 sheen_color=u3(S_PARAM0 + 14), sheen_weight=u(S_PARAM0 + 17),
                 sheen_roughness=u(S_PARAM0 + 18, 0.05, 1.0), sheen_ior=u(S_PARAM0 + 19, 1.05, 2.55))
        if workload == "skin_integrate":
            scene = O.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
            return (lambda: O.skin_integrate(wo, N, T, p, N, scene, 4, SEED, env=(1.0, 0.9, 0.8), nthreads=threads)), 48, \
                "orc_batch_skin_integrate"
        x = cases.xi(SEED, n, 6)
        return (lambda: O.skin(wo, N, T, p, x, nthreads=threads)), 3, "orc_batch_skin"
    raise ValueError(workload)


def cpu_baseline(workload: str, target_seconds: float) -> dict | None:
    """The CPU oracle on a bounded sample of the same synthetic workload, all host cores."""
    import oracle_lib as O
    threads = O.hardware_threads()            # CPUs this process may run on (sched_getaffinity)
    try:                                      # ... capped by a cgroup CPU quota, if the box sets one
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass

    def run(n, seconds):
        fn, spp, what = _cpu_leg(workload, n, threads)
        fn()                                                      # touch pages
        times = []
        t_end = time.perf_counter() + seconds
        while len(times) < 3 or time.perf_counter() < t_end:
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return times, spp, what

    n0 = 1 << 12
    t = min(run(n0, 0.0)[0])
    # one pass of about a second -- a third of the target if that is shorter -- (bounded by 2^24 points), repeated for
    # target_seconds
    n = int(min(1 << 24, max(n0, n0 * min(1.0, target_seconds / 3.0) / max(t, 1e-6))))
    n = 1 << (n.bit_length() - 1)
    times, spp, what = run(n, target_seconds)
    best, mean = min(times), sum(times) / len(times)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(spp * n / best / 1e9, 6), "unit": "Gsamples/s", "cores": threads, "kind": "port",
            "per_core_msamples": round(spp * n / best / 1e6 / threads, 3), "cpu_model": model,
            "mean_value": round(spp * n / mean / 1e9, 6),
            "note": "port of the reference's scalar closure code without its per-point heap allocation "
                    "(src/rlGgx.h:152 make_shared): a conservative (fast) CPU baseline",
            "sample_short": f"{n} points x {spp} samples, best of {len(times)} passes ({sum(times):.0f} s CPU), {what}, {threads} thr; "
                            "port = no per-point heap allocation (conservative)",
            "sample": f"{n} points ({spp} samples each) of the same seeded workload, best of {len(times)} passes "
                      f"({sum(times):.1f} s of CPU work), oracle/rls_oracle.c {what} on {threads} threads"}


def profile_record(workload: str, math: str, suffix: str, key: str):
    """The newest committed PMC record for this workload and arithmetic mode (profiles/*_<suffix>.json), or None.
    Counters come from separate `rocprofv3 --pmc` passes of this very command (tools/profile_round.sh); they cannot
    be collected inside an un-profiled run, so the line says which file each one comes from."""
    best = None
    for p in sorted((ROOT / "profiles").glob(f"*_{suffix}.json")):
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        if d.get("workload") == workload and d.get("math", "exact") == math and d.get(key):
            d["file"] = f"profiles/{p.name}"
            best = d
    return best


# ------------------------------------------------------------------------------------------------
def device_identity(torch, index: int) -> dict:
    """what tells two GPUs apart: uuid and PCI bus id of the HIP device a rank runs on"""
    p = torch.cuda.get_device_properties(index)
    ident = {"index": index, "name": p.name}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(p, k, None)
        if v is not None:
            ident[k] = str(v)
    return ident


def roofline_record(wl, n: int, kernel_ms: float, math: str) -> dict:
    """The `roofline` object of one measured workload (module docstring: "Roofline accounting")."""
    launches = wl.launches_per_step
    bytes_per_launch = n * wl.bytes_per_point / launches              # algorithmic bytes of one kernel launch
    launch_ms = kernel_ms / launches                                   # average duration of one kernel launch
    sec = launch_ms * 1e-3
    achieved_gbs = bytes_per_launch / sec / 1e9
    tr = profile_record(wl.name, math, "traffic", "hbm_bytes_per_launch")
    fl = profile_record(wl.name, math, "flops", "flops_per_point")
    # the PMC passes ran at their own batch size: scale the counters to this launch by the points it covers
    traffic = None
    if tr:
        prof_points = tr.get("points_per_launch") or (1 << 26) / launches
        traffic = tr["hbm_bytes_per_launch"] / prof_points * (n / launches)
    valu = fl["instructions_per_point"].get("SQ_INSTS_VALU") if fl else None
    salu = fl["instructions_per_point"].get("SQ_INSTS_SALU") if fl else None
    issue = valu * (n / launches / 64.0) / sec / VALU_ISSUE_PEAK if valu else None
    tflops = fl["flops_per_point"] * (n / launches) / sec / 1e12 if fl else None
    if wl.bound == "pcie":
        # host-resident batch: the copies bound the pass.  Held against the pinned-memory rates THIS box reaches (measured
        # now, rls_measure_copy_rates): each direction alone, and both at once
        rates = wl.pipe.copy_rates(1 << 28)
        up, down = n * wl.up_planes * 4.0, n * getattr(wl, "down_planes", 12) * 4.0
        t = kernel_ms * 1e-3
        floor = max(up / (rates["h2d"] * 1e9), down / (rates["d2h"] * 1e9))
        return {"bound": "pcie", "achieved": round((up + down) / t / 1e9, 2), "peak": round(rates["both"], 2), "unit": "GB/s",
                "frac": round(floor / t, 4), "traffic": None,
                "frac_note": "time the box's own pinned copies need for this pass (the slower of: all uploads at the measured "
                             "host -> device rate, all downloads at the device -> host rate; the two directions run on "
                             "separate DMA engines) / measured time of the pass; `peak` is the measured rate of both "
                             "directions at once",
                "h2d_gb_per_s": round(up / t / 1e9, 2), "d2h_gb_per_s": round(down / t / 1e9, 2),
                "box_copy_rates_gb_per_s": {k: round(v, 2) for k, v in rates.items()},
                "both_directions_rate_while_settling": getattr(wl, "settle", None),
                "pass_ms": round(kernel_ms, 4), "chunks_per_pass": launches, "kernel": wl.kernel.format(m=1 if math == "fast" else 0),
                "kernel_ms": round(launch_ms, 5), "launches_per_step": launches,
                "algorithmic_bytes_per_point": wl.bytes_per_point, "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "device_resident_note": "the same kernel on a device-resident batch is the workload " +
                                        (wl.name[:-len("_host_materials")] if wl.name.endswith("_host_materials") and
                                         not wl.name.startswith("ggx_reflect") else "ggx_reflect_refract")}
    if wl.bound == "hbm":
        roof = {"bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": int(traffic) if traffic else None}
    else:
        # the integrators run tens of triples per 100-odd bytes: bounded by fp32 vector issue.  Executed flops per
        # point come from the PMC instruction mix of this kernel (add + mul + 2 fma + transcendental, fp64 counted
        # once each), see tools/summarize_workload.py
        roof = {"bound": "valu", "achieved": round(tflops, 3) if fl else None, "peak": VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tflops / VALU_PEAK_TFLOPS, 4) if fl else None,
                "traffic": int(traffic) if traffic else None,
                "flops_per_point": fl["flops_per_point"] if fl else None,
                "hbm_gbs": round(achieved_gbs, 2), "hbm_frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "frac_note": "`frac` prices executed flops against the all-FMA packed peak; the kernels' own limit is "
                             "instruction issue: see issue_slot_frac"}
    roof.update({
        "frac_counter_bytes": round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
        "issue_slot_frac": round(issue, 4) if issue else None,
        "valu_per_point": valu, "salu_per_point": salu,
        "issue_slot_peak": "256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1.2288e12 wave64 VALU instructions per second",
        "counter_source": {"traffic": tr["file"] if tr else None, "instruction_mix": fl["file"] if fl else None,
                           "note": "rocprofv3 --pmc passes of this command (tools/profile_workload.sh), not this run"},
        "kernel": wl.kernel.format(m=1 if math == "fast" else 0),      # as rocprofv3 --kernel-trace names it
        "kernel_ms": round(launch_ms, 5), "launches_per_step": launches,
        "algorithmic_bytes_per_point": wl.bytes_per_point,
        "algorithmic_bytes_per_launch": int(bytes_per_launch)})
    if wl.survey_bytes:
        roof["survey_bytes_per_point"] = wl.survey_bytes
        roof["frac_survey_bytes"] = round(n * wl.survey_bytes / launches / sec / 1e9 / HBM_PEAK_GBS, 4)
    return roof


def measure(R, ctx, ranks, torch, name: str, log2n: int, steps: int, warmup: int, math: str, candidates: int,
            chunk_log2: int, other_mode: bool, depth: int = 3):
    """Build one workload on this rank's shard, warm it up, time `steps` passes between barriers.
    -> (workload, n, elapsed seconds [max over ranks], kernel ms per step [max over ranks], this rank's kernel ms,
        kernel ms per step in the other arithmetic mode or None)"""
    from rlshaders_amd.sharding import shard_range
    world, rank = ranks.world, ranks.rank
    n = 1 << log2n
    # weak scaling: the job is world * n points, rank g owns the index range [g*n, (g+1)*n)
    first, count = shard_range(world * n, rank, world)
    assert count == n
    wl = make_workload(R, ctx, name, n, first=first, candidates=candidates, chunk_log2=chunk_log2, depth=depth)
    torch.cuda.synchronize()
    for _ in range(warmup):
        wl.launch()
    torch.cuda.synchronize()
    ranks.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(steps):
        wl.launch()
    ctx.timer_stop()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0     # THIS rank's K steps (its device is idle again); the slowest rank's is the job's
    ranks.barrier()
    my_kernel_ms = ctx.timer_elapsed_ms() / max(steps, 1)
    if wl.bound == "pcie":                 # the pipeline runs on its own streams and returns when the batch is back in host
        my_kernel_ms = elapsed / max(steps, 1) * 1e3      # memory: the pass is timed on the host clock
    elapsed, kernel_ms = ranks.max_over_ranks([elapsed, my_kernel_ms])
    other_ms = None
    if other_mode and wl.bound != "pcie":
        # the other arithmetic mode, a few launches, for the record (not the headline number)
        ctx.set_math_mode(math != "fast")
        wl.launch()
        torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(5):
            wl.launch()
        ctx.timer_stop()
        other_ms = ranks.max_over_ranks([ctx.timer_elapsed_ms() / 5])[0]
        ctx.set_math_mode(math == "fast")
    return wl, n, elapsed, kernel_ms, my_kernel_ms, other_ms


# what tests/test_gpu_fast_mode.py holds RLS_MATH_FAST to (the measurement: profiles/r04_fast_conditioning.json)
FAST_PARITY = "opt-in, uncredited; protocol (3): 98.8-99.9 % of outliers where the oracle moves in its 1-ulp box, 4e-5 of points > 8x"

HEADLINE_MAX_BYTES = 1800          # the driver keeps a 2000-character tail of stdout: the last line must fit inside it


def _short(text: str, limit: int) -> str:
    return text if len(text) <= limit else text[:limit - 3] + "..."


def headline(detail: dict) -> dict:
    """The compact last line: the contract's keys, `roofline`, `cpu_baseline`, a `ranks` summary and the other arithmetic
    mode -- nothing whose size grows with the rank count or with the number of workloads."""
    pick = lambda d, keys: {k: d[k] for k in keys if k in d}
    line = pick(detail, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data"))
    cfg = detail["config"]
    line["config"] = pick(cfg, ("workload", "name", "baseline_config", "math", "points_per_gpu", "points_total",
                                "samples_per_point", "sharding"))
    line["config"]["workload"] = _short(cfg["workload"], 120)
    line["roofline"] = pick(detail["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                                                 "algorithmic_bytes_per_launch", "issue_slot_frac"))
    if "cpu_baseline" in detail:
        cb = detail["cpu_baseline"]
        line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["cpu_model"] = _short(cb.get("cpu_model", ""), 48)
        line["cpu_baseline"]["sample"] = _short(cb.get("sample_short", cb.get("sample", "")), 160)
    line["ranks"] = pick(detail["ranks"], ("ranks_seen", "world_size", "backend", "distinct_devices", "min_kernel_ms",
                                           "max_kernel_ms"))
    if "validation" in detail:
        line["validation"] = pick(detail["validation"], ("checksum",))
    if "other_math_mode" in detail:
        line["other_math_mode"] = pick(detail["other_math_mode"], ("math", "value", "hbm_frac"))
        line["other_math_mode"]["parity"] = _short(detail["other_math_mode"].get("parity", ""), 120)
    text = json.dumps(line)
    for drop in (("other_math_mode", "parity"), ("cpu_baseline", "sample"), ("config", "workload")):
        if len(text) < HEADLINE_MAX_BYTES:
            break
        if drop[0] in line and drop[1] in line[drop[0]]:
            line[drop[0]][drop[1]] = _short(str(line[drop[0]][drop[1]]), 24)      # never reached with today's strings: a guard
        text = json.dumps(line)
    assert len(text) < HEADLINE_MAX_BYTES, len(text)
    return line


def emit(detail: dict, records: list, records_file) -> None:
    """stdout: one JSON line per long record (`record`: headline_detail / workload), then the compact headline LAST."""
    long_records = [dict(detail, record="headline_detail")] + list(records)
    if records_file != "-":
        path = Path(records_file) if records_file else ROOT / "gpurun_out" / "bench_workloads.json"
        try:
            path.parent.mkdir(parents=True, exist_ok=True)
            path.write_text(json.dumps(long_records, indent=1) + "\n")
        except OSError as e:                          # a read-only checkout must not cost the run its line
            print(f"bench.py: records file not written: {e}", file=sys.stderr)
    for r in long_records:
        print(json.dumps(r), flush=True)
    sys.stderr.flush()
    print(json.dumps(headline(detail)), flush=True)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # launched bare: start one rank per GPU as child processes and return their exit code
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
               *sys.argv[1:]]
        sys.exit(subprocess.call(cmd))

    import torch
    import rlshaders_amd as R
    from rlshaders_amd.sharding import Ranks

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the closures only run on the HIP path")
    # one rank per GPU.  RLS_DIST_BACKEND=gloo (tests only) lets several ranks share one GPU so that the
    # multi-rank control path can be exercised on a single-GPU box, where RCCL refuses duplicate devices.
    backend = os.environ.get("RLS_DIST_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ        # under torchrun, also with one rank
    ranks = Ranks(backend=backend if (world > 1 or launched) else None,
                  device=torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu"),
                  launched=launched)
    rank = ranks.rank

    ctx = R.Context(device_index)
    ctx.set_math_mode(args.math == "fast")

    # who is here: every rank's device, gathered before anything is timed.  Under RCCL two ranks on one GPU would
    # halve each other's throughput silently -- refuse to produce a number then.
    identities = ranks.gather_objects(dict(device_identity(torch, device_index), rank=rank,
                                           host=os.uname().nodename, pid=os.getpid()))
    keys = [(d["host"], d.get("uuid") or d.get("pci_bus_id") or d["index"]) for d in identities]
    if backend == "nccl" and len(set(keys)) != len(keys):
        raise SystemExit(f"bench.py: two ranks share a GPU: {identities}")

    wl, n, elapsed, kernel_ms, my_ms, other_ms = measure(R, ctx, ranks, torch, args.workload, args.log2_points, args.steps,
                                                         args.warmup, args.math, args.arena_candidates, args.chunk_log2,
                                                         not args.checksum,       # (the other mode would overwrite the outputs)
                                                         args.pipeline_depth)
    per_rank_ms = ranks.gather_objects(round(my_ms, 5))
    shard_sums = None
    if args.checksum:
        outs = getattr(wl, "outputs", None)
        if outs is None:
            raise SystemExit(f"bench.py: --checksum covers the BASELINE configurations, not {wl.name}")
        planes = list(outs.values()) if isinstance(outs, dict) else list(outs)
        mine = sum(R.checksum(ctx, t) for t in planes) & 0xFFFFFFFFFFFFFFFF
        from rlshaders_amd.sharding import shard_range
        shard_sums = (ranks.gather_u64(mine), ranks.gather_objects(list(shard_range(world * n, rank, world))))

    detail = None
    if rank == 0:
        samples = world * n * wl.samples_per_point * args.steps
        value = samples / elapsed / 1e9
        roof = roofline_record(wl, n, kernel_ms, args.math)
        bytes_per_step = n * wl.bytes_per_point
        detail = {
            "metric": "BSDF Gsamples/sec (eval+sample+pdf)",
            "value": round(value, 4),
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl.desc, "name": wl.name, "baseline_config": wl.config, "math": args.math,
                       "points_per_gpu": n, "points_total": world * n,
                       "samples_per_point": wl.samples_per_point, "sharding": f"index-range x{world}, no collective",
                       "control_plane": (f"torch.distributed {backend}" if ranks.dist is not None else "single process"),
                       "placement": wl.arena.info(),
                       # the EXACT kernels are vector-ALU-bound: the first ~10 launches after idle run up to 40 % slower
                       # while the clocks ramp ("Warm-up"); fewer warm-up steps under-report
                       "warmup_note": None if args.warmup >= 10 else
                       f"warmup {args.warmup} < 10: the clock ramp of the first launches is inside the timed region "
                       "(under-reports by 1-4 % at 5, ~15 % at 1)"},
            "roofline": roof,
            # the ranks that took part, as gathered over the process group: a straggler or a doubled-up device shows here
            "ranks": {"ranks_seen": len(identities), "world_size": world, "backend": backend if ranks.dist is not None else None,
                      "devices": identities, "distinct_devices": len(set(keys)),
                      "per_rank_kernel_ms": per_rank_ms, "min_kernel_ms": min(per_rank_ms), "max_kernel_ms": max(per_rank_ms)},
        }
        if shard_sums is not None:
            detail["validation"] = {"shards": shard_sums[1], "shard_checksums": [f"{v:016x}" for v in shard_sums[0]],
                                    "checksum": f"{sum(shard_sums[0]) & 0xFFFFFFFFFFFFFFFF:016x}",
                                    "what": "sum of rls_checksum over every output plane of the last timed pass"}
        other = "exact" if args.math == "fast" else "fast"
        if other_ms is not None:
            detail["other_math_mode"] = {"math": other, "kernel_ms": round(other_ms, 5),
                                         "value": round(world * n * wl.samples_per_point / (other_ms * 1e-3) / 1e9, 4),
                                         "hbm_frac": round(bytes_per_step / (other_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "parity": FAST_PARITY if other == "fast" else "bit-exact against the CPU oracle"}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload, args.cpu_seconds)
            if cb:
                detail["cpu_baseline"] = cb
    del wl
    torch.cuda.empty_cache()

    # ---- the other configurations and verbs, same process, each with its own warm-up --------------------------------
    block = {"all": BLOCK_ALL, "configs": BLOCK_CONFIGS, "none": []}[args.workloads]
    if args.block_log2_points is not None:
        block = [(w, args.block_log2_points, min(st, 5)) for w, _, st in block]
        block = sorted(set(block), key=block.index)
    block = [b for b in block if not (b[0] == args.workload and b[1] == args.log2_points)]
    records = []
    for name, log2n, steps in block:
        warm = max(10, args.warmup)
        warm = warm if "_host" not in name else 2
        try:
            w, bn, el, kms, my, _ = measure(R, ctx, ranks, torch, name, log2n, steps, warm, args.math, 1, args.chunk_log2, False,
                                            args.pipeline_depth)
        except Exception as e:                      # noqa: BLE001
            # an extra record must not cost the headline its line (one process: no other rank is waiting at a barrier)
            if world > 1:
                raise
            records.append({"record": "workload", "name": name, "points_per_gpu": 1 << log2n, "error": f"{type(e).__name__}: {e}"})
            torch.cuda.empty_cache()
            continue
        prm = ranks.gather_objects(round(my, 5))
        if rank == 0:
            rec = {"record": "workload", "name": w.name, "baseline_config": w.config, "workload": w.desc, "points_per_gpu": bn,
                   "points_total": world * bn, "samples_per_point": w.samples_per_point,
                   "value": round(world * bn * w.samples_per_point * steps / el / 1e9, 4), "unit": "Gsamples/s",
                   "steps": steps, "warmup": warm, "ms_per_step": round(el / steps * 1e3, 5),
                   "roofline": roofline_record(w, bn, kms, args.math), "per_rank_kernel_ms": prm}
            if world == 1 and not args.no_cpu_baseline:
                rec["cpu_baseline"] = cpu_baseline(name, args.block_cpu_seconds)
            records.append(rec)
        del w
        torch.cuda.empty_cache()

    ranks.close()
    if rank == 0:
        emit(detail, records, args.records_file)


if __name__ == "__main__":
    main()
