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
`effective_clock_ghz` is the shader clock the kernel actually held (in-kernel s_memtime / s_memrealtime stamps of the kernel's
diagnostic instantiation, launched right behind the timed launches: rls_diag_clock_stamps_*), `issue_slot_frac_at_clock`
the same pricing at that clock, and `valu_busy_frac` the share of the launch's SIMD cycles in which the vector ALU holds an
instruction at all (profiles/*_stalls.json: SQ counters by exact kernel name) -- the direct measure of how close to its
issue ceiling a kernel runs, since only the cheapest instruction class costs two cycles.
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
# the workload table; `Workload`, `PLANES` and `make_workload` are re-exported (tools/diag_*.py import them from here)
from bench_workloads import (DISNEY_UNIFORM, PLANES, SEED, SKIN_UNIFORM, S_KS, S_PARAM0, WORKLOADS, Workload,  # noqa: F401
                             make_workload)

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
    ap.add_argument("--no-other-mode", action="store_true",
                    help="skip the few launches in the other arithmetic mode after the timed region (PMC passes: every dispatch "
                         "of the profiled command is then the measured kernel)")
    ap.add_argument("--no-clock", action="store_true",
                    help="skip the in-kernel clock measurement (roofline.effective_clock_ghz): after the timed region the "
                         "BASELINE kernels run once more as their stamped diagnostic instantiation, same launch pattern")
    ap.add_argument("--sustain-seconds", type=float, default=0.0,
                    help="keep launching the kernel for this long before the warm-up (profiling sessions: the clock a kernel "
                         "holds after seconds of load, MI355X_MICROARCH.md 'DVFS give-back' item 6)")
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


# the closures whose reference constructor heap-allocates (std::make_shared<NDSampler>, src/rlGgx.h:152): every GgxSamplerT --
# one per point in rlGgx, two in rlSkin (sheen + specular, src/rlSkin.cpp:191-228); rlDisney and NDProfile allocate nothing
_ALLOCATING = ("ggx", "skin")
CPU_FULL_LOG2 = 24          # >= 2^24 points, best of 3


def cpu_baseline(workload: str, target_seconds: float, full: bool = True) -> dict | None:
    """The CPU oracle on the same seeded workload, all host cores.  Two legs on the same inputs:
    `port` (the restatement as it is: no heap allocation per closure) and `port+alloc` (orc_set_closure_alloc: every
    GgxSampler constructor pays the allocation + release the reference's make_shared costs) -- the reference's real closure
    path lies between the two.  `full`: 2^24 points, best of 3 passes per leg, whenever three such passes fit into
    `target_seconds`; otherwise (and for the records of the workloads block) a bounded sample: passes of about a third of
    `target_seconds`, at least 3."""
    import oracle_lib as O
    threads = O.hardware_threads()            # CPUs this process may run on (sched_getaffinity)
    try:                                      # ... capped by a cgroup CPU quota, if the box sets one
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass

    def passes(fn, seconds, at_least=3):
        times = []
        t_end = time.perf_counter() + seconds
        while len(times) < at_least or time.perf_counter() < t_end:
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return times

    # seconds per point: grow the probe until one pass takes 0.1 s (thread start-up dominates small passes)
    n0 = 1 << 12
    while True:
        fn0, spp, what = _cpu_leg(workload, n0, threads)
        fn0()
        t = min(passes(fn0, 0.0, 3))
        if t >= 0.1 or n0 >= 1 << 20:       # (thread start-up and first-touch page faults dominate shorter passes: only a pass
            break                           # of 0.1 s, or of 2^20 points, says what all threads do)
        n0 <<= 4
    per_point = t / n0
    del fn0
    allocating = workload.startswith(_ALLOCATING)
    legs = 2 if allocating else 1
    n_full = 1 << CPU_FULL_LOG2
    if full and per_point * n_full * 3 * legs <= target_seconds:
        n, budget, protocol = n_full, 0.0, "BASELINE.md section 3 (>= 2^24 points, best of 3)"
    else:
        # one pass of about a second -- a third of the leg's share of the target if that is shorter
        share = target_seconds / legs
        n = int(min(1 << 22, max(1 << 12, min(1.0, share / 3.0) / max(per_point, 1e-9))))      # (2^22: generating the inputs counts too)
        n = 1 << (n.bit_length() - 1)
        budget, protocol = share, f"bounded sample: a 2^24-point pass takes {per_point * n_full:.1f} s here"
    fn, spp, what = _cpu_leg(workload, n, threads)
    fn()                                                           # touch pages
    times = passes(fn, budget)
    alloc_times = None
    if allocating:
        O.set_closure_alloc(True)
        try:
            alloc_times = passes(fn, budget)
        finally:
            O.set_closure_alloc(False)
    best, mean = min(times), sum(times) / len(times)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    rate = lambda t: round(spp * n / t / 1e9, 6)
    rec = {"value": rate(best), "unit": "Gsamples/s", "cores": threads, "kind": "port",
           "per_core_msamples": round(spp * n / best / 1e6 / threads, 3), "cpu_model": model,
           "mean_value": rate(mean), "points": n, "passes": len(times), "protocol": protocol,
           "note": "port of the reference's scalar closure code without its per-closure heap allocation "
                   "(src/rlGgx.h:152 make_shared): the fast end of the bracket; with_alloc is the slow end",
           "sample_short": f"2^{n.bit_length() - 1} pts x {spp}, best of {len(times)}, {what}, {threads} thr",
           "sample": f"{n} points ({spp} samples each) of the same seeded workload, best of {len(times)} passes "
                     f"({sum(times):.1f} s of CPU work), oracle/rls_oracle.c {what} on {threads} threads; {protocol}"}
    if alloc_times:
        ab = min(alloc_times)
        rec["with_alloc"] = {"value": rate(ab), "kind": "port+alloc", "per_core_msamples": round(spp * n / ab / 1e6 / threads, 3),
                             "passes": len(alloc_times),
                             "what": "the same passes with one 48-byte malloc + free per GgxSampler constructor, where the "
                                     "reference does std::make_shared<NDSampler> (src/rlGgx.h:152)"}
    else:
        rec["with_alloc"] = {"value": rec["value"], "kind": "port+alloc", "what": "this closure's reference constructor "
                             "allocates nothing (no make_shared in src/rlDisney.cpp / src/rlSss.cpp): same as port"}
    return rec


def loaded_code(kernel: str | None = None) -> dict:
    """Which device code is being timed: the ids of the library this process has loaded (rlshaders_amd/codeid.py: sha-256 over
    the gfx950 code objects inside the .so) and of the code object that holds `kernel`."""
    from rlshaders_amd import _capi
    from rlshaders_amd.codeid import device_code
    try:
        return device_code(_capi.loaded_path()).record(kernel)
    except Exception as e:                                # noqa: BLE001   (an unreadable library must not cost the line)
        return {"library_id": None, "unit_id": None, "error": f"{type(e).__name__}: {e}"}


def profile_record(workload: str, math: str, suffix: str, key: str, code: dict | None = None):
    """The newest committed PMC record for this workload and arithmetic mode (profiles/*_<suffix>.json), or None.
    Counters come from separate `rocprofv3 --pmc` passes of this very command (tools/profile_workload.sh); they cannot
    be collected inside an un-profiled run, so the line says which file each one comes from -- and the file says which
    BINARY it was taken on (`code`: library_id / unit_id, written by the summarisers from the bench line of the profiled
    session).  With `code` (the ids of the library loaded now, loaded_code()) a record is usable only if it describes the same
    machine code: same kernel_id (the kernel's instructions, descriptor and constants), else same unit_id (the code object that
    holds it), else -- for records that name neither -- same library_id.  The
    newest usable record wins; if there is none the newest record comes back marked `stale` and the caller drops every
    counter-derived key."""
    best, newest = None, None
    for p in sorted((ROOT / "profiles").glob(f"*_{suffix}.json")):
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        if d.get("workload") == workload and d.get("math", "exact") == math and d.get(key):
            d["file"] = f"profiles/{p.name}"
            newest = d
            if code is None or code_matches(d.get("code"), code):
                best = d
    if best is None and newest is not None:
        newest["stale"] = True
        return newest
    return best


def code_matches(recorded: dict | None, live: dict) -> bool:
    """does a profile's `code` stamp describe the device code loaded now?"""
    if not recorded:
        return False                                      # an unstamped profile proves nothing about this binary
    # the kernel's own code (when both ids follow the same definition: rlshaders_amd/codeid.py KERNEL_ID_SCHEME); else the code
    # object that holds it
    if recorded.get("kernel_id") and live.get("kernel_id") and recorded.get("kernel_id_scheme", 1) == live.get("kernel_id_scheme", 1):
        return recorded["kernel_id"] == live["kernel_id"]
    if recorded.get("unit_id") and live.get("unit_id"):
        return recorded["unit_id"] == live["unit_id"]
    return bool(recorded.get("library_id")) and recorded.get("library_id") == live.get("library_id")


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


def roofline_record(wl, n: int, kernel_ms: float, math: str, clock: dict | None = None) -> dict:
    """The `roofline` object of one measured workload (module docstring: "Roofline accounting")."""
    launches = wl.launches_per_step
    bytes_per_launch = n * wl.bytes_per_point / launches              # algorithmic bytes of one kernel launch
    launch_ms = kernel_ms / launches                                   # average duration of one kernel launch
    sec = launch_ms * 1e-3
    achieved_gbs = bytes_per_launch / sec / 1e9
    # which machine code this is, and which committed counter records describe it: a record taken on other device code is
    # STALE -- its numbers are dropped from the line (counter_source.stale says so), never quoted beside a live `frac`
    kernel = wl.kernel.format(m=1 if math == "fast" else 0)          # as rocprofv3 --kernel-trace names it
    code = loaded_code(kernel)
    stale = []

    def counters(suffix, key):
        d = profile_record(wl.name, math, suffix, key, code)
        if d is not None and d.get("stale"):
            stale.append({"file": d["file"], "taken_on": d.get("code") or "unstamped"})
            return None
        return d

    tr = counters("traffic", "hbm_bytes_per_launch")
    fl = counters("flops", "flops_per_point")
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
                           "note": "rocprofv3 --pmc passes of this command (tools/profile_workload.sh), not this run; used "
                                   "only when taken on the device code loaded now (`code`, rlshaders_amd/codeid.py)",
                           "stale": False},
        "code": code,
        "kernel": kernel,
        "kernel_ms": round(launch_ms, 5), "launches_per_step": launches,
        "algorithmic_bytes_per_point": wl.bytes_per_point,
        "algorithmic_bytes_per_launch": int(bytes_per_launch)})
    # the shader clock the kernel held.  Live: in-kernel stamps of the stamped instantiation, launched right behind the timed
    # launches (measure()); cross-check: profiles/*_clock.json (GRBM_GUI_ACTIVE / 8 / kernel time, a rocprofv3 --pmc pass, and
    # the stamps after seconds of sustained load -- tools/profile_workload.sh)
    ck = counters("clock", "effective_clock_ghz")
    ghz = clock["effective_clock_ghz"] if clock else (ck["effective_clock_ghz"] if ck else None)
    if ghz:
        roof["effective_clock_ghz"] = ghz
        roof["clock_source"] = ("live: s_memtime / s_memrealtime stamps around the kernel body, stamped instantiation, "
                                f"{clock['launches']} launches right behind the timed ones, median of {clock['workgroups']} workgroups"
                                if clock else ck["file"])
        if issue:
            # the issue-slot ceiling at the clock the chip actually ran: one wave64 VALU instruction per SIMD per two cycles
            roof["issue_slot_frac_at_clock"] = round(issue * 2.4 / ghz, 4)
        if clock:
            roof["clock"] = clock
        if ck:
            roof["clock_profile"] = {k: ck.get(k) for k in ("file", "effective_clock_ghz", "grbm_clock_ghz", "sustained_clock_ghz")
                                     if ck.get(k) is not None}
    # how busy the issue port is, measured: share of the launch's SIMD quad-cycles in which the vector ALU holds an instruction
    # (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2 over SQ_CYCLES, dispatches of this kernel by exact name: tools/pmc_stalls.sh)
    stl = counters("stalls", "valu_port")
    if stl:
        roof["valu_busy_frac"] = stl["valu_port"]["busy"]
        roof["valu_port"] = {k: stl["valu_port"][k] for k in ("holding_two", "holding_one", "idle", "instructions_issued_in_pairs")
                             if k in stl["valu_port"]}
        roof["valu_port"]["source"] = stl["file"]
    if stale:
        roof["counter_source"]["stale"] = True
        roof["counter_source"]["stale_files"] = stale
        roof["counter_source"]["stale_note"] = ("these records were taken on other device code than the library loaded now: "
                                                "every counter-derived key they would have filled is null or absent")
    if wl.survey_bytes:
        roof["survey_bytes_per_point"] = wl.survey_bytes
        roof["frac_survey_bytes"] = round(n * wl.survey_bytes / launches / sec / 1e9 / HBM_PEAK_GBS, 4)
    return roof


STAMPED = ("ggx_reflect_refract", "disney_integrate", "sss_probe", "skin")      # workloads whose kernel has a stamped instantiation


def measure(R, ctx, ranks, torch, name: str, log2n: int, steps: int, warmup: int, math: str, candidates: int,
            chunk_log2: int, other_mode: bool, depth: int = 3, clock: bool = True, sustain: float = 0.0):
    """Build one workload on this rank's shard, warm it up, time `steps` passes between barriers.
    -> dict: wl, n, elapsed [s, max over ranks of each rank's own K steps], wall [s, rank 0's clock from the opening barrier's
    exit to the closing barrier's exit: last end - first start], kernel_ms [per step, max over ranks], my_kernel_ms,
    other_ms [kernel ms per step in the other arithmetic mode or None], clock [dict or None]"""
    from rlshaders_amd.sharding import shard_range
    world, rank = ranks.world, ranks.rank
    n = 1 << log2n
    # weak scaling: the job is world * n points, rank g owns the index range [g*n, (g+1)*n)
    first, count = shard_range(world * n, rank, world)
    assert count == n
    wl = make_workload(R, ctx, name, n, first=first, candidates=candidates, chunk_log2=chunk_log2, depth=depth)
    torch.cuda.synchronize()
    if sustain > 0.0:
        t_end = time.perf_counter() + sustain
        while time.perf_counter() < t_end:
            for _ in range(8):
                wl.launch()
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
    wall = time.perf_counter() - t0        # barrier exit to barrier exit on this rank: includes the skew between the ranks
    my_kernel_ms = ctx.timer_elapsed_ms() / max(steps, 1)
    if wl.bound == "pcie":                 # the pipeline runs on its own streams and returns when the batch is back in host
        my_kernel_ms = elapsed / max(steps, 1) * 1e3      # memory: the pass is timed on the host clock
    elapsed, kernel_ms = ranks.max_over_ranks([elapsed, my_kernel_ms])
    clock_rec = None
    if clock and name in STAMPED and wl.launches_per_step == 1:
        # the clock the kernel ran at: the same K launches again, right behind the timed ones, as the stamped instantiation
        # (rls_diag_clock_stamps_*: s_memtime / s_memrealtime around the kernel body, first wave of every workgroup)
        ctx.clock_stamps_begin()
        try:
            ctx.timer_start()
            for _ in range(steps):
                wl.launch()
            ctx.timer_stop()
            stamped_ms = ctx.timer_elapsed_ms() / max(steps, 1)
            stamps = ctx.clock_stamps_read()
        finally:
            ctx.clock_stamps_end()
        # (a small batch of the n^2-spp integrator runs several lanes per point: that instantiation carries no stamps)
        if len(stamps):
            clock_rec = R.Context.clock_from_stamps(stamps)
            clock_rec["stamped_kernel_ms"] = round(stamped_ms, 5)
            clock_rec["launches"] = steps
            clock_rec["sustained_s_before"] = sustain
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
    return {"wl": wl, "n": n, "elapsed": elapsed, "wall": wall, "kernel_ms": kernel_ms, "my_kernel_ms": my_kernel_ms,
            "other_ms": other_ms, "clock": clock_rec}


# what tests/test_gpu_fast_mode.py holds RLS_MATH_FAST to (the measurement: profiles/r04_fast_conditioning.json)
FAST_PARITY = "opt-in, uncredited; protocol (3): 98.8-99.9 % of outliers where the oracle moves in its 1-ulp box, 4e-5 of points > 8x"

HEADLINE_MAX_BYTES = 1800          # the driver keeps a 2000-character tail of stdout: the last line must fit inside it


def _short(text: str, limit: int) -> str:
    return text if len(text) <= limit else text[:limit - 3] + "..."


def headline(detail: dict) -> dict:
    """The compact last line: the contract's keys, `roofline`, `cpu_baseline`, a `ranks` summary and the other arithmetic
    mode -- nothing whose size grows with the rank count or with the number of workloads."""
    pick = lambda d, keys: {k: d[k] for k in keys if k in d}
    line = pick(detail, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_wall", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data"))
    cfg = detail["config"]
    line["config"] = pick(cfg, ("workload", "name", "baseline_config", "math", "points_per_gpu", "points_total",
                                "samples_per_point", "sharding"))
    line["config"]["workload"] = _short(cfg["workload"], 120)
    line["roofline"] = pick(detail["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                                                 "algorithmic_bytes_per_launch", "issue_slot_frac", "effective_clock_ghz",
                                                 "issue_slot_frac_at_clock", "valu_busy_frac"))
    if detail["roofline"].get("code"):
        line["roofline"]["library_id"] = detail["roofline"]["code"].get("library_id")
        line["roofline"]["counters_stale"] = bool(detail["roofline"].get("counter_source", {}).get("stale"))
    if "cpu_baseline" in detail:
        cb = detail["cpu_baseline"]
        line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind"))
        if "with_alloc" in cb:
            line["cpu_baseline"]["with_alloc"] = pick(cb["with_alloc"], ("value", "kind"))
        line["cpu_baseline"]["cpu_model"] = _short(cb.get("cpu_model", ""), 48)
        line["cpu_baseline"]["sample"] = _short(cb.get("sample_short", cb.get("sample", "")), 160)
    line["ranks"] = pick(detail["ranks"], ("ranks_seen", "world_size", "backend", "distinct_devices", "min_kernel_ms",
                                           "max_kernel_ms"))
    if "validation" in detail:
        line["validation"] = pick(detail["validation"], ("checksum",))
    if "other_math_mode" in detail:
        line["other_math_mode"] = pick(detail["other_math_mode"], ("math", "value", "hbm_frac"))
        line["other_math_mode"]["parity"] = _short(detail["other_math_mode"].get("parity", ""), 120)
    # The line must fit whatever the strings in it grow to: shorten the free-text fields, then drop optional objects one at a
    # time; the contract's keys, `roofline` and `cpu_baseline`'s numbers are never dropped.  No assertion on the output path:
    # a line is always printed.
    fits = lambda: len(json.dumps(line)) < HEADLINE_MAX_BYTES
    for obj, key in (("other_math_mode", "parity"), ("cpu_baseline", "sample"), ("config", "workload"), ("config", "sharding"),
                     ("cpu_baseline", "cpu_model"), ("roofline", "kernel")):
        if fits():
            break
        if obj in line and key in line[obj]:
            line[obj][key] = _short(str(line[obj][key]), 24)
    for obj, key in (("other_math_mode", None), ("validation", None), ("ranks", None), ("cpu_baseline", "sample"),
                     ("cpu_baseline", "cpu_model"), ("config", "workload"), ("config", "sharding")):
        if fits():
            break
        if key is None:
            line.pop(obj, None)
        elif obj in line:
            line[obj].pop(key, None)
    for k in ("metric", "unit", "data", "dtype"):
        if not fits():
            line[k] = _short(str(line[k]), 40)
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


RANK_FAILED, BAD_LAUNCH = 1, 2          # exit codes: a rank raised / the launch itself cannot work (too few GPUs, no GPU)


def error_line(message: str, rank, world: int, **more) -> None:
    """One JSON line that says what went wrong and where -- on stdout, where the driver looks for the result line (a reader
    that parses the last line finds `error` instead of `value`), and on stderr."""
    line = json.dumps(dict({"error": message, "rank": rank, "world_size": world, "bench": "bench.py"}, **more))
    # several ranks may say this at the same moment into one pipe: each line goes out in ONE write (atomic below PIPE_BUF), so
    # two ranks' lines cannot interleave whatever the buffering of sys.stdout is (print() under PYTHONUNBUFFERED writes the text
    # and the newline separately)
    data = (_short(line, 3500) + "\n").encode()
    for stream, fd in ((sys.stderr, 2), (sys.stdout, 1)):
        try:
            stream.flush()
            os.write(fd, data)
        except OSError:
            pass


def visible_gpus() -> int:
    """GPUs a rank could use, counted in a throwaway CHILD process: the parent of a bare `--gpus N` launch never imports torch
    or touches the HIP runtime, so starting the ranks afterwards is an ordinary spawn from a GPU-less process whatever
    torch.cuda.device_count() does inside (on this image it does not initialise the runtime; elsewhere it may)."""
    try:
        p = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                           text=True, timeout=300)
        return int(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else 0
    except Exception:                                    # noqa: BLE001
        return 0


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` launched bare: check that N ranks CAN run before anything is spawned, then start one rank
    per GPU as CHILD processes (never a re-exec of this process) and return their exit code."""
    backend = os.environ.get("RLS_DIST_BACKEND", "nccl")
    have = visible_gpus()
    if have == 0:
        error_line("no GPU visible; the closures only run on the HIP path", None, args.gpus)
        return BAD_LAUNCH
    if backend == "nccl" and args.gpus > have:
        error_line(f"--gpus {args.gpus} but only {have} GPU(s) visible: one rank per GPU over RCCL needs {args.gpus} devices "
                   "(nothing was launched)", None, args.gpus, gpus_visible=have)
        return BAD_LAUNCH
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
           *sys.argv[1:]]
    return subprocess.call(cmd)


def main():
    args = parse_args()
    # the host driver only does dmabuf IPC: RCCL between processes needs this set before HIP initialises (harmless otherwise)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    try:
        run_rank(args, world)
    except SystemExit as e:
        if e.code in (None, 0):
            raise
        # a refusal with a message (two ranks on one GPU, no GPU, ...): the same one-line shape as any other failure
        error_line(str(e.code), rank, world)
        sys.stdout.flush()
        os._exit(e.code if isinstance(e.code, int) else BAD_LAUNCH)
    except BaseException as e:                            # noqa: BLE001
        # A rank that dies must take the job down NOW and say why: print the line, then leave without running the process
        # group's destructors (they can wait on peers that sit in a barrier).  torchrun sees the non-zero child, stops the
        # other ranks and exits non-zero itself.
        import traceback
        traceback.print_exc(file=sys.stderr)
        error_line(f"{type(e).__name__}: {e}", rank, world)
        sys.stdout.flush()
        os._exit(RANK_FAILED)


DEADLINE_S = 1200.0         # RLS_BENCH_DEADLINE_S: a rank still running after this long gives up (the default run takes ~30 s)


def start_deadline(rank: int, world: int):
    """A rank that hangs where no collective's timeout can see it -- a launch that never completes, a wedged device, a host
    call that blocks -- must not burn the caller's half hour: after DEADLINE_S a timer thread prints the one-line error and
    leaves the process (HIP and torch calls release the GIL, so the timer fires while the main thread is stuck inside one)."""
    import threading
    limit = float(os.environ.get("RLS_BENCH_DEADLINE_S", DEADLINE_S))

    def expire():
        error_line(f"deadline: still running after {limit:g} s (RLS_BENCH_DEADLINE_S); giving up", rank, world)
        sys.stdout.flush()
        os._exit(RANK_FAILED)

    timer = threading.Timer(limit, expire)
    timer.daemon = True
    timer.start()
    return timer


def run_rank(args, world: int):
    deadline = start_deadline(int(os.environ.get("RANK", "0")), world)
    if os.environ.get("RLS_BENCH_STALL_S"):                 # tests: a rank that hangs before it gets anywhere
        time.sleep(float(os.environ["RLS_BENCH_STALL_S"]))
    import torch
    import rlshaders_amd as R
    from rlshaders_amd.sharding import Ranks

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank_env = int(os.environ.get("RANK", "0"))
    # one rank per GPU.  RLS_DIST_BACKEND=gloo (tests only) lets several ranks share one GPU so that the
    # multi-rank control path can be exercised on a single-GPU box, where RCCL refuses duplicate devices.
    backend = os.environ.get("RLS_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()                      # (does not initialise HIP)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    if have == 0 or (backend == "nccl" and local_world > have):
        # no GPU, or launched under torchrun with more ranks than devices: every rank leaves at once, before the rendezvous
        # (no peer waits for anyone).  EVERY rank says why, in one line each: whichever exits first makes torchrun stop the
        # others, so a line left to one designated rank can be lost.
        error_line("no GPU visible; the closures only run on the HIP path" if have == 0 else
                   f"{local_world} ranks on this node but only {have} GPU(s) visible: one rank per GPU over RCCL needs "
                   f"{local_world} devices", rank_env, world, gpus_visible=have)
        sys.stdout.flush()
        os._exit(BAD_LAUNCH)
    device_index = local_rank % have if backend != "nccl" else local_rank

    block = {"all": BLOCK_ALL, "configs": BLOCK_CONFIGS, "none": []}[args.workloads]
    if args.block_log2_points is not None:
        block = [(w, args.block_log2_points, min(st, 5)) for w, _, st in block]
        block = sorted(set(block), key=block.index)
    block = [b for b in block if not (b[0] == args.workload and b[1] == args.log2_points)]

    # ---- the CPU legs: rank 0's host cores, BEFORE the process group is formed and before any GPU work -----------------
    # With N > 1 the other ranks are then asleep in the store rendezvous of init_process_group (a socket wait), not spinning
    # in a device barrier on the cores the oracle's threads want.  Bounded so that the ranks never wait long: <= ~15 s in all.
    cpu_records = {}
    if rank_env == 0 and not args.no_cpu_baseline:
        head_s, block_s = args.cpu_seconds, args.block_cpu_seconds
        if world > 1:
            head_s, block_s = min(head_s, 8.0), min(block_s, 2.0)
        t_cpu = time.perf_counter()
        for key, name, seconds, full in [("headline", args.workload, head_s, True)] + \
                [(f"{n_}@{l2}", n_, block_s, False) for n_, l2, _ in block]:
            try:
                cb = cpu_baseline(name, seconds, full=full)
            except Exception as e:                        # noqa: BLE001   (the GPU number must not be lost to the CPU leg)
                cb = {"error": f"{type(e).__name__}: {e}"}
            if cb is not None:
                cb["when"] = ("before the process group is formed and before any GPU work" +
                              (f": the other {world - 1} ranks sleep in the store rendezvous meanwhile" if world > 1 else ""))
                if world > 1:
                    cb["ranks_waiting"] = world - 1
            cpu_records[key] = cb
        cpu_seconds_total = time.perf_counter() - t_cpu
        for cb in cpu_records.values():
            if cb is not None:
                cb["all_cpu_legs_seconds"] = round(cpu_seconds_total, 2)

    torch.cuda.set_device(device_index)                    # (the first call that touches the HIP runtime: after the CPU legs)
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
    if os.environ.get("RLS_BENCH_FAIL_RANK") == str(rank):          # tests: a rank that dies while the others wait for it
        raise RuntimeError(f"RLS_BENCH_FAIL_RANK={rank}: this rank was told to fail")

    m = measure(R, ctx, ranks, torch, args.workload, args.log2_points, args.steps, args.warmup, args.math, args.arena_candidates,
                args.chunk_log2, not (args.checksum or args.no_other_mode),       # (the other mode would overwrite the outputs)
                args.pipeline_depth, clock=not args.no_clock, sustain=args.sustain_seconds)
    wl, n, elapsed, kernel_ms, my_ms, other_ms = m["wl"], m["n"], m["elapsed"], m["kernel_ms"], m["my_kernel_ms"], m["other_ms"]
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
        roof = roofline_record(wl, n, kernel_ms, args.math, m["clock"])
        bytes_per_step = n * wl.bytes_per_point
        detail = {
            "metric": "BSDF Gsamples/sec (eval+sample+pdf)",
            "value": round(value, 4),
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            # rank 0's clock from the opening barrier's exit to the closing barrier's exit (last end - first start as rank 0
            # sees it): >= ms_per_step by the barrier skew between the ranks; equal within noise on one GPU
            "ms_per_step_wall": round(m["wall"] / args.steps * 1e3, 5),
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
        if cpu_records.get("headline"):
            detail["cpu_baseline"] = cpu_records["headline"]
    del wl, m
    torch.cuda.empty_cache()

    # ---- the other configurations and verbs, same process, each with its own warm-up --------------------------------
    records = []
    for name, log2n, steps in block:
        warm = max(10, args.warmup)
        warm = warm if "_host" not in name else 2
        try:
            bm = measure(R, ctx, ranks, torch, name, log2n, steps, warm, args.math, 1, args.chunk_log2, False,
                         args.pipeline_depth, clock=not args.no_clock)
            w, bn, el, kms, my = bm["wl"], bm["n"], bm["elapsed"], bm["kernel_ms"], bm["my_kernel_ms"]
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
                   "ms_per_step_wall": round(bm["wall"] / steps * 1e3, 5),
                   "roofline": roofline_record(w, bn, kms, args.math, bm["clock"]), "per_rank_kernel_ms": prm}
            if cpu_records.get(f"{name}@{log2n}"):
                rec["cpu_baseline"] = cpu_records[f"{name}@{log2n}"]
            records.append(rec)
        del w, bm
        torch.cuda.empty_cache()

    ranks.close()
    deadline.cancel()
    if rank == 0:
        emit(detail, records, args.records_file)


if __name__ == "__main__":
    main()
