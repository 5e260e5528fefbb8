#!/usr/bin/env python3
"""Print the measurement table of DESIGN.md section 5 from profiles/<tag>_*: one row per workload, nothing typed by hand.
usage: tools/measurement_table.py [tag]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROWS = (("ggx_reflect_refract", "2"), ("sss_probe", "4 (kernel; 2²⁸ points over 8 GPUs)"), ("skin", "5 (kernel; 2³⁰ points over 8 GPUs)"),
        ("disney_integrate", "3, mode R (reduced)"), ("disney_stream", "3, mode S (streamed, 64 chunks of 2²⁰ points)"),
        ("ggx_reflect_refract_uniform", "2 with uniform node parameters"), ("ggx_reflect", "— (reflect triple, fused)"),
        ("ggx_eval", "— (`evalBrdf` alone)"), ("ggx_pdf", "— (`evalPdf` alone)"),
        ("disney_triple_diffuse", "— (rlDisney one-sample triple, diffuse lobe)"),
        ("disney_triple_glossy", "— (rlDisney one-sample triple, glossy lobe)"),
        ("disney_triple_glossy_uniform", "— (the same with uniform node parameters)"),
        ("disney_triple_glossy_colour_map", "— (the same with base_color textured, the scalars uniform)"), ("nd_sample", "4, profile-only variant"),
        ("sss_probe_uniform", "4 with a uniform scatter distance"), ("skin_uniform", "5 with uniform node parameters"),
        ("skin_integrate", "— (rlSkin `shader_evaluate`, 16 samples per layer, no lights)"),
        ("ggx_shade", "— (rlGgx `shader_evaluate`, whole: two lights × 48 + 3 × 16 samples per point)"),
        ("disney_shade", "— (rlDisney `shader_evaluate`, whole: two lights × 48 + 2 × 16 samples per point)"),
        ("disney_direct", "— (rlDisney light loop, two lights × 48 samples per point)"),
        ("ggx_direct", "— (rlGgx light loop, one light, 48 samples per point)"),
        ("sss_scatter", "— (`integrateScatter`, 16 probe rays per point)"))


def load(w, kind):
    p = ROOT / "profiles" / f"{TAG}_{w}_{kind}.json"
    return json.loads(p.read_text()) if p.exists() else None


def thousands(x):
    return f"{x:,.0f}".replace(",", " ")


print("| workload (`bench.py --workload`) | BASELINE config | time per pass (rocprofv3 average) | throughput | roofline | HBM bytes the counters saw ÷ algorithmic (fraction of 8 TB/s on them) | VALU / SALU instr. per point, issue-slot fraction | clock GHz: stamps after 2 s of load / GRBM; issue-slot fraction at that clock | vector ALU busy (two / one / idle) | CPU baseline (oracle, all host threads) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for w, cfg in ROWS:
    b, f, t, tr = load(w, "bench"), load(w, "flops"), load(w, "traffic"), load(w, "bench_trace")
    if b is None:
        continue
    r = b["roofline"]
    prof = tr["roofline"]["kernel_ms"] if tr else None
    ms = b["ms_per_step"]
    lp = r.get("launches_per_step", 1)
    tm = f"{ms:.3f} ms" if ms < 20 else f"{ms:.1f} ms"
    if lp > 1:
        tm += f" = {lp} × {r['kernel_ms']:.3f} ms"
    sec = r["kernel_ms"] * 1e-3
    ppl = b["config"]["points_per_gpu"] / lp
    if r["bound"] == "hbm":
        roof = f"{r['frac']:.3f} of 8 TB/s ({r['achieved'] / 1000:.2f} TB/s on {r['algorithmic_bytes_per_point']} B/point"
        if r.get("survey_bytes_per_point"):
            roof += f"; {r['frac_survey_bytes']:.3f} on SURVEY's {r['survey_bytes_per_point']} B"
        roof += ")"
    else:
        roof = f"VALU: {r['achieved']:.1f} of {r['peak']} TFLOP/s = {r['frac']:.2f} ({thousands(r['flops_per_point'])} executed flops per point)"
    ratio = f"{t['ratio_to_algorithmic']:.3f} ({t['hbm_bytes_per_launch'] / sec / 8e12:.3f})" if t else "—"
    if f:
        ip = f["instructions_per_point"]
        valu = f"{thousands(ip['SQ_INSTS_VALU'])} / {thousands(ip.get('SQ_INSTS_SALU', 0))}, {ip['SQ_INSTS_VALU'] * ppl / 64 / sec / 1.2288e12:.2f}"
    else:
        valu = "—"
    unit = b["unit"]
    ck, st = load(w, "clock"), load(w, "stalls")             # round 5: the measured clock and the issue port (tools/pmc_stalls.sh)
    clock = "—"
    if ck:
        ghz = ck["effective_clock_ghz"]
        clock = f"{ck.get('sustained_clock_ghz', '—')} / {ck.get('grbm_clock_ghz', '—')}"
        if f:
            clock += f"; {f['instructions_per_point']['SQ_INSTS_VALU'] * ppl / 64 / sec / (1024 * ghz * 1e9 / 2):.2f}"
    port = "—"
    if st and st.get("valu_port"):
        v = st["valu_port"]
        port = f"{v['busy']:.3f} ({v['holding_two']:.2f} / {v['holding_one']:.2f} / {v['idle']:.2f})"
    cb = b.get("cpu_baseline")
    print(f"| `{w}` | {cfg} | {tm}" + (f" ({prof:.3f})" if prof and lp == 1 else "") + f" | {b['value']:.1f} {unit} | {roof} | {ratio} | {valu} | "
          f"{clock} | {port} | " + (f"{cb['value']:.3f} ({cb['cores']} threads)" if cb else "—") + " |")
for name, label in (("ggx_reflect_bench_only", "`ggx_reflect` | — (reflect triple only)"), ("bench_fast", "`ggx_reflect_refract --math fast` | 2, FAST arithmetic")):
    p = ROOT / "profiles" / f"{TAG}_{name}.json"
    if p.exists():
        b = json.loads(p.read_text()); r = b["roofline"]
        print(f"| {label} | {b['ms_per_step']:.3f} ms | {b['value']:.1f} {b['unit']} | {r['frac']:.3f} of 8 TB/s | — | — | " +
              (f"{b['cpu_baseline']['value']:.3f}" if b.get("cpu_baseline") else "—") + " |")
