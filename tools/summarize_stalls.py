#!/usr/bin/env python3
"""profiles/<tag>_<workload>_stalls.json from the passes of tools/pmc_stalls.sh: the SQ busy / wait / issue counters of the
workload's dominant kernel BY EXACT NAME (bench.py's roofline.kernel), per launch, and what they say about the issue port.
usage: summarize_stalls.py <tag> <workload> [clock_ghz]

Units (MI355X_MICROARCH.md, instruction-cost table): SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count QUAD-cycles (4
shader cycles), summed over the waves; SQ_INSTS_* count wave-instructions; GRBM_GUI_ACTIVE counts shader cycles summed over
the 8 XCDs.  Derived:
  clock_ghz          GRBM_GUI_ACTIVE / 8 / kernel duration (the dispatch's own End - Start timestamps of the same pass)
  valu_busy_frac     4 x SQ_ACTIVE_INST_VALU / (duration x clock x 1024 SIMDs): share of all SIMD-cycles of the launch in
                     which the SIMD's vector ALU was executing an instruction of some wave
  cycles_per_valu    4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: shader cycles the ALU is held per wave64 VALU instruction (the
                     mix's average; 2 = the nominal issue rate bench.py prices issue_slot_frac against)
  wait_inst_frac     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of a wave's lifetime spent waiting for ANY instruction to issue
                     (with 8 waves per SIMD taking turns on one port this is (waves - 1) / waves when the port never idles)
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(dirname, kernel):
    """{counter: [value per dispatch]}, {dispatch: duration ns} for dispatches whose demangled name contains `kernel`"""
    vals, dur = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kernel not in r["Kernel_Name"]:
                continue
            key = (f, r["Dispatch_Id"])
            acc.setdefault((r["Counter_Name"], key), 0.0)
            acc[(r["Counter_Name"], key)] += float(r["Counter_Value"])
            dur[key] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for (c, key), v in acc.items():
            vals.setdefault(c, {})[key] = v
    return vals, dur


def main(argv=None, root=ROOT):
    argv = sys.argv if argv is None else argv
    tag, w = argv[1], argv[2]
    d = os.path.join(root, "gpurun_out", f"stalls_{tag}_{w}")
    recs = [json.loads(l) for l in open(os.path.join(d, "bench.json")).read().splitlines() if l.startswith("{")]
    line = [r for r in recs if r.get("record") == "headline_detail"][-1]
    kernel = line["roofline"]["kernel"]
    vals, dur = per_dispatch(d, kernel)
    if not vals:
        raise SystemExit(f"no dispatch of {kernel} under {d}")
    # median over the dispatches of a pass (the first launches after idle run at another clock)
    mean = {c: sorted(v.values())[len(v) // 2] for c, v in vals.items()}
    out = {"workload": w, "math": line["config"]["math"], "kernel": kernel,
           # the device code these passes ran (ids computed by bench.py on the GPU box for the library it had loaded)
           "code": line["roofline"].get("code"),
           "points_per_launch": line["config"]["points_per_gpu"],
           "dispatches_per_counter": {c: len(v) for c, v in vals.items()},
           "other_kernels_counted": 0, "counters_per_launch": {c: round(v, 1) for c, v in sorted(mean.items())},
           "bench_kernel_ms_unprofiled": line["roofline"]["kernel_ms"],
           "source": "rocprofv3 --pmc, one pass per counter group (tools/pmc_stalls.sh); dispatches filtered by exact kernel name; "
                     "the profiled command launches no other closure kernel (--no-other-mode --no-clock)"}
    # clock: per dispatch of the GRBM pass, counter / 8 / that dispatch's duration
    clocks = []
    for key, v in vals.get("GRBM_GUI_ACTIVE", {}).items():
        if dur.get(key, 0) > 0:
            clocks.append(v / 8.0 / dur[key])          # cycles per ns = GHz
    d_ns = sorted(dur.values())[len(dur) // 2]
    out["kernel_ms_profiled_median"] = round(d_ns / 1e6, 5)
    if clocks:
        clocks.sort()
        out["grbm_clock_ghz"] = round(clocks[len(clocks) // 2], 4)
        out["grbm_clock_note"] = ("GRBM_GUI_ACTIVE / 8 / dispatch duration; reads high on dispatches shorter than ~0.3 ms, within "
                                  "3 % of the in-kernel clock on dispatches of 10 ms or more (MI355X_MICROARCH.md, DVFS give-back)")
    clock = float(argv[3]) if len(argv) > 3 else out.get("grbm_clock_ghz")
    g = lambda c: mean.get(c)
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        out["of_wave_cycles"] = {c: round(g(c) / wc, 4) for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY",
                                                                   "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
                                                                   "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT")
                                if g(c) is not None}
    tiles = line["config"]["points_per_gpu"] / 64.0              # one point per lane: wave-instructions per 64 points
    out["per_64_points"] = {c: round(g(c) / tiles, 1) for c in mean if c.startswith(("SQ_INSTS", "SQ_WAVE_CYCLES", "SQ_IFETCH"))}
    if g("SQ_CYCLES") and g("SQ_ACTIVE_INST_VALU") and g("SQ_ACTIVE_INST_VALU2") is not None:
        # The vector ALU of a SIMD holds, in one quad-cycle, two instructions of the 2-cycle class (fp32 fma / mul / add, moves,
        # and / add_u32), or one of the 4-cycle class (compares, selects, conversions, v_div_*, fp64), or half a transcendental
        # (profiles/r05_valu_rates.txt has each class measured in true cycles and under these counters).  SQ_CYCLES counts
        # shader cycles per shader engine (32 of them); ACTIVE_INST_VALU counts quad-cycles x instructions held, VALU2 the
        # quad-cycles holding two.
        quads = g("SQ_CYCLES") / 32.0 / 4.0 * 1024.0
        two = g("SQ_ACTIVE_INST_VALU2") / quads
        one = (g("SQ_ACTIVE_INST_VALU") - 2.0 * g("SQ_ACTIVE_INST_VALU2")) / quads
        out["valu_port"] = {"simd_quad_cycles_per_launch": round(quads, 1), "holding_two": round(two, 4), "holding_one": round(one, 4),
                            "idle": round(1.0 - two - one, 4), "busy": round(two + one, 4),
                            "what": "share of the launch's SIMD quad-cycles in which the vector ALU held two / one / no instruction"}
        if g("SQ_INSTS_VALU"):
            n = g("SQ_INSTS_VALU")
            out["valu_port"]["instructions_issued_in_pairs"] = round(2.0 * g("SQ_ACTIVE_INST_VALU2") / n, 4)
            out["valu_port"]["quad_cycles_per_instruction"] = round(quads / n, 4)
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_INSTS_VALU"):
        out["cycles_per_valu"] = round(4.0 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"), 3)
    if g("SQ_INST_CYCLES_SALU") and g("SQ_INSTS_SALU"):
        out["cycles_per_salu"] = round(4.0 * g("SQ_INST_CYCLES_SALU") / g("SQ_INSTS_SALU"), 3)
    if clock:
        out["clock_ghz_used"] = clock
        simd_cycles = d_ns * clock * 1024.0                       # ns x cycles/ns x SIMDs
        for c, k in (("SQ_ACTIVE_INST_VALU", "valu_busy_frac"), ("SQ_ACTIVE_INST_SCA", "scalar_busy_frac"),
                     ("SQ_ACTIVE_INST_VMEM", "vmem_busy_frac"), ("SQ_ACTIVE_INST_LDS", "lds_busy_frac")):
            if g(c):
                out[k] = round(4.0 * g(c) / simd_cycles, 4)
        if g("SQ_WAVE_CYCLES"):
            out["mean_waves_per_simd"] = round(4.0 * g("SQ_WAVE_CYCLES") / simd_cycles, 3)
        if g("SQ_INSTS_VALU"):
            out["valu_issue_frac_at_clock"] = round(g("SQ_INSTS_VALU") * 2.0 / simd_cycles, 4)       # 2 cycles per instruction nominal
    p = os.path.join(root, "profiles", f"{tag}_{w}_stalls.json")
    json.dump(out, open(p, "w"), indent=1)
    print(json.dumps({k: out[k] for k in out if k not in ("counters_per_launch", "source", "dispatches_per_counter")}, indent=1))
    return out


if __name__ == "__main__":
    main()
