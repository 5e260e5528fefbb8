#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of tools/profile_workload.sh into the committed per-workload summaries:
  profiles/<tag>_<workload>_bench.json          the un-profiled bench line (and the traced one)
  profiles/<tag>_<workload>_kernel_stats.csv    rocprofv3 --kernel-trace --stats
  profiles/<tag>_<workload>_traffic.json        HBM bytes per launch (FETCH_SIZE x calibrated factor + WRITE_SIZE)
  profiles/<tag>_<workload>_flops.json          executed flops per point from the dynamic instruction mix
usage: summarize_workload.py <tag> <workload>"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(files):
    """gpurun MERGES a call's files into gpurun_out/: earlier profile passes of the same workload are still there.  Keep
    the files of the latest pass (written within a minute of the newest one)."""
    if not files:
        return []
    t = max(os.path.getmtime(f) for f in files)
    return [f for f in files if t - os.path.getmtime(f) < 60.0]


def pmc(dirname, sub):
    """mean counter value per dispatch of kernels whose name contains `sub` (sum over the XCD/SE dimensions rocprofv3
    has already aggregated)"""
    acc = {}
    for f in newest(glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                per_dispatch.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
                per_dispatch[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (c, _), v in per_dispatch.items():
            acc.setdefault(c, []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def clock_record(src, dst, tag, workload, ksub, math, code=None):
    """profiles/<tag>_<workload>_clock.json: the shader clock the kernel holds, two independent ways --
    (a) GRBM_GUI_ACTIVE / 8 / dispatch duration per dispatch of the exact kernel (rocprofv3 --pmc pass on a batch large enough
        for dispatches of >= 10 ms, where the quotient is within 3 % of the in-kernel clock: MI355X_MICROARCH.md, DVFS give-back);
    (b) the in-kernel stamps (s_memtime / s_memrealtime, stamped instantiation) after two seconds of back-to-back launches
        (bench.py --sustain-seconds 2).  `effective_clock_ghz` = (b) when there is one, else (a)."""
    rec = {"workload": workload, "math": math, "kernel": ksub, "code": code}
    clocks, durs = [], []
    for f in newest(glob.glob(os.path.join(src, "clock_grbm", "**", "*counter_collection.csv"), recursive=True)):
        per = {}                                        # rows of one dispatch (one per XCD, if rocprofv3 splits them) add up
        for r in csv.DictReader(open(f)):
            if ksub in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                e = per.setdefault(r["Dispatch_Id"], [0.0, float(r["End_Timestamp"]) - float(r["Start_Timestamp"])])
                e[0] += float(r["Counter_Value"])
        for v, d in per.values():
            if d > 0:
                clocks.append(v / 8.0 / d)
                durs.append(d / 1e6)
    if clocks:
        clocks.sort(); durs.sort()
        rec["grbm_clock_ghz"] = round(clocks[len(clocks) // 2], 4)
        rec["grbm_dispatch_ms"] = round(durs[len(durs) // 2], 4)
        rec["grbm_dispatches"] = len(clocks)
    p = os.path.join(src, "clock_sustained.json")
    if os.path.exists(p):
        recs = [json.loads(l) for l in open(p).read().splitlines() if l.startswith("{")]
        full = [r for r in recs if r.get("record") == "headline_detail"]
        if full and full[-1]["roofline"].get("clock"):
            c = full[-1]["roofline"]["clock"]
            rec["sustained_clock_ghz"] = c["effective_clock_ghz"]
            rec["stamps"] = c
            rec["sustained_kernel_ms"] = full[-1]["roofline"]["kernel_ms"]
    ghz = rec.get("sustained_clock_ghz") or rec.get("grbm_clock_ghz")
    if not ghz:
        return
    rec["effective_clock_ghz"] = ghz
    rec["source"] = ("tools/profile_workload.sh: rocprofv3 --pmc GRBM_GUI_ACTIVE (own pass) and bench.py --sustain-seconds 2 "
                     "(in-kernel stamps, rls_diag_clock_stamps_*)")
    json.dump(rec, open(os.path.join(dst, f"{tag}_{workload}_clock.json"), "w"), indent=1)
    print("clock", {k: rec[k] for k in rec if k.endswith("ghz") or k.endswith("_ms")})


def main():
    tag, workload = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{workload}")
    dst = os.path.join(ROOT, "profiles")
    line = None
    for name in ("bench.json", "bench_trace.json"):
        p = os.path.join(src, name)
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")] if os.path.exists(p) else []
        # bench.py prints the long records first and the compact headline last: keep the headline's FULL record
        recs = [json.loads(l) for l in lines]
        full = [r for r in recs if r.get("record") == "headline_detail"]
        pick = full[-1] if full else (recs[-1] if recs else None)
        if pick is not None and name == "bench.json":
            line = pick
        if pick is not None:
            json.dump(pick, open(os.path.join(dst, f"{tag}_{workload}_{name}"), "w"), indent=1)
    if line is None:
        raise SystemExit(f"no bench line under {src}")
    ksub = line["roofline"]["kernel"]
    # which binary the passes of this session ran: the ids bench.py computed ON THE GPU BOX for the library it had loaded
    # (rlshaders_amd/codeid.py).  Every summary carries it; bench.py quotes a summary only beside the same device code.
    code = line["roofline"].get("code")
    if not code or not code.get("library_id"):
        print("WARNING: the bench line carries no device-code id: these summaries will read as stale", file=sys.stderr)
    n = line["config"]["points_per_gpu"]
    math = line["config"]["math"]
    launches = line["roofline"].get("launches_per_step", 1)
    points_per_launch = n / launches
    for f in newest(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        shutil.copy(f, os.path.join(dst, f"{tag}_{workload}_kernel_stats.csv"))
        for r in csv.DictReader(open(f)):
            if ksub in r["Name"]:
                print(f"rocprofv3: {r['Name'][:60]} calls {r['Calls']} avg {float(r['AverageNs']) / 1e6:.4f} ms "
                      f"(bench line: {line['roofline']['kernel_ms']} ms)")
    fetch, write = pmc(os.path.join(src, "fetch"), ksub), pmc(os.path.join(src, "write"), ksub)
    calib = pmc(os.path.join(src, "calib"), "checksum_kernel")
    # FETCH_SIZE / WRITE_SIZE count KiB; gfx950's FETCH_SIZE under-reports streaming reads (MI355X_MICROARCH.md, HBM
    # section): the factor is re-measured on rls_checksum, a kernel of known size in the same one-dword-per-lane pattern
    factor = (4 * (1 << 28)) / (calib["FETCH_SIZE"] * 1024.0) if calib.get("FETCH_SIZE") else 2.0
    if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        rd, wr = fetch["FETCH_SIZE"] * 1024.0 * factor, write["WRITE_SIZE"] * 1024.0
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        t = {"workload": workload, "math": math, "kernel": ksub, "code": code, "points_per_launch": points_per_launch,
             "hbm_bytes_per_launch": int(round(rd + wr)),
             "read": int(round(rd)), "write": int(round(wr)), "algorithmic_bytes_per_launch": alg,
             "ratio_to_algorithmic": round((rd + wr) / alg, 4), "fetch_size_factor": round(factor, 4),
             "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_workload.sh); FETCH_SIZE x "
                       "the factor measured on rls_checksum (known byte count) in the same session"}
        json.dump(t, open(os.path.join(dst, f"{tag}_{workload}_traffic.json"), "w"), indent=1)
        print("traffic", t["hbm_bytes_per_launch"], "ratio", t["ratio_to_algorithmic"])
    clock_record(src, dst, tag, workload, ksub, math, code)
    mix = {}
    for d in ("mix_a", "mix_b", "mix_c", "mix_d"):
        mix.update(pmc(os.path.join(src, d), ksub))
    if mix:
        waves = points_per_launch / 64.0          # one point per lane: counters per wave-instruction -> per point
        per = {k: round(v / waves, 2) for k, v in mix.items() if k.startswith("SQ_INSTS")}
        g = lambda k: per.get(k, 0.0)
        flops = (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32")
                 + g("SQ_INSTS_VALU_TRANS_F32") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64")
                 + 2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64"))
        f = {"workload": workload, "math": math, "kernel": ksub, "code": code, "points_per_launch": points_per_launch,
             "flops_per_point": round(flops, 1),
             "instructions_per_point": per,
             "note": "wave-instructions per wave of 64 points = instructions per point (G = 1: one lane per point); flops = "
                     "add + mul + 2 fma + transcendental, fp32 and fp64 alike; divergent lanes count as executed",
             "source": "rocprofv3 --pmc SQ_INSTS_VALU_* in four passes (tools/profile_workload.sh)"}
        json.dump(f, open(os.path.join(dst, f"{tag}_{workload}_flops.json"), "w"), indent=1)
        print("flops/point", f["flops_per_point"], "VALU/point", per.get("SQ_INSTS_VALU"))
        # the bench lines of this session priced their VALU roofline with the flops file the snapshot carried (the previous
        # profile pass); re-price them with the mix measured in THIS session, and with the traffic measured in it
        for name in ("bench.json", "bench_trace.json"):
            q = os.path.join(dst, f"{tag}_{workload}_{name}")
            if not os.path.exists(q):
                continue
            b = json.load(open(q))
            r = b["roofline"]
            if r.get("bound") == "valu":
                tf = f["flops_per_point"] * points_per_launch / (r["kernel_ms"] * 1e-3) / 1e12
                r.update(achieved=round(tf, 3), frac=round(tf / r["peak"], 4), flops_per_point=f["flops_per_point"],
                         flops_source=f"profiles/{tag}_{workload}_flops.json (rocprofv3 --pmc passes of the same session)")
            t = os.path.join(dst, f"{tag}_{workload}_traffic.json")
            if os.path.exists(t):
                r.update(traffic=round(json.load(open(t))["hbm_bytes_per_launch"]),
                         traffic_source=f"profiles/{tag}_{workload}_traffic.json (rocprofv3 --pmc passes of the same session)")
            json.dump(b, open(q, "w"), indent=1)


if __name__ == "__main__":
    main()
