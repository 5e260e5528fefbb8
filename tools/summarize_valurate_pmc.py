#!/usr/bin/env python3
"""What one wave64 instruction of each class does to SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_VALU2 (the counters the stall
analysis of the BASELINE kernels rests on): tools/micro/valurate 8 under rocprofv3 --pmc, one pure instruction stream per
kernel, 8 waves per SIMD.  usage: summarize_valurate_pmc.py <dir with counter_collection.csv>
Per kernel: quad-cycles the ALU is held per instruction (ACTIVE / INSTS), the share of instructions issued as one of a PAIR in
a quad-cycle (2 x VALU2 / INSTS), SIMD quad-cycles of the launch per instruction (SQ_CYCLES / 32 SEs / 4 x 1024 SIMDs / INSTS:
0.5 = two per quad-cycle = the 2-cycle class, 1 = the 4-cycle class, 2 = the 8-cycle class), and the share of the launch's
SIMD quad-cycles in which the ALU held at least one instruction."""
import csv, glob, os, sys

d = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if not name.startswith("k_"):
            continue
        per.setdefault((name, r["Dispatch_Id"], r["Counter_Name"]), 0.0)
        per[(name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (name, _, c), v in per.items():
        acc.setdefault(name, {}).setdefault(c, []).append(v)
print(f"{'kernel':10s} {'ACTIVE/INST':>12s} {'paired share':>13s} {'SIMD quads/inst':>16s} {'ALU busy':>9s}")
for name, cs in acc.items():
    m = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    n, a, v2, cyc = m.get("SQ_INSTS_VALU"), m.get("SQ_ACTIVE_INST_VALU"), m.get("SQ_ACTIVE_INST_VALU2"), m.get("SQ_CYCLES")
    if not (n and a and cyc):
        continue
    quads = cyc / 32.0 / 4.0 * 1024.0
    v2 = v2 or 0.0
    print(f"{name:10s} {a / n:12.3f} {2 * v2 / n:13.3f} {quads / n:16.3f} {(a - v2) / quads:9.3f}")
