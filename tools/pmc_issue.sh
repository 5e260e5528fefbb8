#!/bin/bash
# Issue-side counters of one workload's dominant kernel, per launch: instruction counts by class, VALU-busy and wait cycles.
# usage: [ENV=...] tools/pmc_issue.sh <tag> <workload> [log2 points]   -> appends to gpurun_out/pmc_issue.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=$1; W=$2; L=${3:-26}
d=gpurun_out/issue_$TAG; rm -rf $d; mkdir -p $d
SHORT="python3 bench.py --workload $W --log2-points $L --steps 3 --warmup 1 --no-cpu-baseline --arena-candidates 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $d/a -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $d/b -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $d/c -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $d/e -- $SHORT > /dev/null 2>&1
python3 - "$d" "$TAG $W 2^$L" <<'PY' >> gpurun_out/pmc_issue.txt
import csv, glob, sys
d, w = sys.argv[1:3]
acc = {}
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in ("skin_kernel", "ggx_kernel", "sss_kernel", "disney_kernel", "integrate_kernel", "shade_kernel", "scatter_kernel", "direct_kernel")): continue
        per.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
        per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, _), v in per.items():
        acc.setdefault(c, []).append(v)
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(w, {k: f"{v:.4g}" for k, v in sorted(m.items())})
if m.get("SQ_WAVES"):
    wv = m["SQ_WAVES"]
    print("   per wavefront: " + "  ".join(f"{k[3:]} {m[k] / wv:.1f}" for k in sorted(m) if k.startswith("SQ_INSTS")))
if m.get("SQ_WAVE_CYCLES"):
    print(f"   valu-busy / wave_cycles {m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES']:.4f}  wait_inst_any / wave_cycles {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.4f}")
PY
tail -3 gpurun_out/pmc_issue.txt
