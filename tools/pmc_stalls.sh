#!/bin/bash
# Where the waves of a workload's dominant kernel spend their cycles: busy / wait counters of the SQ, per launch.
# usage: tools/pmc_stalls.sh <workload>...   -> gpurun_out/pmc_stalls.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_stalls.txt; : > $OUT
for W in "$@"; do
  d=gpurun_out/stalls_$W; rm -rf $d; mkdir -p $d
  SHORT="python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --arena-candidates 1"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $d/a -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $d/b -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $d/c -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $d/e -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $d/f -- $SHORT > /dev/null 2>&1
  python3 - "$d" "$W" <<'PY' >> $OUT
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
wc = m.get("SQ_WAVE_CYCLES", 0) or 1
print(w)
for k in sorted(m):
    print(f"   {k:24s} {m[k]:.4g}   / wave_cycles {m[k] / wc:.4f}" + (f"   per wave {m[k] / m['SQ_WAVES']:.1f}" if m.get("SQ_WAVES") and k.startswith("SQ_INSTS") else ""))
PY
done
cat $OUT
