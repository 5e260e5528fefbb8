#!/bin/bash
# Instruction-cache behaviour of the pointwise kernels: rlSkin's kernel is 86 KB of code, the instruction cache 64 KB per
# pair of CUs.  usage: tools/pmc_icache.sh <workload>...   -> gpurun_out/pmc_icache.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_icache.txt
: > $OUT
for W in "$@"; do
  d=gpurun_out/icache_$W; rm -rf $d; mkdir -p $d
  SHORT="python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --arena-candidates 1"
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $d/a -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $d/b -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d $d/c -- $SHORT > /dev/null 2>&1
  python3 - "$d" "$W" <<'PY' >> gpurun_out/pmc_icache.txt
import csv, glob, sys
d, w = sys.argv[1:3]
acc = {}
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "probe" in r["Kernel_Name"] or "gen_" in r["Kernel_Name"] or "fill" in r["Kernel_Name"]: continue
        if not any(k in r["Kernel_Name"] for k in ("skin_kernel", "ggx_kernel", "sss_kernel", "disney_kernel", "integrate_kernel", "shade_kernel", "scatter_kernel", "direct_kernel")): continue
        per.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
        per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (c, _), v in per.items():
        acc.setdefault(c, []).append(v)
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(w, {k: f"{v:.4g}" for k, v in sorted(m.items())})
if m.get("SQC_ICACHE_REQ"):
    print(f"   icache miss rate {m.get('SQC_ICACHE_MISSES', 0) / m['SQC_ICACHE_REQ']:.4f}  hits/req {m.get('SQC_ICACHE_HITS', 0) / m['SQC_ICACHE_REQ']:.4f}")
if m.get("SQ_WAVE_CYCLES"):
    print(f"   wait_inst_any / wave_cycles {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.4f}   active_inst_any / wave_cycles {m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.4f}")
PY
done
cat $OUT
