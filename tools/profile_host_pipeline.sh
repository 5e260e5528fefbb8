#!/bin/bash
# Kernel and copy time of one pass of a host-resident workload against its wall time: how much of the copies the pipeline hides.
# usage: tools/profile_host_pipeline.sh <workload> [log2 points]  -> gpurun_out/host_pipeline_<workload>.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
W=$1; L=${2:-24}
d=gpurun_out/hp_$W; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $d -- python3 bench.py --workload $W --log2-points $L --steps 6 --warmup 2 --no-cpu-baseline > $d/bench.json 2> /dev/null
python3 - "$d" "$W" <<'PY' | tee gpurun_out/host_pipeline_$W.txt
import csv, glob, json, sys
d, w = sys.argv[1:3]
b = json.loads(open(f"{d}/bench.json").read().strip().splitlines()[-1])
passes = b["steps"] + b["warmup"]
print(f"{w}: {b['value']} {b['unit']}, {b['ms_per_step']} ms per pass of 2^24 points (under rocprofv3 --kernel-trace --memory-copy-trace)")


def union(iv):
    iv = sorted(iv); tot = 0; cur_s, cur_e = None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot


kern, cop = [], {}
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("shade_kernel", "integrate_kernel", "ggx_kernel")):
            kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0, t1 = min(k[0] for k in kern), max(k[1] for k in kern)
for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e < t0 - 5e7 or s > t1 + 5e7 or e - s < 50_000: continue           # the chunk copies of the passes (>= 50 us)
        cop.setdefault((r["Direction"], r["Source_Agent_Id"], r["Destination_Agent_Id"]), []).append((s, e))
ms = lambda ns: ns / 1e6 / passes
print(f"  per pass: {len(kern) // passes} chunk kernels, sum of their durations {ms(sum(e - s for s, e in kern)):.2f} ms, "
      f"time with at least one running {ms(union(kern)):.2f} ms")
allc = []
for k, v in sorted(cop.items()):
    allc += v
    print(f"  per pass: {len(v) // passes} copies {k[0]} {k[1]} -> {k[2]}: sum {ms(sum(e - s for s, e in v)):.2f} ms, "
          f"time with at least one running {ms(union(v)):.2f} ms")
both = union(kern + allc)
print(f"  per pass: time with a kernel OR a copy running {ms(both):.2f} ms; kernels and copies overlapped for "
      f"{ms(union(kern) + union(allc) - both):.2f} ms")
PY
