#!/bin/bash
# round 5, GPU session 10: every workload of bench.py on the round's final library (--workloads all: the verbs a stub calls one
# by one, the uniform / by-reference variants, the whole nodes, the host-resident pipelines), for the record
mkdir -p gpurun_out
( time python bench.py --workloads all ) > gpurun_out/r05_bench_all.out 2> gpurun_out/r05_bench_all.err; tail -3 gpurun_out/r05_bench_all.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r05_bench_all.out'):
    if not l.startswith('{'): continue
    r = json.loads(l)
    if r.get('record') == 'workload':
        rf = r.get('roofline') or {}
        print(f"{r['name']:40s} 2^{r['points_per_gpu'].bit_length()-1:<3d} {rf.get('kernel_ms')!s:>10} ms  {r.get('value')!s:>9} {r.get('unit','')}  frac {rf.get('frac')}  clock {rf.get('effective_clock_ghz')}" if 'error' not in r else f"{r['name']} ERROR {r['error']}")
PY
