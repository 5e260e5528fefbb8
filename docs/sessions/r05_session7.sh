#!/bin/bash
# round 5, GPU session 7: the cache policy of the plane accesses under the power cap (RLS_MEM_POLICY: 0 non-temporal loads and
# stores = the product, 1 plain loads, 2 plain stores, 3 both plain), A/B on one box, two interleaved repetitions, three kernels
mkdir -p gpurun_out
{ for w in ggx_reflect_refract sss_probe skin; do echo "== $w"; bash tools/ab.sh $w pol1 pol2 pol3; done; } > gpurun_out/r05_mem_policy.txt 2>&1; cat gpurun_out/r05_mem_policy.txt
