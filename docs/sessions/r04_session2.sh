#!/bin/bash
# round 4, second GPU session: FAST under protocol (3), rlGgx reload / occupancy A/B, SQ busy / wait counters
mkdir -p gpurun_out
python -m pytest tests/test_libm_flavour_api.py -q 2>&1 | tail -2
python tools/fast_conditioning.py --log2-points 16 --out gpurun_out/r04_fast_conditioning_small.json > gpurun_out/r04_fast_small.log 2>&1; tail -3 gpurun_out/r04_fast_small.log
python tools/fast_conditioning.py --log2-points 24 > gpurun_out/r04_fast_conditioning.log 2>&1; tail -3 gpurun_out/r04_fast_conditioning.log
bash tools/ab.sh ggx_reflect_refract ggxnoreload ggxw4 ggxw5 ggxw6 > gpurun_out/r04_ggx_ab.txt 2>&1; cat gpurun_out/r04_ggx_ab.txt
bash tools/pmc_stalls.sh sss_probe nd_sample skin ggx_reflect_refract > /dev/null 2>&1; cp gpurun_out/pmc_stalls.txt gpurun_out/r04_pmc_stalls.txt; cat gpurun_out/r04_pmc_stalls.txt
