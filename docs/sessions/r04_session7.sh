#!/bin/bash
# round 4, GPU session 7: parity soak on the round's library (fourteen closure families, 2^25 points x 4 seeds; lane groups)
mkdir -p gpurun_out
python tools/parity_soak.py --log2-points 25 --seeds 401,402,403,404 --out gpurun_out/r04_parity_soak.json > gpurun_out/r04_parity_soak.log 2>&1; tail -4 gpurun_out/r04_parity_soak.log
python tools/parity_soak.py --log2-points 23 --seeds 411,412 --groups 4,16 --spp-n 3 --out gpurun_out/r04_parity_soak_lane_groups.json > gpurun_out/r04_parity_soak_lg.log 2>&1; tail -4 gpurun_out/r04_parity_soak_lg.log
python tools/parity_soak.py --log2-points 23 --seeds 421,422 --by-reference 1,37,4096 --out gpurun_out/r04_parity_soak_by_reference.json > gpurun_out/r04_parity_soak_br.log 2>&1; tail -4 gpurun_out/r04_parity_soak_br.log
