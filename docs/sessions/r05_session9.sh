#!/bin/bash
# round 5, GPU session 9: more of the lease's minutes on the parity soak of the final library (EXACT against the oracle, words that
# differ at all): 2^24 points x 60 seeds of the fourteen families.  (A first attempt with 120 seeds ran into gpurun's one-hour limit
# per call before the script wrote its result.)
mkdir -p gpurun_out
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_long.json --seeds $(seq -s, 8001 8060) > gpurun_out/r05_parity_soak_long.log 2>&1; tail -4 gpurun_out/r05_parity_soak_long.log
