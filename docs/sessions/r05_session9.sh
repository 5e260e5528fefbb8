#!/bin/bash
# round 5, GPU session 9: the lease's remaining minutes on parity soaks of the final library (EXACT against the oracle, words that
# differ at all): 2^24 points x 120 seeds of the fourteen families, 320 more uniform parameter sets
mkdir -p gpurun_out
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_long.json --seeds $(seq -s, 8001 8120) > gpurun_out/r05_parity_soak_long.log 2>&1; tail -4 gpurun_out/r05_parity_soak_long.log
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_long_uniform.json --uniform-draws 40 --seeds $(seq -s, 8201 8208) > gpurun_out/r05_parity_soak_long_uniform.log 2>&1; tail -4 gpurun_out/r05_parity_soak_long_uniform.log
