#!/bin/bash
# round 5, GPU session 5: long parity soaks of the round's final library (EXACT against the oracle, counted in output words that
# differ at all): every closure family on 2^24 points x 40 seeds, the lane-group widths of the n^2-spp loops, uniform parameter
# sets through the hoisting kernels, parameters by reference against planes
mkdir -p gpurun_out
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_big.json --seeds $(seq -s, 7001 7040) > gpurun_out/r05_parity_soak_big.log 2>&1; tail -4 gpurun_out/r05_parity_soak_big.log
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_big_groups.json --seeds 7101,7102,7103,7104 --groups 1,4,16,64 --spp-n 3 --log2-points 22 > gpurun_out/r05_parity_soak_big_groups.log 2>&1; tail -4 gpurun_out/r05_parity_soak_big_groups.log
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_big_uniform.json --uniform-draws 40 --seeds 7201,7202,7203,7204 > gpurun_out/r05_parity_soak_big_uniform.log 2>&1; tail -4 gpurun_out/r05_parity_soak_big_uniform.log
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_big_by_reference.json --by-reference 1,37,4096,65536 --seeds 7301,7302 > gpurun_out/r05_parity_soak_big_by_reference.log 2>&1; tail -4 gpurun_out/r05_parity_soak_big_by_reference.log
