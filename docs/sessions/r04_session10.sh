#!/bin/bash
# round 4, GPU session 10: the n^2-spp loops' rescaled rx and A = 2 rx / G1 - 1 through per-point reciprocals (RLS_LOOP_RECIP)
mkdir -p gpurun_out
OUT=gpurun_out/r04_loop_recip.txt; : > $OUT
for W in disney_integrate ggx_shade disney_shade skin_integrate; do echo "== $W" >> $OUT; bash tools/ab.sh $W nolooprecip >> $OUT 2>&1; done
cat $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/parity_soak.py --log2-points 24 --seeds 801,802 --out gpurun_out/r04_parity_soak_loop_recip.json > gpurun_out/soak_lr.log 2>&1; tail -3 gpurun_out/soak_lr.log
python tools/parity_soak.py --log2-points 22 --seeds 811 --groups 4,16 --spp-n 3 > gpurun_out/soak_lr2.log 2>&1; tail -3 gpurun_out/soak_lr2.log
