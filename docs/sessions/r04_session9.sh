#!/bin/bash
# round 4, GPU session 9: D_GTR2Aniso's quotients through per-point reciprocals (RLS_DISNEY_D_RECIP) -- A/B and parity
mkdir -p gpurun_out
OUT=gpurun_out/r04_disney_d_recip.txt; : > $OUT
for W in disney_integrate disney_triple_glossy disney_triple_glossy_uniform; do echo "== $W" >> $OUT; bash tools/ab.sh $W dnorecip >> $OUT 2>&1; done
cat $OUT
python -m pytest tests -m gpu -x -q -k "disney or shade or config3 or parity_sweep or hostile or by_reference or uniform" 2>&1 | tail -4
python tools/parity_soak.py --log2-points 24 --seeds 601,602 --out gpurun_out/r04_parity_soak_d_recip.json > gpurun_out/soak_dr.log 2>&1; tail -4 gpurun_out/soak_dr.log
