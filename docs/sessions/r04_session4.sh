#!/bin/bash
mkdir -p gpurun_out
python tools/fast_conditioning.py --log2-points 24 > gpurun_out/r04_fast_conditioning.log 2>&1; tail -14 gpurun_out/r04_fast_conditioning.log
