#!/bin/bash
# round 4, third GPU session: FAST under protocol (3) with the corner sampling; the wake phase A/B on the driver's command
mkdir -p gpurun_out
python tools/fast_conditioning.py --log2-points 16 --out gpurun_out/r04_fast_conditioning_small.json > gpurun_out/r04_fast_small.log 2>&1; tail -2 gpurun_out/r04_fast_small.log
python tools/fast_conditioning.py --log2-points 24 > gpurun_out/r04_fast_conditioning.log 2>&1; tail -2 gpurun_out/r04_fast_conditioning.log
for rep in 1 2; do
  for wake in 0 60; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --wake-ms $wake --no-cpu-baseline --workloads none | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wake $wake:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04_bench_driver_cmd.out 2> gpurun_out/r04_bench_driver_cmd.err; tail -1 gpurun_out/r04_bench_driver_cmd.out | wc -c; tail -3 gpurun_out/r04_bench_driver_cmd.err
