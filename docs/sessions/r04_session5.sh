#!/bin/bash
# round 4, GPU session 5: FAST with refined arithmetic -- kernel time (config 2, 4, 5 kernels) and conditioning
mkdir -p gpurun_out
OUT=gpurun_out/r04_fast_refined.txt; : > $OUT
run() { # lib-suffix workload
  lib=rlshaders_amd/lib/librlshaders_amd${1:+_$1}.so
  RLSHADERS_AMD_LIB=$PWD/$lib python3 bench.py --workload $2 --math fast --steps 40 --warmup 10 --no-cpu-baseline --arena-candidates 4 2>/dev/null \
    | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', '${1:-product}', d['roofline']['kernel_ms'], 'ms', d['value'], 'Gsamples/s frac', d['roofline'].get('frac'))"
}
for W in ggx_reflect_refract sss_probe skin disney_triple_glossy; do
  for rep in 1 2; do for v in "" fastold fastz; do run "$v" $W >> $OUT; done; done
done
cat $OUT
python tools/fast_conditioning.py --log2-points 24 --out gpurun_out/r04_fast_conditioning_refined.json > gpurun_out/r04_fast_conditioning_refined.log 2>&1; grep "^fast\|wrote" gpurun_out/r04_fast_conditioning_refined.log | cut -c1-700
RLSHADERS_AMD_LIB=$PWD/rlshaders_amd/lib/librlshaders_amd_fastz.so python tools/fast_conditioning.py --log2-points 24 --out gpurun_out/r04_fast_conditioning_fastz.json > gpurun_out/r04_fast_conditioning_fastz.log 2>&1; grep "^fast\|wrote" gpurun_out/r04_fast_conditioning_fastz.log | cut -c1-700
python -m pytest tests/test_gpu_fast_mode.py -x -q 2>&1 | tail -5
