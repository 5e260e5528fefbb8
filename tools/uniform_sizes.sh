#!/bin/bash
# What hoisting the parameter-only arithmetic is worth at each batch size, and the launch geometry that gets it:
# uniform kernel (RLS_HOIST_TILES_PER_THREAD x RLS_HOIST_MIN_BLOCKS_PER_CU settings) against the same values as planes.
# usage: tools/uniform_sizes.sh > gpurun_out/uniform_sizes.txt
run() { python3 bench.py --workload $1 --log2-points $2 --steps 200 --warmup 20 --no-cpu-baseline --arena-candidates 1 2>/dev/null \
  | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'n=2^$2', '$3', d['roofline']['kernel_ms'], 'ms')"; }
for L in 16 18 20 22 24 26; do for w in disney_triple_glossy_uniform sss_probe_uniform skin_uniform ggx_reflect_refract_uniform; do
  RLS_BENCH_UNIFORM_AS_PLANES=1 run $w $L "as-planes"
  for cfg in "8 8" "8 4" "4 8" "16 4" "1 64"; do set -- $cfg
    RLS_HOIST_TILES_PER_THREAD=$1 RLS_HOIST_MIN_BLOCKS_PER_CU=$2 run $w $L "hoisted(tiles/thread=$1,min-blocks/CU=$2)"
  done
done; done
