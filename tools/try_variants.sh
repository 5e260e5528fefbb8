#!/bin/bash
# Bench the default library and experiment builds (rlshaders_amd/lib/librlshaders_amd_<variant>.so,
# made by `python -m rlshaders_amd.build --variant NAME ...`) back to back on one box.
# usage: tools/try_variants.sh [workload] variant...
W=${1:-ggx_reflect_refract}; shift
for v in "" "$@"; do
  lib=rlshaders_amd/lib/librlshaders_amd${v:+_$v}.so
  for m in exact fast; do
    RLSHADERS_AMD_LIB=$PWD/$lib python3 bench.py --workload $W --math $m --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-default}', '$m', d['roofline']['kernel_ms'], 'ms', d['value'], d['unit'])"
  done
done
