#!/bin/bash
# A/B experiment builds on one box: EXACT kernel time of each variant library, twice (interleaved, so that clock / placement
# drift shows), then optionally the parity gate of each.   usage: tools/ab.sh <workload> [--parity] variant...
# ("" = the product library; variants are rlshaders_amd/lib/librlshaders_amd_<variant>.so from `python -m rlshaders_amd.build --variant`)
W=$1; shift
PAR=0; if [ "$1" = "--parity" ]; then PAR=1; shift; fi
run() {
  lib=rlshaders_amd/lib/librlshaders_amd${1:+_$1}.so
  RLSHADERS_AMD_LIB=$PWD/$lib python3 bench.py --workload $W --math exact --steps 40 --warmup 10 --no-cpu-baseline --arena-candidates 4 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${1:-default}', d['roofline']['kernel_ms'], 'ms', d['value'], d['unit'], 'frac', d['roofline'].get('frac'))"
}
for rep in 1 2; do for v in "" "$@"; do run "$v"; done; done
if [ $PAR = 1 ]; then
  for v in "$@"; do
    lib=rlshaders_amd/lib/librlshaders_amd_$v.so
    echo "== parity $v"; RLSHADERS_AMD_LIB=$PWD/$lib python3 -m pytest tests -m gpu -x -q -k "not bench_multirank and not host_cpp" 2>&1 | tail -3
  done
fi
