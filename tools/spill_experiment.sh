#!/bin/bash
# where does the extra HBM traffic of the whole-node kernels come from?  ggx_shade_kernel spills 20 VGPRs
# (84 B of scratch per lane) at its four waves per SIMD, disney_shade_kernel 20 at its three.  Variant libraries with one
# wave fewer (librlshaders_amd_w3.so: -DRLS_INT_WAVES=3, librlshaders_amd_dw2.so: -DRLS_DISNEY_LIGHT_WAVES=2) have no
# vector spills: time (tools/ab.sh) and HBM traffic (FETCH_SIZE / WRITE_SIZE) of both, on one box.
# usage: tools/spill_experiment.sh   -> gpurun_out/spill_experiment.txt
OUT=gpurun_out/spill_experiment.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
traffic() {   # <workload> <variant or ""> 
  lib=$PWD/rlshaders_amd/lib/librlshaders_amd${2:+_$2}.so
  d=gpurun_out/spill_$1_${2:-default}; rm -rf $d; mkdir -p $d
  SHORT="python3 bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline --arena-candidates 1"
  RLSHADERS_AMD_LIB=$lib rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- $SHORT > /dev/null 2>&1
  RLSHADERS_AMD_LIB=$lib rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -- $SHORT > /dev/null 2>&1
  python3 - "$d" "$1" "${2:-default}" <<'PY' >> gpurun_out/spill_experiment.txt
import csv, glob, sys
d, w, v = sys.argv[1:4]
def pmc(sub, name):
    acc = {}
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "shade_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                acc.setdefault(r["Dispatch_Id"], 0.0)
                acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return sum(acc.values()) / max(len(acc), 1)
rd, wr = pmc("fetch", "FETCH_SIZE") * 1024 * 2.0, pmc("write", "WRITE_SIZE") * 1024     # FETCH_SIZE x 2: gfx950 streaming reads (MI355X_MICROARCH.md)
print(f"traffic {w} {v}: read {rd / 1e9:.3f} GB + write {wr / 1e9:.3f} GB = {(rd + wr) / 1e9:.3f} GB per launch")
PY
}
bash tools/ab.sh ggx_shade w3 >> $OUT 2>&1
traffic ggx_shade ""; traffic ggx_shade w3
bash tools/ab.sh disney_shade dw2 >> $OUT 2>&1
traffic disney_shade ""; traffic disney_shade dw2
cat $OUT
