#!/bin/bash
# round 4, GPU session 8: the SSE2 flavour of the device libm (-DRLM_GLIBC_FMA=0) against a glibc told to run its SSE2 build
# (GLIBC_TUNABLES=glibc.cpu.hwcaps=-AVX2,-FMA): the library can follow either glibc build bit for bit
mkdir -p gpurun_out
export RLSHADERS_AMD_LIB=$PWD/rlshaders_amd/lib/librlshaders_amd_sse2.so
export GLIBC_TUNABLES=glibc.cpu.hwcaps=-AVX2,-FMA
python -c "import rlshaders_amd as R; print('library follows', R.libm_flavour(), '; host mismatches', R.host_libm_mismatches())"
python tools/parity_soak.py --log2-points 23 --seeds 501,502 --out gpurun_out/r04_parity_soak_sse2.json > gpurun_out/r04_parity_soak_sse2.log 2>&1; tail -4 gpurun_out/r04_parity_soak_sse2.log
python tools/libm_exhaustive.py --out gpurun_out/r04_libm_exhaustive_sse2.json > gpurun_out/r04_libm_exhaustive_sse2.log 2>&1; tail -25 gpurun_out/r04_libm_exhaustive_sse2.log
