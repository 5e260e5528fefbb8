"""Build driver: compiles the HIP translation units under ``csrc/`` for gfx950 and links them
into ``rlshaders_amd/lib/librlshaders_amd.so`` (in-tree, so the library travels with the repo
snapshot to the GPU box).  hipcc cross-compiles without a GPU present.

Parity build flags: ``-ffp-contract=off`` and no fast-math (FMA contraction alone moves ~5 % of
chained GGX values past 1e-5 relative).  fp32 divide / sqrt stay correctly
rounded (hipcc default ``-fhip-fp32-correctly-rounded-divide-sqrt``).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
OBJDIR = PKG / "build"
LIB = LIBDIR / "librlshaders_amd.so"
ARCH = "gfx950"

# the largest units first (the thread pool starts them first); integrate / lights / scatter / shade share csrc/rls_loops.hpp
SOURCES = ["shade.hip", "integrate.hip", "lights.hip", "scatter.hip", "skin.hip", "ggx.hip", "disney.hip", "sss.hip", "alternates.hip", "context.hip", "pipeline.hip",
           "libm_check.hip"]
FAST_UNITS = {"ggx.hip", "disney.hip", "sss.hip", "skin.hip", "integrate.hip", "lights.hip", "scatter.hip", "shade.hip", "alternates.hip"}
HEADERS = [CSRC / "rls_device.hpp", CSRC / "rls_libm.hpp", CSRC / "rls_libm_tables.inc", CSRC / "rls_libm_flavour_args.inc", CSRC / "rls_internal.hpp", CSRC / "rls_loops.hpp",
           PKG.parent / "include" / "rlshaders_amd.h", PKG.parent / "include" / "rlshaders_amd_diag.h"]

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math",
    # gfx950 issues v_pk_{mul,add}_f32 at half the rate of the scalar forms, and pairing operands costs
    # moves and registers: with SLP packing off config 2 runs 4 % (EXACT) / 6 % (FAST) faster; measured again in round 3 on the
    # n^2-spp loops (tools/ab.sh, packing ON against this default): rlDisney 64 spp +16 %, rlSkin shader_evaluate +25 %,
    # rlGgx shader_evaluate +7 %, its light loop +8 %, integrateScatter +2.5 % slower
    "-fno-slp-vectorize",
    "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the rlshaders_amd HIP library cannot be built")
    return exe


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def sources():
    return [CSRC / s for s in SOURCES if (CSRC / s).exists()]


def build_library(force: bool = False, verbose: bool = False, variant: str = "", defines=(), extra=()) -> Path:
    """Compile every HIP TU for gfx950 and link the C-ABI shared library.  Returns its path.

    ``variant`` / ``defines`` build an experiment flavour next to the product library
    (``librlshaders_amd_<variant>.so``, selected at run time with RLSHADERS_AMD_LIB)."""
    hipcc = _hipcc()
    LIBDIR.mkdir(exist_ok=True)
    objdir = OBJDIR if not variant else PKG / f"build_{variant}"
    lib = LIB if not variant else LIBDIR / f"librlshaders_amd_{variant}.so"
    objdir.mkdir(exist_ok=True)
    srcs = sources()
    jobs = []
    objs = []
    for src in srcs:
        # every closure unit twice: EXACT (carries the C ABI) and FAST (kernels behind a hidden symbol)
        flavours = [("", "RLS_FAST=0")] + ([("_fast", "RLS_FAST=1")] if src.name in FAST_UNITS else [])
        for suffix, flag in flavours:
            obj = objdir / (src.stem + suffix + ".o")
            objs.append(obj)
            if force or _stale(obj, [src, *HEADERS, Path(__file__)]):
                # -cuid: clang names one symbol per unit `__hip_cuid_<hash>` and by default hashes the PATHS on its command line
                # into it -- the same sources in another checkout or object directory then differ in that symbol's length now
                # and again, which can move a code object's layout by a cache line (rlshaders_amd/codeid.py).  A fixed id per
                # unit makes the device code a function of sources and flags alone.
                jobs.append([hipcc, *HIPCC_FLAGS, *extra, f"-cuid=rlshaders_amd.{src.stem}{suffix}", f"-D{flag}",
                             *[f"-D{d}" for d in defines], "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{p.stdout}\n{p.stderr}")
        return p

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)])
    return lib


def build_host_examples(verbose: bool = False) -> Path:
    """Compile-check the C++ host mirror (header-only) and its example against the C ABI."""
    OBJDIR.mkdir(exist_ok=True)
    first = OBJDIR / "example_arnold_stub"
    for name in ("example_arnold_stub", "example_multi_gpu", "example_host_pipeline", "test_arnold_stub"):
        src = PKG / "host" / f"{name}.cpp"
        out = OBJDIR / name
        if not src.exists():
            continue
        if _stale(out, [src, PKG / "host" / "rls_batch.hpp", PKG / "host" / "rl_arnold_stub.hpp", LIB, *HEADERS]):
            cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-pthread", f"-I{PKG.parent / 'include'}", f"-I{PKG / 'host'}",
                   str(src), "-o", str(out), f"-L{LIBDIR}", "-lrlshaders_amd", f"-Wl,-rpath,{LIBDIR}",
                   "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            p = subprocess.run(cmd, capture_output=True, text=True)
            if p.returncode != 0:
                raise RuntimeError(f"host example {name} failed to build:\n{p.stdout}\n{p.stderr}")
    return first


if __name__ == "__main__":
    # python -m rlshaders_amd.build [--force] [--variant NAME -DFOO=1 -fsome-flag -mllvm -some-option ...]
    args = sys.argv[1:]
    variant = args[args.index("--variant") + 1] if "--variant" in args else ""
    defines = [a[2:] for a in args if a.startswith("-D")]
    extra = []
    for i, a in enumerate(args):
        if a.startswith(("-f", "-m", "-O")):
            extra.append(a)
            if a == "-mllvm":
                extra.append(args[i + 1])
    print(build_library(force="--force" in args, verbose=True, variant=variant, defines=defines, extra=extra))
