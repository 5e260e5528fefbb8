"""CPU: the top-level CMakeLists.txt (the reference is a CMake project: /root/reference/CMakeLists.txt:58,
src/CMakeLists.txt) builds librlshaders_amd.so for gfx950 in the GPU-less container with the flags of
rlshaders_amd/build.py, exports exactly the symbols include/rlshaders_amd.h declares (the checks of
tests/test_capi_symbols.py, run against this library), installs header + exported target, and an outside project finds
the installed package and links host/example_arnold_stub.cpp against it.  No compute call (no GPU here)."""
import os
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "rlshaders_amd" / "build" / "cmake"          # under the git-ignored build directory
PREFIX = BUILD / "prefix"


def _run(cmd, **kw):
    p = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert p.returncode == 0, f"{' '.join(map(str, cmd))}\n{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    return p.stdout


@pytest.fixture(scope="module")
def built():
    if shutil.which("cmake") is None:
        pytest.skip("cmake is not installed")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    _run(["cmake", "-S", str(ROOT), "-B", str(BUILD), *gen])
    _run(["cmake", "--build", str(BUILD), "-j", str(min(6, os.cpu_count() or 2))])      # incremental after the first run
    shutil.rmtree(PREFIX, ignore_errors=True)
    _run(["cmake", "--install", str(BUILD), "--prefix", str(PREFIX)])
    return BUILD


def _exported(lib: Path):
    out = _run(["nm", "-D", "--defined-only", str(lib)])
    return sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("rls_"))


def test_same_exported_symbols_as_the_header_and_the_python_build(built):
    from test_capi_symbols import declared_symbols
    declared = declared_symbols()
    got = _exported(built / "librlshaders_amd.so")
    assert got == declared, sorted(set(declared) ^ set(got))
    from rlshaders_amd import build as b
    assert got == _exported(b.build_library())


def test_same_device_code_as_the_python_build(built):
    """The library a plugin maintainer links is the CMake one; the one the GPU tests, the soaks and bench.py run is
    rlshaders_amd/build.py's.  CMAKE_BUILD_TYPE=Release adds -DNDEBUG and spells / orders the flags its own way, so equality
    of the flag lists proves nothing: compare what the GPU executes.  Every gfx950 code object in the two libraries carries the
    same instructions, kernel descriptors and data (rlshaders_amd/codeid.py: sha-256 per code object over .text / .rodata /
    .data, then over the sorted set), so the parity evidence gathered on one build holds for the other bit for bit."""
    from rlshaders_amd import build as b
    from rlshaders_amd.codeid import DeviceCode
    cm, py = DeviceCode(built / "librlshaders_amd.so"), DeviceCode(b.build_library())
    assert len(cm.units) == len(py.units) >= 12
    assert sorted(u for u, _ in cm.units) == sorted(u for u, _ in py.units)
    assert cm.library_id == py.library_id
    for k in ("ggx_kernel<5, 0, 1>", "sss_kernel<3, 0, 0>", "skin_kernel<0, 1>", "disney_integrate_kernel<1, 0>"):
        assert cm.unit_of_kernel(k) is not None and cm.unit_of_kernel(k) == py.unit_of_kernel(k), k


def test_flags_are_the_parity_flags(built):
    from rlshaders_amd import build as b
    ninja = (built / "build.ninja")
    text = ninja.read_text() if ninja.exists() else "\n".join(p.read_text() for p in built.rglob("flags.make"))
    hip_lines = [l for l in text.splitlines() if "--offload-arch=gfx950" in l and l.strip().startswith(("FLAGS =", "HIP_FLAGS ="))]
    assert hip_lines
    for flag in b.HIPCC_FLAGS:
        if flag.startswith("--offload-arch") or flag == "-fPIC":       # (CMake spells these itself)
            continue
        assert all(flag in l for l in hip_lines), flag
    assert "-ffast-math" not in text.replace("-fno-fast-math", "")
    defs = [l for l in text.splitlines() if "DEFINES" in l and "RLS_FAST" in l]
    assert any("RLS_FAST=0" in l for l in defs) and any("RLS_FAST=1" in l for l in defs)


def test_install_tree_and_outside_consumer(built, tmp_path):
    for rel in ("include/rlshaders_amd.h", "include/rlshaders_amd_diag.h", "include/rls_batch.hpp", "include/rl_arnold_stub.hpp", "lib/librlshaders_amd.so",
                "lib/cmake/rlshaders_amd/rlshaders_amdConfig.cmake", "lib/cmake/rlshaders_amd/rlshaders_amdTargets.cmake",
                "share/doc/rlshaders_amd/THIRD_PARTY.md"):
        assert (PREFIX / rel).exists(), rel
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    src = ROOT / "tests" / "native" / "cmake_consumer"
    _run(["cmake", "-S", str(src), "-B", str(tmp_path / "b"), *gen, f"-DCMAKE_PREFIX_PATH={PREFIX}",
          f"-DRLS_EXAMPLE_SOURCE={ROOT / 'rlshaders_amd' / 'host' / 'example_arnold_stub.cpp'}"])
    _run(["cmake", "--build", str(tmp_path / "b")])
    exe = tmp_path / "b" / "consumer"
    assert exe.exists()
    needed = _run(["readelf", "-d", str(exe)])
    assert "librlshaders_amd.so" in needed
