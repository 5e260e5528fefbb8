"""CPU: the C-ABI library builds, loads, exports every symbol include/rlshaders_amd.h declares, and
fails loudly (status + message, no fallback) when there is no GPU."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


HEADERS = ("rlshaders_amd.h", "rlshaders_amd_diag.h")     # the drop-in boundary; the measurement aids (same library)


def declared_symbols(headers=HEADERS):
    names = set()
    for h in headers:
        text = (ROOT / "include" / h).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(rls_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_diagnostics_are_not_in_the_drop_in_header():
    """what a plugin binds (rlshaders_amd.h) declares no measurement aid; rlshaders_amd_diag.h declares
    nothing else"""
    drop_in, diag = declared_symbols(HEADERS[:1]), declared_symbols(HEADERS[1:])
    assert not [s for s in drop_in if s.startswith("rls_diag_")]
    assert diag and all(s.startswith("rls_diag_") for s in diag), diag
    assert "rlshaders_amd_diag.h" not in re.sub(r"/\*.*?\*/", "", (ROOT / "include" / HEADERS[0]).read_text(), flags=re.S)


def test_header_symbols_exported_and_bound():
    from rlshaders_amd import _capi as capi, build
    build.build_library()
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 40
    for nm in names:
        assert hasattr(lib, nm), f"{nm} declared in the header but not exported"
        assert nm in capi.PROTOTYPES, f"{nm} has no ctypes prototype"
    extra = set(capi.PROTOTYPES) - set(names)
    assert not extra, f"prototypes without a header declaration: {extra}"


def test_header_compiles_as_c_and_cxx(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "rlshaders_amd.h"\nint main(void){ rls_ggx_closure c; (void)c; return rls_version() ? 0 : 0; }\n')
    diag = tmp_path / "d.c"
    diag.write_text('#include "rlshaders_amd_diag.h"\nint main(void){ return rls_diag_clock_stamps_end(0) ? 0 : 0; }\n')
    for unit in (src, diag):
        for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++14")):
            p = subprocess.run([cc, std, "-Wall", "-Werror", "-pedantic", "-fsyntax-only", f"-I{ROOT / 'include'}",
                                "-x", "c" if cc == "gcc" else "c++", str(unit)], capture_output=True, text=True)
            assert p.returncode == 0, p.stderr


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import rlshaders_amd as R
    lib = R.load()
    h = C.c_void_p()
    st = lib.rls_context_create(0, C.byref(h))
    assert st == 2 and not h.value
    assert b"no HIP device" in lib.rls_last_error()
    with pytest.raises(RuntimeError):
        R.Context(0)


def test_struct_layouts_match_header():
    """sizeof of every ABI struct as the C compiler sees it == the ctypes mirror"""
    import subprocess
    import tempfile
    from rlshaders_amd import _capi as capi
    pairs = [("rls_material_index", capi.MaterialIndex), ("rls_cvec3", capi.CVec3), ("rls_vec3", capi.Vec3), ("rls_rgb", capi.Rgb), ("rls_param", capi.Param),
             ("rls_param_rgb", capi.ParamRgb), ("rls_ggx_closure", capi.GgxClosure),
             ("rls_disney_closure", capi.DisneyClosure), ("rls_disney_stream_out", capi.DisneyStreamOut),
             ("rls_sss_closure", capi.SssClosure), ("rls_skin_closure", capi.SkinClosure),
             ("rls_skin_out", capi.SkinOut)]
    body = "".join(f'printf("%zu\\n", sizeof({c}));' for c, _ in pairs)
    with tempfile.TemporaryDirectory() as d:
        src = Path(d) / "s.c"
        src.write_text(f'#include <stdio.h>\n#include "rlshaders_amd.h"\nint main(void){{{body} return 0;}}\n')
        exe = Path(d) / "s"
        subprocess.run(["gcc", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
        sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    for (cname, ct), sz in zip(pairs, sizes):
        assert C.sizeof(ct) == sz, (cname, C.sizeof(ct), sz)


def test_product_never_touches_oracle():
    """the product path must not import, link or call anything under oracle/"""
    import subprocess
    pkg = ROOT / "rlshaders_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.cpp")):
        t = p.read_text()
        assert "oracle_lib" not in t and "librls_oracle" not in t and "rls_oracle.h" not in t, p
    out = subprocess.run(["ldd", str(pkg / "lib" / "librlshaders_amd.so")], capture_output=True, text=True).stdout
    assert "rls_oracle" not in out
    # bench: the workload table (what the GPU legs launch) never names the oracle; the driver imports it in the two functions
    # of its cpu_baseline leg and nowhere else
    import ast
    t = (ROOT / "bench_workloads.py").read_text()
    assert "oracle_lib" not in t and "import cases" not in t and "librls_oracle" not in t
    tree = ast.parse((ROOT / "bench.py").read_text())
    where = set()
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
        for node in (fn.body if isinstance(fn, ast.Module) else ast.walk(fn)):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                names = [a.name for a in node.names] + ([node.module] if isinstance(node, ast.ImportFrom) else [])
                if any(n in ("oracle_lib", "cases") for n in names):
                    where.add(fn.name if isinstance(fn, ast.FunctionDef) else "<module>")
    assert where == {"_cpu_leg", "cpu_baseline"}, where


def test_shard_range_matches_the_python_sharding():
    """rls_shard_range (what a C++ host shards with) == rlshaders_amd.sharding.shard_range (what bench.py shards
    with); touches no device."""
    import ctypes as C
    import random
    from rlshaders_amd import _capi
    from rlshaders_amd.sharding import shard_range
    lib = _capi.load()
    f, c = C.c_int64(), C.c_int64()
    rng = random.Random(1)
    for _ in range(5000):
        total = rng.choice([0, 1, 7, 2 ** 26, 2 ** 30, 10 ** 9, rng.randrange(0, 2 ** 40), 2 ** 62 + 12345])
        world = rng.choice([1, 2, 3, 4, 7, 8, 64])
        rank = rng.randrange(world)
        assert lib.rls_shard_range(total, rank, world, C.byref(f), C.byref(c)) == 0
        assert (f.value, c.value) == shard_range(total, rank, world), (total, rank, world)
    covered = 0
    for rank in range(8):
        assert lib.rls_shard_range(10 ** 9, rank, 8, C.byref(f), C.byref(c)) == 0
        assert f.value == covered
        covered += c.value
    assert covered == 10 ** 9
    assert lib.rls_shard_range(10, 3, 3, C.byref(f), C.byref(c)) == 1 and lib.rls_shard_range(-1, 0, 1, C.byref(f), C.byref(c)) == 1
    assert lib.rls_device_count() >= 0
