"""CPU: the generated Arnold-side stub (rlshaders_amd/host/rl_arnold_stub.hpp) declares what the reference's plugin
declares -- node names and node_loader entries (src/_PluginMain.cpp:16-46), parameters in declaration order with their
types, defaults and in-code metadata (node_parameters of src/rlGgx.cpp:170-198, src/rlDisney.cpp:604-638,
src/rlSkin.cpp:107-139), and the positional p_* enumerators -- checked by replaying the stub's node_parameters /
node_loader through a recording host and comparing with the records extracted from the reference (tests/golden)."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
FIX = json.loads((ROOT / "tests" / "golden" / "param_surface.json").read_text())


def _driver() -> Path:
    sys.path.insert(0, str(ROOT))
    from rlshaders_amd import build as b
    b.build_library()
    b.build_host_examples()
    return b.OBJDIR / "test_arnold_stub"


def test_generated_header_is_current():
    subprocess.run([sys.executable, str(ROOT / "tools" / "gen_arnold_stub.py"), "--check"], check=True)


def test_stub_declares_the_reference_nodes():
    out = subprocess.run([str(_driver()), "decl"], capture_output=True, text=True, check=True).stdout
    got = json.loads(out)
    assert list(got["nodes"]) == ["rlGgx", "rlDisney", "rlSkin"]
    for node, spec in FIX["nodes"].items():
        mine = got["nodes"][node]
        assert [p["name"] for p in mine["parameters"]] == [p["name"] for p in spec["parameters"]], node
        assert mine["enum"] == spec["enum"], node
        assert mine["maya.id"] == spec["mtd"]["maya.id"]
        for a, b in zip(mine["parameters"], spec["parameters"]):
            assert a["type"] == b["type"], (node, a["name"])
            # the reference's defaults are fp32 literals (0.6f, 1.44f, 0.35f): compare as fp32
            f32 = lambda v: [np.float32(x) if isinstance(x, float) else x for x in v]
            assert f32(a["default"]) == f32(b["default"]), (node, a["name"], a["default"], b["default"])
            assert a["meta"] == b["meta"], (node, a["name"], a["meta"], b["meta"])
    assert len(got["node_loader"]) == len(FIX["node_loader"]) == 3
    for a, b in zip(got["node_loader"], FIX["node_loader"]):
        for k in ("id", "methods", "output_type", "name", "node_type"):
            assert a[k] == b[k], (k, a, b)
        assert a["version"] == "4.2.11.0"


def test_arnold_glue_is_well_formed():
    """the RLS_STUB_WITH_ARNOLD block (ArnoldHost / ArnoldEval / the node_parameters and node_loader bodies) compiled
    with -fsyntax-only against mock declarations of the SDK symbols it touches (tests/native/mock_arnold/ai.h; Arnold's
    node_parameters / node_loader are MACROS, so nothing of the stub may carry those names)"""
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Werror", "-I", str(ROOT / "tests" / "native" / "mock_arnold"),
           "-I", str(ROOT / "include"), "-I", str(ROOT / "rlshaders_amd" / "host"),
           str(ROOT / "tests" / "native" / "arnold_glue_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def _rows(tmp_path, cols):
    cols = np.ascontiguousarray(cols, np.float32)
    K, n = cols.shape
    inp, outp = tmp_path / "cols.bin", tmp_path / "rows.bin"
    with open(inp, "wb") as f:
        np.array([n, K], np.float32).tofile(f)
        cols.tofile(f)
    p = subprocess.run([str(_driver()), "rows", str(inp), str(outp)], capture_output=True, text=True, check=True)
    info = json.loads(p.stdout.strip().splitlines()[-1])
    raw = np.fromfile(outp, np.uint32)
    if not info["by_reference"]:
        return info, None, None
    ids = raw[:n]
    table = raw[n:].view(np.float32).reshape(K, info["count"])
    return info, ids, table


def test_stub_finds_the_distinct_parameter_rows(tmp_path):
    """detail::Materials::find_rows (host only, no device): a batch of the hits of a few node instances goes by reference -- the
    table holds each distinct row of parameter BITS once, in order of first appearance, and table[:, id] rebuilds every point's
    row; one row (the uniform case), many rows (textures) and tiny batches do not."""
    rng = np.random.default_rng(0)
    n, K, m = 5000, 6, 23
    inst = rng.normal(size=(K, m)).astype(np.float32)
    inst[0, 3] = np.float32(-0.0); inst[0, 4] = np.float32(0.0)            # +-0 are different bits: different rows
    inst[1:, 4] = inst[1:, 3]
    inst[2, 7] = np.float32("nan"); inst[2, 8] = np.float32("nan")          # NaNs with equal bits: one row if the rest agrees
    inst[:2, 8] = inst[:2, 7]; inst[3:, 8] = inst[3:, 7]
    which = np.repeat(rng.integers(0, m, n // 10), 10)                      # runs of consecutive hits on one instance
    cols = inst[:, which]
    info, ids, table = _rows(tmp_path, cols)
    assert info["by_reference"] is True
    rows_bits = {cols[:, i].view(np.uint32).tobytes() for i in range(n)}
    assert info["count"] == len(rows_bits) <= m
    assert np.array_equal(table[:, ids].view(np.uint32), cols.view(np.uint32))
    first = [np.flatnonzero(ids == k)[0] for k in range(info["count"])]
    assert first == sorted(first)                                           # ids in order of first appearance
    # one row: uniform; every point different: textures; too few points: nothing to gain
    assert _rows(tmp_path, np.repeat(inst[:, :1], n, axis=1))[0] == {"by_reference": False, "count": 1}
    assert _rows(tmp_path, rng.normal(size=(K, n)).astype(np.float32))[0]["by_reference"] is False
    assert _rows(tmp_path, cols[:, :12])[0]["by_reference"] is False
    # more than one row per eight points: planes
    many = inst[:, rng.integers(0, m, 100)]
    assert _rows(tmp_path, many)[0]["by_reference"] is False
