"""CPU: the closure parameter surface is the reference's -- same names, types and defaults
(tests/golden/param_surface.json, extracted from the reference's node_parameters blocks and
rlShaders.mtd by tools/extract_param_surface.py).  Parameters the closure path does not consume
(diffuse Oren-Nayar lobe, Ks/Kt post-scales, opacity, AOV names, indirect scales) are listed
explicitly as out of path so that nothing is silently missing."""
import json
from pathlib import Path

from rlshaders_amd import _capi as capi
from rlshaders_amd.closures import SkinShader

FIXTURE = json.loads((Path(__file__).parent / "golden" / "param_surface.json").read_text())
G = FIXTURE["nodes"]

# declared by the reference's nodes, consumed outside the closure layer
OUT_OF_PATH = {
    "rlGgx": {"KdColor", "Kd", "diffuseRoughness", "Ks", "KtColor", "Kt", "opacity", "opacity_color"},
    "rlDisney": {"opacity", "indirectDiffuseScale", "indirectSpecularScale"},
    "rlSkin": {"sss_cavity_fadeout", "opacity", "opacity_color"},
}


def fields(struct):
    return {n: t for n, t in struct._fields_}


def check(node, struct, rename=None):
    rename = rename or {}
    f = fields(struct)
    for p in G[node]["parameters"]:
        name = p["name"]
        if name in OUT_OF_PATH[node] or p["type"] == "STR":        # AOV names: renderer glue
            continue
        fld = rename.get(name, name)
        assert fld in f, f"{node}.{name} missing from the C-ABI closure struct"
        if p["type"] == "RGB":
            assert f[fld] is capi.ParamRgb, (node, name)
        elif p["type"] == "VEC":
            assert f[fld]._length_ == 3 and f[fld]._type_ is capi.Param, (node, name)
        else:
            assert f[fld] is capi.Param, (node, name)


def test_rlggx_parameters():
    check("rlGgx", capi.GgxClosure)
    assert [p["name"] for p in G["rlGgx"]["parameters"]][:4] == ["KdColor", "Kd", "diffuseRoughness", "KsColor"]
    assert G["rlGgx"]["mtd"]["maya.id"] == "0x04700001"


def test_rldisney_parameters():
    check("rlDisney", capi.DisneyClosure)
    scal = [p["name"] for p in G["rlDisney"]["parameters"] if p["type"] == "FLT" and p["name"] not in OUT_OF_PATH["rlDisney"]]
    assert tuple(scal) == capi.DISNEY_SCALARS          # same order as src/rlDisney.cpp:608-610
    assert all(p["default"] == [0.0] for p in G["rlDisney"]["parameters"] if p["name"] in scal)


def test_rlskin_parameters_and_defaults():
    check("rlSkin", capi.SkinClosure)
    for p in G["rlSkin"]["parameters"]:
        if p["name"] in OUT_OF_PATH["rlSkin"] or p["type"] == "STR":
            continue
        d = SkinShader.DEFAULTS[p["name"]]
        got = list(d) if isinstance(d, (tuple, list)) else [d]
        assert got == p["default"], (p["name"], got, p["default"])
    # UI ranges in the .mtd are consistent with the defaults
    for name, r in G["rlSkin"]["mtd"]["attrs"].items():
        if name in SkinShader.DEFAULTS and "min" in r:
            assert SkinShader.DEFAULTS[name] >= r["min"]


def test_header_names_every_consumed_parameter():
    text = (Path(__file__).resolve().parent.parent / "include" / "rlshaders_amd.h").read_text()
    for node in G:
        for p in G[node]["parameters"]:
            if p["name"] not in OUT_OF_PATH[node] and p["type"] != "STR":
                assert p["name"] in text, (node, p["name"])


def test_product_parameter_table_matches_reference():
    """rlshaders_amd.params.NODES (what a stub declares) == the reference's node_parameters blocks, p_* enums,
    in-code metadata, node_loader table and .mtd"""
    from rlshaders_amd import params
    for node, spec in G.items():
        mine = params.NODES[node]["params"]
        ref = spec["parameters"]
        assert [p.name for p in mine] == [p["name"] for p in ref], node           # declaration order
        assert [p.enum for p in mine] == spec["enum"], node                       # = positional enum order
        for a, b in zip(mine, ref):
            assert a.type == b["type"] and list(a.default) == b["default"], (node, a.name)
            assert a.closure == (a.name not in OUT_OF_PATH[node] and a.type != "STR"), (node, a.name)
            assert dict(a.meta) == b["meta"], (node, a.name, a.meta, b["meta"])
        assert params.NODES[node]["maya.id"] == spec["mtd"]["maya.id"]
        # UI ranges of rlShaders.mtd (rlDisney sets its ranges in code, so its .mtd block has none)
        for attr, r in spec["mtd"]["attrs"].items():
            p = next(p for p in mine if p.name == attr)
            assert {k: v for k, v in (("min", p.min), ("max", p.max), ("softmax", p.softmax)) if v is not None} == r
    assert [(e["enumerator"], e["methods"], e["name"]) for e in FIXTURE["node_loader"]] == list(params.NODE_LOADER)
    assert all(e["output_type"] == "AI_TYPE_RGB" and e["node_type"] == "AI_NODE_SHADER" for e in FIXTURE["node_loader"])
    assert [e["id"] for e in FIXTURE["node_loader"]] == [0, 1, 2]


def test_emitted_mtd_round_trips():
    """the emitted metadata file parses back to the reference's ids and attribute ranges"""
    import re
    from rlshaders_amd import params
    node = attr = None
    got = {}
    for line in params.emit_mtd().splitlines():
        t = line.strip()
        m = re.match(r"\[node (\w+)\]", t)
        if m:
            node = m.group(1); got[node] = {"maya.id": None, "attrs": {}}; continue
        m = re.match(r"\[attr (\w+)\]", t)
        if m:
            attr = m.group(1); got[node]["attrs"][attr] = {}; continue
        m = re.match(r"maya\.id\s+INT\s+(\S+)", t)
        if m:
            got[node]["maya.id"] = m.group(1); continue
        m = re.match(r"(min|max|softmax)\s+FLOAT\s+(\S+)", t)
        if m:
            got[node]["attrs"][attr][m.group(1)] = float(m.group(2))
    for n, spec in G.items():
        assert got[n]["maya.id"] == spec["mtd"]["maya.id"]
        assert got[n]["attrs"] == spec["mtd"]["attrs"], n
