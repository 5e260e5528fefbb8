"""The parameter surface of the three shader nodes -- the contract an Arnold-side stub keeps when it
forwards shading points to this library: names, types, defaults in declaration
order (= Arnold's positional ``p_*`` enum order) and the UI ranges of ``rlShaders.mtd``.

``closure`` marks what the batched closures consume (and therefore what the C-ABI closure structs
name); the rest is renderer glue the stub keeps to itself (diffuse Oren-Nayar lobe, post-scales,
opacity, AOV names).  ``emit_mtd()`` writes the Arnold metadata file for a plugin built on the stub.
tests/test_param_surface.py checks all of it against the fixture extracted from the reference.
"""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional, Tuple


class Param(NamedTuple):
    name: str
    type: str                      # RGB | FLT | VEC | BOOL | STR
    default: Tuple
    enum: str                      # the positional p_* enumerator the node's shader_evaluate reads it through
    closure: bool = True           # consumed by the batched closure layer
    min: Optional[float] = None    # UI ranges of rlShaders.mtd
    max: Optional[float] = None
    softmax: Optional[float] = None
    meta: Tuple = ()               # metadata the node sets in code next to the declaration: ((key, value), ...)


def _aov(name, default, enum):
    return Param(name, "STR", (default,), enum, closure=False, meta=(("aov.type", "AI_TYPE_RGB"),))


def _d(name, enum, hard_max=False):
    # src/rlDisney.cpp:612-620: min 0 and max 1 (specular, roughness, sheen) or softmax 1, set in code
    return Param(name, "FLT", (0.0,), enum, meta=(("min", 0.0), ("max" if hard_max else "softmax", 1.0)))


_LIN = (("always_linear", True),)

# node_loader table, src/_PluginMain.cpp:8-46: (ShaderId enumerator, methods symbol, node name); every node is an
# AI_NODE_SHADER with output type AI_TYPE_RGB
NODE_LOADER = (("kGgx", "GgxMethod", "rlGgx"), ("kDisney", "DisneyMethod", "rlDisney"), ("kSkin", "SkinMethod", "rlSkin"))

NODES: Dict[str, dict] = {
    # enum src/rlGgx.cpp:106-126, declarations src/rlGgx.cpp:172-197, ranges src/rlShaders.mtd:1-29
    "rlGgx": {"maya.id": "0x04700001", "params": [
        Param("KdColor", "RGB", (1.0, 1.0, 1.0), "p_Kd_color", closure=False),
        Param("Kd", "FLT", (0.5,), "p_Kd", closure=False, min=0.0, softmax=1.0),
        Param("diffuseRoughness", "FLT", (0.0,), "p_Kd_roughness", closure=False, min=0.0, softmax=1.0),
        Param("KsColor", "RGB", (1.0, 1.0, 1.0), "p_Ks_color"),
        Param("Ks", "FLT", (0.5,), "p_Ks", closure=False, min=0.0, softmax=1.0),
        Param("specularRoughness", "FLT", (0.0,), "p_Ks_roughness", min=0.0, softmax=1.0),
        Param("KtColor", "RGB", (1.0, 1.0, 1.0), "p_Kt_Color", closure=False),
        Param("Kt", "FLT", (0.0,), "p_Kt", closure=False, min=0.0, softmax=1.0),
        Param("ior", "FLT", (1.0,), "p_ior", min=0.0),
        Param("anisotropic", "FLT", (0.0,), "p_anisotropic", min=0.0, softmax=1.0),
        Param("opacity", "FLT", (1.0,), "p_opacity", closure=False, min=0.0, max=1.0),
        Param("opacity_color", "RGB", (1.0, 1.0, 1.0), "p_opacity_color", closure=False),
        _aov("aov_direct_diffuse", "direct_diffuse", "p_aov_direct_diffuse"),
        _aov("aov_direct_specular", "direct_specular", "p_aov_direct_specular"),
        _aov("aov_refract", "refraction", "p_aov_refract"),
        _aov("aov_indirect_diffuse", "indirect_diffuse", "p_aov_indirect_diffuse"),
        _aov("aov_indirect_specular", "indirect_specular", "p_aov_indirect_specular"),
    ]},
    # enum src/rlDisney.cpp:24-46, declarations src/rlDisney.cpp:606-637, src/rlShaders.mtd:31-35
    "rlDisney": {"maya.id": "0x04700002", "params": [
        Param("base_color", "RGB", (1.0, 1.0, 1.0), "p_base_color"),
        _d("subsurface", "p_subsurface"), _d("metallic", "p_metallic"), _d("specular", "p_Ks", True),
        _d("specular_tint", "p_specular_tint"), _d("roughness", "p_roughness", True), _d("anisotropic", "p_anisotropic"),
        _d("sheen", "p_sheen", True), _d("sheen_tint", "p_sheen_tint"), _d("clearcoat", "p_clearcoat"),
        _d("clearcoat_gloss", "p_clearcoat_gloss"),
        Param("opacity", "RGB", (1.0, 1.0, 1.0), "p_opacity", closure=False),
        Param("indirectDiffuseScale", "FLT", (1.0,), "p_indirect_diffuse", closure=False, meta=(("min", 0.0), ("max", 1.0))),
        Param("indirectSpecularScale", "FLT", (1.0,), "p_indirect_specular", closure=False, meta=(("min", 0.0), ("max", 1.0))),
        _aov("aov_direct_diffuse", "direct_diffuse", "p_aov_direct_diffuse"),
        _aov("aov_direct_specular", "direct_specular", "p_aov_direct_specular"),
        _aov("aov_indirect_diffuse", "indirect_diffuse", "p_aov_indirect_diffuse"),
        _aov("aov_indirect_specular", "indirect_specular", "p_aov_indirect_specular"),
    ]},
    # enum src/rlSkin.cpp:11-35, declarations src/rlSkin.cpp:109-138, src/rlShaders.mtd:37-64
    "rlSkin": {"maya.id": "0x04700003", "params": [
        Param("sss_color", "RGB", (1.0, 1.0, 1.0), "p_sss_color", meta=_LIN),
        Param("sss_weight", "FLT", (1.0,), "p_sss_weight", min=0.0, softmax=1.0),
        Param("sss_dist_multiplier", "FLT", (1.0,), "p_distance_multiplier", min=0.0, softmax=3.0),
        Param("sss_scatter_dist", "VEC", (1.0, 1.0, 1.0), "p_scatter_distance"),
        Param("sss_cavity_fadeout", "BOOL", (True,), "p_cavity_fadeout", closure=False, meta=(("linkable", False),)),
        Param("specular_color", "RGB", (1.0, 1.0, 1.0), "p_specular_color", meta=_LIN),
        Param("specular_weight", "FLT", (0.6,), "p_specular_weight", min=0.0, softmax=1.0),
        Param("specular_roughness", "FLT", (0.5,), "p_specular_roughness", min=0.0, softmax=1.0),
        Param("specular_ior", "FLT", (1.44,), "p_specular_ior", min=0.0),
        Param("sheen_color", "RGB", (1.0, 1.0, 1.0), "p_sheen_color", meta=_LIN),
        Param("sheen_weight", "FLT", (0.0,), "p_sheen_weight", min=0.0, softmax=1.0),
        Param("sheen_roughness", "FLT", (0.35,), "p_sheen_roughness", min=0.0, max=1.0),
        Param("sheen_ior", "FLT", (1.44,), "p_sheen_ior", min=0.0),
        Param("opacity", "FLT", (1.0,), "p_opacity", closure=False),
        Param("opacity_color", "RGB", (1.0, 1.0, 1.0), "p_opacity_color", closure=False),
        _aov("aov_sheen", "sheen", "p_aov_sheen"), _aov("aov_specular", "specular", "p_aov_specular"),
        _aov("aov_sss", "sss", "p_aov_sss"),
    ]},
}


def closure_parameters(node: str) -> List[Param]:
    return [p for p in NODES[node]["params"] if p.closure]


def emit_mtd() -> str:
    """Arnold metadata (.mtd) for a plugin that registers the three nodes (same ids, same attribute
    ranges as the reference's src/rlShaders.mtd)."""
    out = []
    for node, spec in NODES.items():
        out.append(f"[node {node}]")
        out.append('    desc                    STRING      ""')
        out.append(f'    maya.name               STRING      "{node}"')
        out.append('    maya.classification     STRING      "shader/surface"')
        out.append(f"    maya.id                 INT         {spec['maya.id']}")
        out.append("")
        for p in spec["params"]:
            if p.min is None and p.max is None and p.softmax is None:
                continue
            out.append(f"    [attr {p.name}]")
            for key in ("min", "max", "softmax"):
                v = getattr(p, key)
                if v is not None:
                    out.append(f"        {key:<19s} FLOAT   {v}")
        out.append("")
    return "\n".join(out)
