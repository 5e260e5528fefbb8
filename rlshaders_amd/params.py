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
    closure: bool = True           # consumed by the batched closure layer
    min: Optional[float] = None
    max: Optional[float] = None
    softmax: Optional[float] = None


def _aov(name, default):
    return Param(name, "STR", (default,), closure=False)


NODES: Dict[str, dict] = {
    # src/rlGgx.cpp:172-197, src/rlShaders.mtd:1-29
    "rlGgx": {"maya.id": "0x04700001", "params": [
        Param("KdColor", "RGB", (1.0, 1.0, 1.0), closure=False),
        Param("Kd", "FLT", (0.5,), closure=False, min=0.0, softmax=1.0),
        Param("diffuseRoughness", "FLT", (0.0,), closure=False, min=0.0, softmax=1.0),
        Param("KsColor", "RGB", (1.0, 1.0, 1.0)),
        Param("Ks", "FLT", (0.5,), closure=False, min=0.0, softmax=1.0),
        Param("specularRoughness", "FLT", (0.0,), min=0.0, softmax=1.0),
        Param("KtColor", "RGB", (1.0, 1.0, 1.0), closure=False),
        Param("Kt", "FLT", (0.0,), closure=False, min=0.0, softmax=1.0),
        Param("ior", "FLT", (1.0,), min=0.0),
        Param("anisotropic", "FLT", (0.0,), min=0.0, softmax=1.0),
        Param("opacity", "FLT", (1.0,), closure=False, min=0.0, max=1.0),
        Param("opacity_color", "RGB", (1.0, 1.0, 1.0), closure=False),
        _aov("aov_direct_diffuse", "direct_diffuse"), _aov("aov_direct_specular", "direct_specular"),
        _aov("aov_refract", "refraction"), _aov("aov_indirect_diffuse", "indirect_diffuse"),
        _aov("aov_indirect_specular", "indirect_specular"),
    ]},
    # src/rlDisney.cpp:606-637 (ranges are set in code there, src/rlDisney.cpp:612-620), src/rlShaders.mtd:31-35
    "rlDisney": {"maya.id": "0x04700002", "params": [
        Param("base_color", "RGB", (1.0, 1.0, 1.0)),
        Param("subsurface", "FLT", (0.0,)), Param("metallic", "FLT", (0.0,)), Param("specular", "FLT", (0.0,)),
        Param("specular_tint", "FLT", (0.0,)), Param("roughness", "FLT", (0.0,)), Param("anisotropic", "FLT", (0.0,)),
        Param("sheen", "FLT", (0.0,)), Param("sheen_tint", "FLT", (0.0,)), Param("clearcoat", "FLT", (0.0,)),
        Param("clearcoat_gloss", "FLT", (0.0,)),
        Param("opacity", "RGB", (1.0, 1.0, 1.0), closure=False),
        Param("indirectDiffuseScale", "FLT", (1.0,), closure=False),
        Param("indirectSpecularScale", "FLT", (1.0,), closure=False),
        _aov("aov_direct_diffuse", "direct_diffuse"), _aov("aov_direct_specular", "direct_specular"),
        _aov("aov_indirect_diffuse", "indirect_diffuse"), _aov("aov_indirect_specular", "indirect_specular"),
    ]},
    # src/rlSkin.cpp:109-138, src/rlShaders.mtd:37-64
    "rlSkin": {"maya.id": "0x04700003", "params": [
        Param("sss_color", "RGB", (1.0, 1.0, 1.0)),
        Param("sss_weight", "FLT", (1.0,), min=0.0, softmax=1.0),
        Param("sss_dist_multiplier", "FLT", (1.0,), min=0.0, softmax=3.0),
        Param("sss_scatter_dist", "VEC", (1.0, 1.0, 1.0)),
        Param("sss_cavity_fadeout", "BOOL", (True,), closure=False),
        Param("specular_color", "RGB", (1.0, 1.0, 1.0)),
        Param("specular_weight", "FLT", (0.6,), min=0.0, softmax=1.0),
        Param("specular_roughness", "FLT", (0.5,), min=0.0, softmax=1.0),
        Param("specular_ior", "FLT", (1.44,), min=0.0),
        Param("sheen_color", "RGB", (1.0, 1.0, 1.0)),
        Param("sheen_weight", "FLT", (0.0,), min=0.0, softmax=1.0),
        Param("sheen_roughness", "FLT", (0.35,), min=0.0, max=1.0),
        Param("sheen_ior", "FLT", (1.44,), min=0.0),
        Param("opacity", "FLT", (1.0,), closure=False),
        Param("opacity_color", "RGB", (1.0, 1.0, 1.0), closure=False),
        _aov("aov_sheen", "sheen"), _aov("aov_specular", "specular"), _aov("aov_sss", "sss"),
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
