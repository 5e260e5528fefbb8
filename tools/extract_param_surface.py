#!/usr/bin/env python3
"""Extract the reference's closure parameter surface -- names, types, defaults (node_parameters blocks)
and UI ranges (src/rlShaders.mtd) -- into tests/golden/param_surface.json.  Runs only in the build
container (reads /root/reference); the JSON fixture is what travels.  Data, not source: one record
per declared parameter."""
import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "param_surface.json"


def node_parameters(path: Path):
    text = path.read_text()
    m = re.search(r"node_parameters\s*\{(.*?)\n\}", text, re.S)
    body = m.group(1)
    params = []
    # rlDisney declares its ten scalars through a loop over a name list
    lm = re.search(r"scalarAttrList\s*=\s*\{(.*?)\};", body, re.S)
    loop_names = re.findall(r'"([a-z_]+)"', lm.group(1)) if lm else []
    sm = re.search(r"softMaxAttrSet\s*=\s*\{(.*?)\};", body, re.S)
    hard_max = set(re.findall(r'"([a-z_]+)"', sm.group(1))) if sm else set()
    for line in body.splitlines():
        m = re.search(r'AiParameter(RGB|FLT|Flt|Vec|STR|Bool)\(\s*(?:"([A-Za-z_]+)"|attr\.c_str\(\))\s*,\s*(.*?)\)\s*;', line)
        if m:
            kind, name, default = m.group(1).upper(), m.group(2), m.group(3)
            if kind == "STR":
                params.append({"name": name, "type": "STR", "default": [default.strip().strip('"')], "meta": {}})
                continue
            vals = [float(v.rstrip("f")) for v in re.findall(r"-?\d+\.?\d*f?", default)] if kind != "BOOL" else [default.strip() == "true"]
            if name is None:
                for n in loop_names:
                    # src/rlDisney.cpp:612-620: min 0 and max (specular, roughness, sheen) or softmax 1, set in code
                    params.append({"name": n, "type": "FLT", "default": vals,
                                   "meta": {"min": 0.0, ("max" if n in hard_max else "softmax"): 1.0}})
            else:
                params.append({"name": name, "type": "FLT" if kind in ("FLT",) else kind, "default": vals, "meta": {}})
            continue
        # metadata set in code next to a declaration (always_linear, linkable, aov.type, min / max)
        m = re.search(r'AiMetaDataSet(Bool|Int|Flt)\(mds,\s*"([A-Za-z_]+)",\s*"([a-z_.]+)",\s*([^)]+)\)', line)
        if m:
            kind, name, key, val = m.groups()
            val = val.strip()
            v = (val == "true") if kind == "Bool" else (float(val.rstrip("f")) if kind == "Flt" else val)
            for q in params:
                if q["name"] == name:
                    q["meta"][key] = v
    return params


def param_enum(path: Path):
    """the positional p_* enumerators, in order (Arnold binds parameter ids by declaration order)"""
    m = re.search(r"enum\s+\w+Params\s*\{(.*?)\};", path.read_text(), re.S)
    return re.findall(r"\b(p_\w+)\b", m.group(1))


def node_loader(path: Path):
    text = path.read_text()
    ids = re.findall(r"\b(k\w+),", re.search(r"enum\s+ShaderId\s*\{(.*?)\};", text, re.S).group(1))
    out = []
    for case in re.finditer(r"case (k\w+):(.*?)break;", text, re.S):
        body = case.group(2)
        out.append({"id": ids.index(case.group(1)), "enumerator": case.group(1),
                    "methods": re.search(r"methods = (\w+);", body).group(1),
                    "output_type": re.search(r"output_type = (\w+);", body).group(1),
                    "name": re.search(r'name = "(\w+)";', body).group(1),
                    "node_type": re.search(r"node_type = (\w+);", body).group(1)})
    return out


def mtd_ranges(path: Path):
    out, node, attr = {}, None, None
    for line in path.read_text().splitlines():
        m = re.match(r"\[node (\w+)\]", line.strip())
        if m:
            node = m.group(1); out[node] = {"maya.id": None, "attrs": {}}; continue
        m = re.match(r"\[attr (\w+)\]", line.strip())
        if m:
            attr = m.group(1); out[node]["attrs"][attr] = {}; continue
        m = re.match(r"maya\.id\s+INT\s+(\S+)", line.strip())
        if m and node:
            out[node]["maya.id"] = m.group(1); continue
        m = re.match(r"(min|max|softmin|softmax)\s+FLOAT\s+(\S+)", line.strip())
        if m and node and attr:
            out[node]["attrs"][attr][m.group(1)] = float(m.group(2))
    return out


def main():
    if not REF.exists():
        sys.exit("reference not mounted: the committed fixture is used as is")
    surface = {"_provenance": "tools/extract_param_surface.py over src/rlGgx.cpp:106-126,170-198, "
                              "src/rlDisney.cpp:24-46,604-638, src/rlSkin.cpp:11-35,107-139, src/_PluginMain.cpp:8-46 and "
                              "src/rlShaders.mtd of the reference (one data record per declared parameter / enumerator)",
               "nodes": {}}
    ranges = mtd_ranges(REF / "rlShaders.mtd")
    for node, f in (("rlGgx", "rlGgx.cpp"), ("rlDisney", "rlDisney.cpp"), ("rlSkin", "rlSkin.cpp")):
        surface["nodes"][node] = {"parameters": node_parameters(REF / f), "enum": param_enum(REF / f),
                                  "mtd": ranges.get(node, {})}
    surface["node_loader"] = node_loader(REF / "_PluginMain.cpp")     # src/_PluginMain.cpp:16-46
    OUT.write_text(json.dumps(surface, indent=1) + "\n")
    print(OUT, {k: len(v["parameters"]) for k, v in surface["nodes"].items()})


if __name__ == "__main__":
    main()
