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
    for line in body.splitlines():
        m = re.search(r'AiParameter(RGB|FLT|Flt|Vec|STR|Bool)\(\s*(?:"([A-Za-z_]+)"|attr\.c_str\(\))\s*,\s*(.*?)\)\s*;', line)
        if not m:
            continue
        kind, name, default = m.group(1).upper(), m.group(2), m.group(3)
        if kind == "STR":
            continue                      # AOV names: out of the closure path
        vals = [float(v.rstrip("f")) for v in re.findall(r"-?\d+\.?\d*f?", default)] if kind != "BOOL" else [default.strip() == "true"]
        if name is None:
            for n in loop_names:
                params.append({"name": n, "type": "FLT", "default": vals})
        else:
            params.append({"name": name, "type": "FLT" if kind in ("FLT",) else kind, "default": vals})
    return params


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
    surface = {"_provenance": "tools/extract_param_surface.py over src/rlGgx.cpp:170-198, src/rlDisney.cpp:604-638, "
                              "src/rlSkin.cpp:107-139 and src/rlShaders.mtd of the reference",
               "nodes": {}}
    ranges = mtd_ranges(REF / "rlShaders.mtd")
    for node, f in (("rlGgx", "rlGgx.cpp"), ("rlDisney", "rlDisney.cpp"), ("rlSkin", "rlSkin.cpp")):
        surface["nodes"][node] = {"parameters": node_parameters(REF / f), "mtd": ranges.get(node, {})}
    OUT.write_text(json.dumps(surface, indent=1) + "\n")
    print(OUT, {k: len(v["parameters"]) for k, v in surface["nodes"].items()})


if __name__ == "__main__":
    main()
