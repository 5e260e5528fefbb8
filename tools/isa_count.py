#!/usr/bin/env python3
"""Count gfx950 ISA instructions per kernel in a hipcc -S dump (static counts, all paths).

usage: isa_count.py file.s [kernel-substring]
"""
import re
import sys
from collections import Counter


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().split("\n")
    name = None
    ins = []
    for l in lines:
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", l)
        if m and not m.group(1).startswith(".L"):
            name = m.group(1)
            ins = []
            continue
        t = l.strip()
        if name is None or not t or t[0] in ".;" or t.endswith(":"):
            continue
        op = t.split()[0]
        ins.append(op)
        if op == "s_endpgm":
            if want in name:
                c = Counter(ins)
                valu = sum(n for k, n in c.items() if k.startswith("v_"))
                trans = sum(n for k, n in c.items() if re.match(r"v_(rcp|sqrt|rsq|sin|cos|exp|log)", k))
                salu = sum(n for k, n in c.items() if k.startswith("s_"))
                gl = sum(n for k, n in c.items() if k.startswith("global_load"))
                gs = sum(n for k, n in c.items() if k.startswith("global_store"))
                br = sum(n for k, n in c.items() if k.startswith("s_cbranch") or k == "s_branch")
                print(f"{name}: total={len(ins)} valu={valu} trans={trans} salu={salu} "
                      f"branches={br} gload={gl} gstore={gs}")
                if want:
                    print("  ", c.most_common(30))
            name = None


if __name__ == "__main__":
    main()
