#!/usr/bin/env python3
"""Compare two device-assembly listings kernel by kernel (hipcc -S --cuda-device-only of the same translation unit before and
after a change): number of instructions, how many lines differ, and how the differing lines split into VALU and other.
usage: tools/isa_compare.py old.s new.s [substring of the mangled or demangled kernel name ...]"""
import difflib, re, subprocess, sys


def kernels(path):
    out, cur = {}, None
    for l in open(path).read().splitlines():
        m = re.match(r"^(_Z\S+):\s*(;.*)?$", l)
        if m and "kernel" in m.group(1):
            cur = m.group(1)
            out[cur] = []
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
        if cur:
            t = l.split(";")[0].strip()
            if t and not t.startswith("."):
                out[cur].append(re.sub(r"\.LBB\d+_", ".LBB_", t))
    return out


def demangle(k):
    return subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    want = sys.argv[3:]
    for k in list(a) + [k for k in b if k not in a]:
        name = demangle(k)
        if want and not any(w in name or w in k for w in want):
            continue
        if k not in b:
            print(f"{name}: only in old"); continue
        if k not in a:
            nb = b[k]
            print(f"{name}: NEW  {len(nb)} instructions, {sum(t.startswith('v_') for t in nb)} VALU, "
                  f"{sum(t.startswith('s_memtime') or t.startswith('s_memrealtime') for t in nb)} counter reads")
            continue
        # opcodes only: register allocation renames operands all over a kernel when one scalar pair is freed; what matters is
        # which instructions run, in which order
        oa, ob = [t.split()[0] for t in a[k]], [t.split()[0] for t in b[k]]
        d = [l for l in difflib.unified_diff(oa, ob, lineterm="", n=0) if l[0] in "+-" and not l.startswith(("+++", "---"))]
        valu = sum(l[1:].startswith("v_") for l in d)
        va, vb = [o for o in oa if o.startswith("v_")], [o for o in ob if o.startswith("v_")]
        full = sum(1 for l in difflib.unified_diff(a[k], b[k], lineterm="", n=0) if l[0] in "+-" and not l.startswith(("+++", "---")))
        print(f"{name}: {len(oa)} -> {len(ob)} instructions ({len(va)} -> {len(vb)} VALU); opcode sequence: {len(d)} differing lines "
              f"({valu} VALU); VALU opcode sequence {'identical' if va == vb else 'differs'}; with operands {full} lines differ"
              + ("; IDENTICAL" if not full else ""))


if __name__ == "__main__":
    main()
