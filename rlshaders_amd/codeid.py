"""Identity of the DEVICE code inside a built librlshaders_amd.so.

The committed counter profiles (profiles/*_{traffic,flops,clock,stalls}.json) describe kernels of one particular
binary.  This module says which: it reads the gfx950 code objects straight out of the shared library's ``.hip_fatbin``
section (one clang offload bundle per translation unit) and hashes what the GPU executes --

* ``unit_id``: sha-256 (16 hex digits) over the loadable contents of ONE code object: ``.text`` (instructions),
  ``.rodata`` (kernel descriptors: register counts, LDS size, scratch) and ``.data``;
* ``library_id``: sha-256 over the sorted unit ids -- independent of link order, so the library built by
  rlshaders_amd/build.py and the one built by CMakeLists.txt carry the same id exactly when every kernel in them is the
  same machine code;
* ``unit_of_kernel(name)``: the unit that holds a kernel, found by its Itanium-mangled name fragment (``ggx_kernel<5, 0, 1>``
  -> ``10ggx_kernelILi5ELi0ELi1EE``), so a profile taken by exact kernel name is tied to that kernel's unit and is not
  invalidated by a change to some other unit or to host code.

Host code, comments and link order do not enter; compiler flags and every header a kernel includes do, through the
instructions they produce.  Pure Python (struct + hashlib), no tool of the ROCm installation is run: bench.py calls it on
the GPU box for the library it has just loaded.
"""
from __future__ import annotations

import hashlib
import re
import struct
from functools import lru_cache
from pathlib import Path
from typing import Dict, List, Optional, Tuple

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
HASHED_SECTIONS = (".text", ".rodata", ".data")


def _sections(elf: bytes) -> Dict[str, Tuple[int, int, int]]:
    """name -> (file offset, size, sh_type) of an ELF64 little-endian image"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        raise ValueError("not a little-endian ELF64 image")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    heads = []
    for k in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + k * shentsize)
        heads.append((name, typ, off, size))
    _, _, stroff, strsize = heads[shstrndx]
    strtab = elf[stroff:stroff + strsize]
    out = {}
    for name, typ, off, size in heads:
        end = strtab.index(b"\0", name)
        out[strtab[name:end].decode()] = (off, size, typ)
    return out


def _symbols(elf: bytes, secs) -> List[str]:
    """names of the FUNC symbols (the kernels and whatever device functions were not inlined)"""
    if ".symtab" not in secs or ".strtab" not in secs:
        return []
    off, size, _ = secs[".symtab"]
    soff, ssize, _ = secs[".strtab"]
    strtab = elf[soff:soff + ssize]
    names = []
    for k in range(size // 24):
        name, info = struct.unpack_from("<IB", elf, off + 24 * k)
        if info & 0xF == 2:                                   # STT_FUNC
            names.append(strtab[name:strtab.index(b"\0", name)].decode())
    return names


def fatbin(path) -> bytes:
    """the .hip_fatbin section of a host shared library or object"""
    data = Path(path).read_bytes()
    secs = _sections(data)
    if ".hip_fatbin" not in secs:
        raise ValueError(f"{path}: no .hip_fatbin section (not a HIP library)")
    off, size, _ = secs[".hip_fatbin"]
    return data[off:off + size]


def code_objects(fb: bytes) -> List[bytes]:
    """every amdgcn code object of the fat binary (uncompressed clang offload bundles, as hipcc of ROCm 7.2 writes them)"""
    out = []
    for m in re.finditer(re.escape(BUNDLE_MAGIC), fb):
        base = m.start()
        p = base + len(BUNDLE_MAGIC)
        count, = struct.unpack_from("<Q", fb, p)
        p += 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", fb, p)
            p += 24
            triple = fb[p:p + tlen].decode(errors="replace")
            p += tlen
            if "amdgcn" in triple and size:
                out.append(fb[base + off:base + off + size])
    if not out and b"CCOB" in fb:
        raise ValueError("compressed offload bundles: build with --no-offload-compress (rlshaders_amd/build.py does)")
    return out


def unit_digest(elf: bytes) -> Tuple[str, List[str]]:
    """(unit id, FUNC symbol names) of one code object"""
    secs = _sections(elf)
    h = hashlib.sha256()
    for name in HASHED_SECTIONS:
        if name in secs:
            off, size, typ = secs[name]
            body = b"" if typ == 8 else elf[off:off + size]         # SHT_NOBITS occupies no file bytes
            h.update(name.encode() + b"\0" + struct.pack("<Q", size) + body)
    return h.hexdigest()[:16], _symbols(elf, secs)


def mangled_fragment(kernel: str) -> Optional[str]:
    """`ggx_kernel<5, 0, 1>` (or rocprofv3's `void (anonymous namespace)::ggx_kernel<5, 0, 1>(rlsh::GgxIO)`) -> the
    Itanium fragment `10ggx_kernelILi5ELi0ELi1EE` every mangled name of that instantiation contains.  Integer template
    arguments only (all the kernels here)."""
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*<([^<>]*)>\s*(?:\(|$)", kernel.strip())
    if not m:
        m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\(|$)", kernel.strip())
        return f"{len(m.group(1))}{m.group(1)}" if m else None
    name, args = m.group(1), [a.strip() for a in m.group(2).split(",") if a.strip()]
    enc = ""
    for a in args:
        if not re.fullmatch(r"-?\d+", a):
            return None
        enc += f"Li{'n' + a[1:] if a.startswith('-') else a}E"
    return f"{len(name)}{name}I{enc}E"


class DeviceCode:
    """the code objects of one library: ids and kernel lookup"""

    def __init__(self, path):
        self.path = str(path)
        self.units = [unit_digest(e) for e in code_objects(fatbin(path))]          # [(unit id, [symbols])]
        self.library_id = hashlib.sha256("\n".join(sorted(u for u, _ in self.units)).encode()).hexdigest()[:16]

    def unit_of_kernel(self, kernel: str) -> Optional[str]:
        frag = mangled_fragment(kernel)
        if not frag:
            return None
        hits = sorted({u for u, syms in self.units if any(frag in s for s in syms)})
        return hits[0] if len(hits) == 1 else None

    def record(self, kernel: Optional[str] = None) -> dict:
        rec = {"library_id": self.library_id}
        if kernel:
            rec["kernel"] = kernel
            rec["unit_id"] = self.unit_of_kernel(kernel)
        return rec


@lru_cache(maxsize=8)
def _cached(path: str, mtime: float, size: int) -> DeviceCode:
    return DeviceCode(path)


def device_code(path) -> DeviceCode:
    p = Path(path).resolve()
    st = p.stat()
    return _cached(str(p), st.st_mtime, st.st_size)


if __name__ == "__main__":
    # python -m rlshaders_amd.codeid [library.so] [kernel name ...]
    import json
    import sys
    args = sys.argv[1:]
    lib = args[0] if args and args[0].endswith((".so", ".o")) or (args and ".so." in args[0]) else None
    if lib is None:
        from rlshaders_amd.build import LIB
        lib, names = LIB, args
    else:
        names = args[1:]
    dc = DeviceCode(lib)
    out = {"library": str(lib), "library_id": dc.library_id, "units": len(dc.units)}
    if names:
        out["kernels"] = {k: dc.unit_of_kernel(k) for k in names}
    else:
        out["unit_ids"] = sorted(u for u, _ in dc.units)
    print(json.dumps(out, indent=1))
