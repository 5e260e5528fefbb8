"""Identity of the DEVICE code inside a built librlshaders_amd.so.

The committed counter profiles (profiles/*_{traffic,flops,clock,stalls}.json) describe kernels of one particular
binary.  This module says which: it reads the gfx950 code objects straight out of the shared library's ``.hip_fatbin``
section (one clang offload bundle per translation unit) and hashes what the GPU executes --

* ``unit_id``: sha-256 (16 hex digits) over the loadable contents of ONE code object: ``.text`` (instructions),
  ``.rodata`` (kernel descriptors: register counts, LDS size, scratch; constants) and ``.data``;
* ``library_id``: sha-256 over the sorted unit ids -- independent of link order, so the library built by
  rlshaders_amd/build.py and the one built by CMakeLists.txt carry the same id exactly when every kernel in them is the
  same machine code;
* ``unit_of_kernel(name)``: the unit that holds a kernel, found by its Itanium-mangled name fragment (``ggx_kernel<5, 0, 1>``
  -> ``10ggx_kernelILi5ELi0ELi1EE``), so a profile taken by exact kernel name is tied to that kernel's unit and is not
  invalidated by a change to some other unit or to host code;
* ``kernel_id(name)``: sha-256 over ONE kernel -- its instructions (the function's bytes in ``.text``; every device function
  is inlined, the code objects hold no other FUNC symbol) with the PC-relative ADDRESS LITERALS blanked, its 64-byte kernel
  descriptor, and the unit's constant data (``.rodata`` with the descriptors cut out -- here the libm tables -- and
  ``.data``).  The literals are the two dwords behind ``s_add_u32`` / ``s_addc_u32 ... <literal>`` that follow an
  ``s_getpc_b64``: the distance from the instruction to a constant, i.e. layout.  So a kernel keeps its kernel_id when it
  moves to another translation unit, when a neighbour joins or leaves its unit (the diagnostic instantiations of
  RLS_DIAGNOSTICS) or when the unit is laid out differently -- and loses it with any change to its instructions, registers,
  LDS, scratch or constants.  (Round 6 first hashed the raw bytes: round 5's counter files of config 3, taken before
  ``integrate.hip`` was split, then read as stale over four moved literals.)

Host code, comments and link order do not enter; compiler flags and every header a kernel includes do, through the
instructions they produce.  One field of every kernel descriptor is left out: KERNEL_CODE_ENTRY_BYTE_OFFSET, the distance
from the descriptor to the kernel's first instruction.  It is layout, not code, and it moves by a cache line with the
length of the ``__hip_cuid_<hash>`` symbol clang derives from the PATHS on its command line (a 15-digit hash instead of a
16-digit one shortens ``.dynstr`` by a byte): the same sources compiled into another object directory would otherwise read
as other device code now and again.  (Both builds of this repository pass a fixed `-cuid=` per unit since round 6, which removes
the cause; leaving the field out keeps the ids stable for libraries built any other way.)  Pure Python (struct + hashlib), no tool of the ROCm installation is run: bench.py
calls it on the GPU box for the library it has just loaded.
"""
from __future__ import annotations

import hashlib
import re
import struct
from functools import lru_cache
from pathlib import Path
from typing import Dict, List, Optional, Tuple

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
KERNEL_ID_SCHEME = 2        # 1: raw instruction bytes (first half of round 6); 2: address literals blanked, descriptors cut out of the constants
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


def _symtab(elf: bytes, secs) -> List[Tuple[str, int, int, int, int]]:
    """(name, type, section index, value, size) of every symbol"""
    if ".symtab" not in secs or ".strtab" not in secs:
        return []
    off, size, _ = secs[".symtab"]
    soff, ssize, _ = secs[".strtab"]
    strtab = elf[soff:soff + ssize]
    out = []
    for k in range(size // 24):
        name, info, _other, shndx, value, sz = struct.unpack_from("<IBBHQQ", elf, off + 24 * k)
        out.append((strtab[name:strtab.index(b"\0", name)].decode(), info & 0xF, shndx, value, sz))
    return out


def _symbols(elf: bytes, secs) -> List[str]:
    """names of the FUNC symbols (the kernels and whatever device functions were not inlined)"""
    return [s[0] for s in _symtab(elf, secs) if s[1] == 2]             # STT_FUNC


def _section_table(elf: bytes) -> List[Tuple[str, int, int, int, int]]:
    """(name, type, address, file offset, size) by section index"""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    raw = [struct.unpack_from("<IIQQQQ", elf, shoff + k * shentsize) for k in range(shnum)]
    stroff, strsize = raw[shstrndx][4], raw[shstrndx][5]
    strtab = elf[stroff:stroff + strsize]
    return [(strtab[n:strtab.index(b"\0", n)].decode(), typ, addr, off, size) for n, typ, _f, addr, off, size in raw]


def _mask_pc_relative(code: bytes) -> bytes:
    """the kernel's instruction bytes with the literals of `s_getpc_b64 sN` ... `s_add_u32 / s_addc_u32 sM, sM, <literal>`
    sequences blanked (gfx9 encodings: SOP1 0xBE80_1C00 | sdst << 16; SOP2 op 0 / op 4 with SSRC1 = 0xFF, the literal marker;
    the compiler may schedule a few instructions in between)"""
    n = len(code) // 4
    if n == 0 or len(code) % 4:
        return code
    w = list(struct.unpack(f"<{n}I", code))
    for i in range(n):
        if (w[i] & 0xFF80FFFF) != 0xBE801C00:                      # s_getpc_b64
            continue
        j, found = i + 1, 0
        while j < min(n - 1, i + 12) and found < 2:
            if (w[j] & 0xFF80FF00) in (0x8000FF00, 0x8200FF00):     # s_add_u32 / s_addc_u32 with a literal
                w[j + 1] = 0
                found += 1
                j += 2
            else:
                j += 1
    return struct.pack(f"<{n}I", *w)


KD_SIZE = 64
KD_ENTRY_OFFSET = slice(16, 24)         # amd_kernel_descriptor_t.kernel_code_entry_byte_offset (int64): layout, not code


def _normalised_kd(kd: bytes) -> bytes:
    b = bytearray(kd)
    if len(b) == KD_SIZE:
        b[KD_ENTRY_OFFSET] = b"\0" * 8
    return bytes(b)


def kernel_digests(elf: bytes) -> Dict[str, str]:
    """mangled kernel name -> kernel id (module docstring) for every kernel of one code object"""
    table = _section_table(elf)
    secs = {name: (off, size, typ) for name, typ, _a, off, size in table}
    syms = _symtab(elf, secs)

    def body(sym):
        _n, _t, shndx, value, size = sym
        if not (0 < shndx < len(table)):
            return None
        _sn, typ, addr, off, _ss = table[shndx]
        return None if typ == 8 else elf[off + value - addr:off + value - addr + size]

    kds = {s[0][:-3]: s for s in syms if s[0].endswith(".kd")}
    # the unit's constant data: .rodata with the kernel descriptors CUT OUT (how many kernels share the unit does not enter), .data
    const = hashlib.sha256()
    for name in (".rodata", ".data"):
        idx = [k for k, s in enumerate(table) if s[0] == name]
        if not idx or table[idx[0]][1] == 8:
            continue
        _sn, _typ, addr, off, size = table[idx[0]]
        blob, keep, pos = elf[off:off + size], bytearray(), 0
        for lo, sz in sorted((s[3] - addr, s[4]) for s in kds.values() if s[2] == idx[0]):
            keep += blob[pos:lo]
            pos = lo + sz
        keep += blob[pos:]
        const.update(name.encode() + b"\0" + bytes(keep))
    const = const.digest()
    out = {}
    for s in syms:
        if s[1] != 2 or s[0] not in kds:
            continue
        code, kd = body(s), body(kds[s[0]])
        if code is None or kd is None:
            continue
        out[s[0]] = hashlib.sha256(struct.pack("<QQ", len(code), len(kd)) + _mask_pc_relative(code) + _normalised_kd(kd) +
                                   const).hexdigest()[:16]
    return out


def kernel_images(elf: bytes) -> Dict[str, Tuple[bytes, bytes]]:
    """mangled kernel name -> (the kernel's instruction bytes, its descriptor without the entry offset) of one code object"""
    table = _section_table(elf)
    secs = {name: (off, size, typ) for name, typ, _a, off, size in table}
    syms = _symtab(elf, secs)
    kds = {s[0][:-3]: s for s in syms if s[0].endswith(".kd")}
    out = {}
    for s in syms:
        if s[1] != 2 or s[0] not in kds:
            continue
        _sn, _typ, addr, off, _ss = table[s[2]]
        k = kds[s[0]]
        _kn, _kt, kaddr, koff, _ks = table[k[2]]
        out[s[0]] = (elf[off + s[3] - addr:off + s[3] - addr + s[4]], _normalised_kd(elf[koff + k[3] - kaddr:koff + k[3] - kaddr + k[4]]))
    return out


def same_code_up_to_relocation(a: bytes, b: bytes) -> bool:
    """Two images of one kernel from differently laid out code objects: equal, or of equal length with every differing dword
    off by ONE common amount -- the PC-relative address literals of a kernel whose unit's constant data sits at another distance
    (a hash cannot see through that; a pairwise comparison can)."""
    if a == b:
        return True
    if len(a) != len(b) or len(a) % 4:
        return False
    n = len(a) // 4
    wa, wb = struct.unpack(f"<{n}I", a), struct.unpack(f"<{n}I", b)
    deltas = {(y - x) & 0xFFFFFFFF for x, y in zip(wa, wb) if x != y}
    return len(deltas) == 1


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
    table = _section_table(elf)
    secs = {name: (off, size, typ) for name, typ, _a, off, size in table}
    syms = _symtab(elf, secs)
    h = hashlib.sha256()
    for name in HASHED_SECTIONS:
        idx = [k for k, s in enumerate(table) if s[0] == name]
        if not idx:
            continue
        _n, typ, addr, off, size = table[idx[0]]
        body = bytearray(b"" if typ == 8 else elf[off:off + size])         # SHT_NOBITS occupies no file bytes
        if body:
            for s in syms:                                                  # the descriptors' entry offsets are layout
                if s[0].endswith(".kd") and s[2] == idx[0] and s[4] == KD_SIZE:
                    lo = s[3] - addr
                    body[lo:lo + KD_SIZE] = _normalised_kd(bytes(body[lo:lo + KD_SIZE]))
        h.update(name.encode() + b"\0" + struct.pack("<Q", size) + bytes(body))
    return h.hexdigest()[:16], [s[0] for s in syms if s[1] == 2]


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
        objs = code_objects(fatbin(path))
        self.units = [unit_digest(e) for e in objs]                                # [(unit id, [symbols])]
        self.kernels: Dict[str, str] = {}                                          # mangled name -> kernel id
        for e in objs:
            self.kernels.update(kernel_digests(e))
        self.library_id = hashlib.sha256("\n".join(sorted(u for u, _ in self.units)).encode()).hexdigest()[:16]

    def unit_of_kernel(self, kernel: str) -> Optional[str]:
        frag = mangled_fragment(kernel)
        if not frag:
            return None
        hits = sorted({u for u, syms in self.units if any(frag in s for s in syms)})
        return hits[0] if len(hits) == 1 else None

    def kernel_id(self, kernel: str) -> Optional[str]:
        frag = mangled_fragment(kernel)
        if not frag:
            return None
        hits = sorted({k for name, k in self.kernels.items() if frag in name})
        return hits[0] if len(hits) == 1 else None

    def record(self, kernel: Optional[str] = None) -> dict:
        rec = {"library_id": self.library_id}
        if kernel:
            rec["kernel"] = kernel
            rec["unit_id"] = self.unit_of_kernel(kernel)
            rec["kernel_id"] = self.kernel_id(kernel)
            rec["kernel_id_scheme"] = KERNEL_ID_SCHEME
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
        out["kernels"] = {k: {"unit_id": dc.unit_of_kernel(k), "kernel_id": dc.kernel_id(k)} for k in names}
    else:
        out["unit_ids"] = sorted(u for u, _ in dc.units)
    print(json.dumps(out, indent=1))
