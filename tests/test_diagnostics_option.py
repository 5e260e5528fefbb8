"""CPU: the build option RLS_DIAGNOSTICS=0 ("what's weak" 6: measurement surface ships in the product).

With it the library exports exactly the drop-in surface of include/rlshaders_amd.h, no code object holds a `*_kernel_stamped`
instantiation, no launch path tests for stamps -- and the PRODUCT kernels are the same machine code as in the default build:
215 of the 243 byte for byte, the other 28 -- the kernels that sit behind a removed instantiation in their unit and hold
address literals, the four BASELINE kernels among them -- of the same length, with the same descriptor (registers, LDS,
scratch) and identical up to ONE common relocation amount (4 ... 68 bytes of 25 ... 88 k: PC-relative literals of a unit
whose constant data now sits at another distance).  So everything measured and soaked on the default build describes the
kernels a maintainer ships with the option off."""
import subprocess

import pytest

from test_capi_symbols import declared_symbols, HEADERS


@pytest.fixture(scope="module")
def libs():
    from rlshaders_amd import build as b
    default = b.build_library()
    nodiag = b.build_library(variant="nodiag", defines=["RLS_DIAGNOSTICS=0"])       # objects cached under build_nodiag/
    return default, nodiag


def _exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("rls_"))


def test_exports_exactly_the_drop_in_surface(libs):
    default, nodiag = libs
    assert _exported(nodiag) == declared_symbols(HEADERS[:1])
    assert _exported(default) == declared_symbols(HEADERS)
    assert not [s for s in _exported(nodiag) if "diag" in s]


def test_no_measurement_kernel_and_the_same_product_kernels(libs):
    from rlshaders_amd import codeid as C
    default, nodiag = libs
    a, b = {}, {}
    for lib, into in ((default, a), (nodiag, b)):
        for elf in C.code_objects(C.fatbin(lib)):
            into.update(C.kernel_images(elf))
    stamped = sorted(k for k in a if "stamped" in k)
    assert len(stamped) == 8 and not [k for k in b if "stamped" in k]           # four BASELINE kernels x EXACT, FAST
    assert sorted(b) == sorted(k for k in a if "stamped" not in k)              # nothing else came or went
    identical = [k for k in b if a[k] == b[k]]
    relocated = [k for k in b if a[k] != b[k]]
    for k in relocated:
        assert a[k][1] == b[k][1], k                                            # same descriptor: registers, LDS, scratch
        assert C.same_code_up_to_relocation(a[k][0], b[k][0]), k
    assert len(identical) >= 200 and len(relocated) <= 40, (len(identical), len(relocated))
    # ... and by hash: every product kernel keeps its kernel_id (address literals blanked, descriptors cut out of the constants), so
    # a counter profile stamped on the default build is accepted beside this library's kernels
    da, db = C.DeviceCode(default), C.DeviceCode(nodiag)
    assert all(da.kernels[k] == db.kernels[k] for k in db.kernels) and da.library_id != db.library_id
    # the four BASELINE kernels: present once each, same length, same descriptor, same instructions up to the relocation amount
    for name in ("ggx_kernel<5, 0, 1>", "sss_kernel<3, 0, 0>", "skin_kernel<0, 1>", "disney_integrate_kernel<1, 0>"):
        frag = C.mangled_fragment(name)
        hit = [k for k in b if frag in k]
        assert len(hit) == 1, name
        ca, cb = a[hit[0]][0], b[hit[0]][0]
        assert len(ca) == len(cb) and C.same_code_up_to_relocation(ca, cb), name
        assert sum(1 for x, y in zip(ca, cb) if x != y) <= 128, name


def test_same_code_up_to_relocation_is_strict():
    from rlshaders_amd.codeid import same_code_up_to_relocation as same
    import struct
    base = struct.pack("<8I", 1, 2, 3, 4, 5, 6, 7, 8)
    assert same(base, base)
    assert same(base, struct.pack("<8I", 1, 2 + 64, 3, 4, 5 + 64, 6, 7, 8))          # two literals, one amount
    assert not same(base, struct.pack("<8I", 1, 2 + 64, 3, 4, 5 + 32, 6, 7, 8))      # two amounts: not a relocation
    assert not same(base, base + b"\0\0\0\0") and not same(base, base[:-4])


def test_the_python_binding_loads_it_and_says_what_is_missing(libs):
    import sys
    from pathlib import Path
    default, nodiag = libs
    root = Path(__file__).resolve().parent.parent
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import rlshaders_amd as R\n"
            "lib = R.load()\n"
            "assert lib.rls_version() and not hasattr(lib, 'rls_diag_clock_stamps_begin')\n"
            "print('loaded')\n" % str(root))
    import os
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, RLSHADERS_AMD_LIB=str(nodiag)))
    assert p.returncode == 0 and "loaded" in p.stdout, p.stderr[-1500:]
