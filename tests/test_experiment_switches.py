"""CPU: the compile-time experiment switches of the kernels (what tools/ab.sh flips to build A/B variants: every "-x %" of
DESIGN.md was measured that way) still compile.  ADVICE r4: switches nobody builds rot.  Two variant builds flip every switch
away from its default between them (front end + template instantiation of every closure unit, device and host pass,
`-fsyntax-only`: seconds); the off-by-default code blocks that were flagged (two tiles per lane, refined FAST) are gone."""
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "rlshaders_amd" / "csrc"
UNITS = ["ggx", "disney", "sss", "skin", "integrate", "lights", "scatter", "shade", "alternates"]

# every switch with a value that is not its default; ATAN_SELECTS / ANGLE_SELECTS are alternatives of one #if chain
GROUP_A = {"RLS_DISNEY_RELOAD": 0, "RLS_GGX_RELOAD": 0, "RLS_INT_RELOAD": 0, "RLS_RELOAD_ARGS": 0, "RLS_NO_STREAMED": 1,
           "RLS_NO_XCD_TILES": 1, "RLS_CAP_MULT": 1, "RLS_NO_WAVE_UNIFORM": 1, "RLS_NO_SADDR": 1, "RLS_AT_SCALAR_BARRIER": 0,
           "RLS_ATAN_SELECTS": 1, "RLS_POW5_GENERAL": 1, "RLS_LOOP_RECIP": 0, "RLS_DISNEY_D_RECIP": 0, "RLS_ND_RECIP_D": 0,
           "RLS_ND_PP_RANGE_ONCE": 0, "RLS_ND_PROFILE_WINDOWED": 0, "RLS_ND_MAKE_RANGE_ONCE": 0, "RLS_SQRT_NO_FALLBACK": 1,
           "RLS_NO_FAST_RCP": 1, "RLS_SKIN_SGPR": 0, "RLS_SSS_UNIFORM_SGPR": 0, "RLS_SPEC_BLOCK": 2, "RLS_INT_WAVES": 3,
           "RLS_WAVES_PER_EU": 5, "RLS_SKIN_WAVES": 5, "RLS_FAST_VIEW_Z_AS_REFERENCE": 0, "RLS_MEM_POLICY": 3}
GROUP_B = {"RLS_ANGLE_SELECTS": 1, "RLS_LOAD_RENEW": 1, "RLS_ND_ONE_SAMPLE_RECIP": 1, "RLS_SKIN_ND_RECIP": 1,
           "RLS_NO_PAIR_COMPACTION": 1, "RLS_ND_DIVC_UNGUARDED": 1, "RLS_ND_MERGE_RADIUS": 1, "RLS_ND_RADIUS_SELECTS": 1,
           "RLS_ND_PDF3_MERGED": 1, "RLS_PROBE_SELECTS": 1, "RLS_DISNEY_LIGHT_WAVES": 4, "RLS_SPEC_BLOCK": 8}


def _syntax_only(job):
    unit, fast, group = job
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-gpu-rdc", f"-DRLS_FAST={fast}",
           *[f"-D{k}={v}" for k, v in group.items()], "-fsyntax-only", str(CSRC / f"{unit}.hip")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    return unit, fast, p.returncode, [l for l in p.stderr.splitlines() if "error" in l][:3]


def test_every_switch_still_compiles():
    jobs = [(u, f, g) for g in (GROUP_A, GROUP_B) for u in UNITS for f in (0, 1)]
    with ThreadPoolExecutor(8) as ex:
        bad = [r for r in ex.map(_syntax_only, jobs) if r[2] != 0]
    assert not bad, bad


def test_the_groups_cover_the_switches_in_the_sources():
    """a switch added to the kernels must be added here (or be one of the structural constants)"""
    import re
    structural = {"RLS_FAST", "RLS_BLOCK", "RLS_DEV", "RLS_HIDDEN", "RLS_GGX_WAVES", "RLS_SSS_WAVES", "RLS_DISNEY_WAVES",
                  "RLS_HOIST_TILES_PER_THREAD", "RLS_HOIST_MIN_BLOCKS_PER_CU"}      # (the *_WAVES(OP) macros take an argument)
    found = set()
    for f in list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")):
        for m in re.finditer(r"^\s*#\s*(?:ifndef|ifdef|if|elif)\s+(?:!\s*)?(?:defined\s*\(\s*)?(RLS_[A-Z0-9_]+)", f.read_text(), re.M):
            found.add(m.group(1))
        for m in re.finditer(r"defined\s*\(\s*(RLS_[A-Z0-9_]+)\s*\)", f.read_text()):
            found.add(m.group(1))
    missing = found - structural - set(GROUP_A) - set(GROUP_B)
    assert not missing, missing
