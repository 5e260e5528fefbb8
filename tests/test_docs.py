"""CPU: DESIGN.md stays auditable (106 KB in 910 lines of up to 700 columns was not) -- at most 300 lines of at
most 120 columns; the experiment diaries live under docs/ -- and every profiles/ file it names exists."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_design_md_is_short_and_narrow():
    lines = (ROOT / "DESIGN.md").read_text(encoding="utf8").split("\n")
    assert len(lines) <= 300, len(lines)
    wide = [(i + 1, len(l)) for i, l in enumerate(lines) if len(l) > 120]
    assert not wide, wide
    for must in ("## 0. Scope", "Closed Arnold services", "byte accounting", "what binds each", "## 7. Multi-GPU", "parity unpinned"):
        assert must.lower() in "\n".join(lines).lower(), must


def test_named_profiles_exist():
    text = (ROOT / "DESIGN.md").read_text(encoding="utf8") + (ROOT / "docs" / "experiments_r04.md").read_text(encoding="utf8")
    missing = []
    for m in set(re.findall(r"`(?:profiles/)?(r0\d_[A-Za-z0-9_.*]+)`", text)):
        name = m.rstrip(".")
        if not list((ROOT / "profiles").glob(name if "*" in name else name + "*")):
            missing.append(name)
    assert not missing, missing
