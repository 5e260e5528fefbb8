#!/usr/bin/env python3
"""Stamp committed counter profiles with the identity of the device code they were taken on.

    tools/stamp_profiles.py <tag> <commit> [suffix,suffix...]
        e.g.  tools/stamp_profiles.py r05 87b6375 traffic,flops,clock      (round 5, session 1 ran on that tree)
              tools/stamp_profiles.py r05 ebdce91 stalls                   (session 2)

Profiles written from round 6 on carry `code` = {library_id, unit_id, kernel} natively: bench.py computes the ids on the GPU
box for the library it has loaded and the summarisers (tools/summarize_workload.py, summarize_stalls.py) copy them out of
the bench line of the profiled session.  Older profiles carry none, and bench.py treats an unstamped profile as stale.  This
tool supplies the stamp for them WITHOUT guessing: it checks out <commit> (the tree the passes ran on) into a scratch git
worktree, builds the library there from clean (rlshaders_amd/build.py; hipcc cross-compiles gfx950 without a GPU), reads the
ids out of that build (rlshaders_amd/codeid.py) and writes them into every <tag>_*_{traffic,flops,clock,stalls}.json under
profiles/ and profiles/archive/
together with how they were obtained.  The build is deterministic: the same sources and flags give the same code objects, so
the ids are those of the library that ran.

No GPU needed; takes about a minute.
"""
import json
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SUFFIXES = ("traffic", "flops", "clock", "stalls")


def build_commit(commit: str) -> Path:
    """clean build of `commit` in a scratch worktree -> path of its library (the worktree is left for the caller to remove)"""
    tmp = Path(tempfile.mkdtemp(prefix="rls_stamp_"))
    wt = tmp / "tree"
    subprocess.run(["git", "-C", str(ROOT), "worktree", "add", "--detach", "-f", str(wt), commit], check=True,
                   capture_output=True)
    p = subprocess.run([sys.executable, "-m", "rlshaders_amd.build", "--force"], cwd=str(wt), capture_output=True, text=True)
    if p.returncode != 0:
        raise SystemExit(f"build of {commit} failed:\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}")
    return wt / "rlshaders_amd" / "lib" / "librlshaders_amd.so"


def main():
    if len(sys.argv) not in (3, 4):
        raise SystemExit(__doc__)
    tag, commit = sys.argv[1], sys.argv[2]
    suffixes = tuple(sys.argv[3].split(",")) if len(sys.argv) == 4 else SUFFIXES
    if not set(suffixes) <= set(SUFFIXES):
        raise SystemExit(f"suffixes are {SUFFIXES}")
    from rlshaders_amd.codeid import DeviceCode
    full = subprocess.run(["git", "-C", str(ROOT), "rev-parse", commit], check=True, capture_output=True, text=True).stdout.strip()
    lib = build_commit(commit)
    try:
        dc = DeviceCode(lib)
        done = []
        for suffix in suffixes:
            for p in sorted(list((ROOT / "profiles").glob(f"{tag}_*_{suffix}.json")) +
                            list((ROOT / "profiles" / "archive").glob(f"{tag}_*_{suffix}.json"))):
                d = json.loads(p.read_text())
                kernel = d.get("kernel")
                if not kernel:
                    continue
                rec = dc.record(kernel)
                if rec.get("unit_id") is None or rec.get("kernel_id") is None:
                    raise SystemExit(f"{p.name}: kernel {kernel!r} is in no code object of the build of {commit}")
                rec["stamped"] = (f"tools/stamp_profiles.py {tag} {commit}: ids of a clean build of commit {full[:12]} (the tree "
                                  "these passes ran on), not recorded at collection time")
                # a stamp this tool wrote may be rewritten (the id definitions were refined during round 6); one recorded at
                # collection time on the GPU box is never touched
                if d.get("code") and not d["code"].get("stamped"):
                    raise SystemExit(f"{p.name} carries a stamp recorded at collection time: {d['code']}")
                # keep the key order readable: code right behind kernel
                out = {}
                for k, v in d.items():
                    if k == "code":
                        continue
                    out[k] = v
                    if k == "kernel":
                        out["code"] = rec
                p.write_text(json.dumps(out, indent=1))
                done.append((p.name, rec["kernel_id"]))
        print(json.dumps({"tag": tag, "commit": full, "library_id": dc.library_id, "stamped": done}, indent=1))
    finally:
        subprocess.run(["git", "-C", str(ROOT), "worktree", "remove", "--force", str(lib.parents[2])], capture_output=True)
        subprocess.run(["git", "-C", str(ROOT), "worktree", "prune"], capture_output=True)
        shutil.rmtree(lib.parents[3], ignore_errors=True)


if __name__ == "__main__":
    main()
