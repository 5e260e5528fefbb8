"""Per-kernel instruction statistics of a built object: tools/isa_stats.py <object.o> [kernel-name filter]
(static counts over the whole kernel body: VALU / SALU / VMEM / LDS, spills as v_readlane / v_writelane / scratch_)"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"


def device_code(obj):
    d = tempfile.mkdtemp()
    tmp = os.path.join(d, os.path.basename(obj))
    subprocess.run(["cp", obj, tmp], check=True)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, capture_output=True)
    co = [f for f in os.listdir(d) if "amdgcn" in f]
    return os.path.join(d, co[0])


def stats(obj, flt=""):
    co = device_code(obj)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    dem = {}
    out = collections.OrderedDict()
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            if name not in dem:
                dem[name] = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = dem[name]
            out.setdefault(cur, collections.Counter())
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if cur and m:
            op = m.group(1)
            c = out[cur]
            c["total"] += 1
            if op.startswith("v_readlane") or op.startswith("v_writelane"): c["lane_spill"] += 1
            if op.startswith("scratch_"): c["scratch"] += 1
            if op.startswith("v_"): c["valu"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): c["vmem"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
    # register counts from the notes
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    regs = {}
    for blk in notes.split("- .agpr_count")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm: continue
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, 0])[1]
        regs[nm.group(1)] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                                 scratch=g("private_segment_fixed_size"))
    for mangled, d in dem.items():
        if flt and flt not in d: continue
        if mangled.endswith(".kd"): continue
        c = out[d]
        r = regs.get(mangled, {})
        print(f"{d[:100]}\n    total {c['total']} valu {c['valu']} salu {c['salu']} vmem {c['vmem']} lds {c['lds']} lane-spill {c['lane_spill']} "
              f"scratch-ops {c['scratch']}  regs {r}")


if __name__ == "__main__":
    stats(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
