#!/usr/bin/env python3
"""Board power and clocks while a BASELINE kernel runs back to back: is the clock the kernel holds (in-kernel stamps:
profiles/r06_*_clock.json) the power limit at work?  For each workload: launch the kernel continuously for `--seconds` while a
thread samples the hwmon files of the device (power1_average / power1_cap, freq1_input = sclk) -- read-only sysfs, no
privileges -- then read the in-kernel clock of the last launches (rls_diag_clock_stamps_*).
usage: tools/power_clock_trace.py [--seconds 6] [--workloads ggx_reflect_refract,sss_probe,skin,disney_integrate] [--math exact]
Prints one JSON object per workload (and an idle sample first)."""
import argparse
import glob
import json
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def hwmon_dirs(pci=None):
    """hwmon directories of the amdgpu cards; with `pci` ("0000:75:00.0") only the one of that PCI function (a box shows the
    hwmon files of every card of its node in sysfs, whichever single GPU HIP is allowed to see)"""
    dirs = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    if pci:
        import os
        dirs = [d for d in dirs if pci.lower() in os.path.realpath(os.path.dirname(os.path.dirname(d))).lower()]
    return dirs


def read_int(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None


def sample(d):
    out = {}
    for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
        v = read_int(f"{d}/{name}")
        if v is not None:
            out[name] = v
    return out


class Sampler(threading.Thread):
    def __init__(self, d, period=0.1):
        super().__init__(daemon=True)
        self.d, self.period, self.rows, self.stop = d, period, [], False

    def run(self):
        while not self.stop:
            self.rows.append(sample(self.d))
            time.sleep(self.period)


def summarize(rows):
    out = {"samples": len(rows)}
    for k in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
        v = sorted(r[k] for r in rows if k in r)
        if v:
            scale = 1e6 if k.startswith("power") else (1e6 if k.startswith("freq") else 1e3)      # uW -> W, Hz -> MHz, mC -> C
            out[k] = {"median": round(v[len(v) // 2] / scale, 1), "max": round(v[-1] / scale, 1), "min": round(v[0] / scale, 1)}
    cap = [r["power1_cap"] for r in rows if "power1_cap" in r]
    if cap:
        out["power1_cap_w"] = cap[0] / 1e6
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--workloads", default="ggx_reflect_refract,sss_probe,skin,disney_integrate")
    ap.add_argument("--math", default="exact")
    a = ap.parse_args()
    import torch
    import rlshaders_amd as R
    from bench_workloads import make_workload
    pr = torch.cuda.get_device_properties(0)
    pci = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    dirs = hwmon_dirs(pci)
    print(json.dumps({"device": pr.name, "pci": pci, "hwmon": dirs,
                      "files": sorted(p.name for p in Path(dirs[0]).iterdir() if p.is_file()) if dirs else []}))
    if not dirs:
        raise SystemExit("no hwmon directory visible")
    d = dirs[0]
    ctx = R.Context(0)
    ctx.set_math_mode(a.math == "fast")
    s = Sampler(d); s.start(); time.sleep(1.5); s.stop = True; s.join()
    print(json.dumps({"workload": "idle", **summarize(s.rows)}))
    sizes = {"ggx_reflect_refract": 26, "sss_probe": 25, "skin": 27, "disney_integrate": 26}
    for w in a.workloads.split(","):
        wl = make_workload(R, ctx, w, 1 << sizes.get(w, 26), first=0, candidates=1)
        torch.cuda.synchronize()
        s = Sampler(d); s.start()
        t_end = time.perf_counter() + a.seconds
        launches = 0
        while time.perf_counter() < t_end:
            for _ in range(8):
                wl.launch()
            launches += 8
            torch.cuda.synchronize()
        # the second half of the run: the settled state
        rows = s.rows[len(s.rows) // 2:]
        ctx.clock_stamps_begin()
        for _ in range(8):
            wl.launch()
        stamps = ctx.clock_stamps_read()
        ctx.clock_stamps_end()
        s.stop = True; s.join()
        rec = {"workload": w, "math": a.math, "seconds": a.seconds, "launches": launches, **summarize(rows)}
        if len(stamps):
            rec["in_kernel_clock_ghz"] = R.Context.clock_from_stamps(stamps)["effective_clock_ghz"]
        print(json.dumps(rec), flush=True)
        del wl
        torch.cuda.empty_cache()
        time.sleep(1.0)


if __name__ == "__main__":
    main()
