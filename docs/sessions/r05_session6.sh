#!/bin/bash
# round 5, GPU session 6: board power and clocks (read-only hwmon / rocm-smi) while each BASELINE kernel runs back to back --
# evidence for or against "the pointwise kernels are held at 1.8-1.95 GHz by the power limit" -- and the driver's own
# bench command (--steps 20 --warmup 5) on the final library
mkdir -p gpurun_out
ls /sys/class/drm/ 2>&1 | head; ls /sys/class/drm/card*/device/hwmon/ 2>&1 | head
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 > gpurun_out/r05_rocm_smi_idle.txt; head -30 gpurun_out/r05_rocm_smi_idle.txt
python3 tools/power_clock_trace.py --seconds 6 > gpurun_out/r05_power_clock_exact.jsonl 2> gpurun_out/r05_power_clock_exact.err; cat gpurun_out/r05_power_clock_exact.jsonl; tail -3 gpurun_out/r05_power_clock_exact.err
python3 tools/power_clock_trace.py --seconds 6 --math fast --workloads ggx_reflect_refract > gpurun_out/r05_power_clock_fast.jsonl 2>/dev/null; cat gpurun_out/r05_power_clock_fast.jsonl
( python3 -c "
import sys, time, subprocess
sys.path.insert(0, '.')
import torch, rlshaders_amd as R
from bench_workloads import make_workload
ctx = R.Context(0)
wl = make_workload(R, ctx, 'ggx_reflect_refract', 1 << 26, first=0, candidates=1)
t_end = time.time() + 8
import threading
def smi():
    time.sleep(4)
    print(subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout)
th = threading.Thread(target=smi); th.start()
while time.time() < t_end:
    for _ in range(8): wl.launch()
    torch.cuda.synchronize()
th.join()
" ) > gpurun_out/r05_rocm_smi_config2.txt 2>&1; cat gpurun_out/r05_rocm_smi_config2.txt | head -40
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_driver_cmd.out 2> gpurun_out/r05_bench_driver_cmd.err; tail -3 gpurun_out/r05_bench_driver_cmd.err; tail -c 600 gpurun_out/r05_bench_driver_cmd.out
