mkdir -p gpurun_out
python3 tools/power_clock_trace.py --seconds 6 > gpurun_out/r05_power_clock_exact.jsonl 2> gpurun_out/r05_power_clock_exact.err; cut -c1-600 gpurun_out/r05_power_clock_exact.jsonl; tail -3 gpurun_out/r05_power_clock_exact.err
python3 tools/power_clock_trace.py --seconds 6 --math fast --workloads ggx_reflect_refract,sss_probe > gpurun_out/r05_power_clock_fast.jsonl 2>/dev/null; cut -c1-600 gpurun_out/r05_power_clock_fast.jsonl
