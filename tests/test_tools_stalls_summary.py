"""CPU: tools/summarize_stalls.py -- the arithmetic behind DESIGN.md section 5's "ALU busy" column -- on a synthetic rocprofv3
counter file: dispatches are selected by the EXACT kernel name of the bench line (round 4's file had averaged the EXACT, FAST
and other dispatches of a run), rows of one dispatch add up, the median over dispatches is taken, and the port decomposition
follows the counters' units (SQ_CYCLES per shader engine x 32; quad-cycles; VALU2 = quad-cycles holding two)."""
import csv
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
EXACT = "void (anonymous namespace)::ggx_kernel<5, 0, 1>(rlsh::GgxIO)"
FAST = "void (anonymous namespace)::ggx_kernel<5, 1, 1>(rlsh::GgxIO)"
STAMPED = "void (anonymous namespace)::ggx_kernel_stamped<5, 0, 1>(rlsh::GgxIO, unsigned long long*)"
COLS = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
        "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
        "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def _write(path, rows):
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(COLS)
        for disp, name, counter, value, t0, t1 in rows:
            w.writerow([disp, disp, "Agent 2", 2, 1, 1, 1 << 26, 7, name, 256, 0, 0, 64, 0, 78, counter, float(value), t0, t1])


def run_summary(tmp_path, code=None):
    """synthetic pmc_stalls.sh session under tmp_path -> summarize_stalls.main's record (also used by
    tests/test_profile_binding.py: `code` is the device-code stamp bench.py puts on the session's bench line)"""
    spec = importlib.util.spec_from_file_location("summarize_stalls", ROOT / "tools" / "summarize_stalls.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = tmp_path / "gpurun_out" / "stalls_t_ggx_reflect_refract"
    (tmp_path / "profiles").mkdir()
    n = 1 << 26
    (d).mkdir(parents=True)
    (d / "bench.json").write_text(json.dumps({"record": "headline_detail", "config": {"math": "exact", "points_per_gpu": n},
                                              "roofline": dict({"kernel": "ggx_kernel<5, 0, 1>", "kernel_ms": 2.0},
                                                               **({"code": code} if code else {}))}) + "\n")
    # 1000 shader cycles per SE -> Q = 1000 / 4 * 1024 = 256 000 SIMD quad-cycles; the counter comes as 8 rows (one per XCD)
    rows = []
    for disp, scale in ((1, 5.0), (2, 1.0), (3, 1.0)):          # dispatch 1: a cold outlier the median drops
        for x in range(8):
            rows.append((disp, EXACT, "SQ_CYCLES", 32 * 1000 * scale / 8, 100, 100 + 2_000_000))
    rows += [(9, FAST, "SQ_CYCLES", 1, 0, 10), (10, STAMPED, "SQ_CYCLES", 1, 0, 10)]
    _write(d / "i" / "x" / "1_counter_collection.csv", rows)
    rows = []
    for disp in (1, 2, 3):
        rows += [(disp, EXACT, "SQ_ACTIVE_INST_VALU", 400_000, 0, 1), (disp, EXACT, "SQ_ACTIVE_INST_VALU2", 150_000, 0, 1),
                 (disp, EXACT, "SQ_INSTS_VALU", 380_000, 0, 1)]
    rows += [(7, FAST, "SQ_ACTIVE_INST_VALU", 9e9, 0, 1), (8, STAMPED, "SQ_ACTIVE_INST_VALU2", 9e9, 0, 1)]
    _write(d / "b" / "x" / "2_counter_collection.csv", rows)
    return mod.main(["summarize_stalls.py", "t", "ggx_reflect_refract"], root=str(tmp_path))


def test_by_name_median_and_port_decomposition(tmp_path):
    out = run_summary(tmp_path)
    assert out["code"] is None                                             # a bench line without a stamp: the summary says so
    assert out["kernel"] == "ggx_kernel<5, 0, 1>" and out["dispatches_per_counter"]["SQ_CYCLES"] == 3
    v = out["valu_port"]
    assert v["simd_quad_cycles_per_launch"] == 256_000.0                   # the median dispatch (scale 1.0), its 8 rows added up
    assert abs(v["holding_two"] - 150_000 / 256_000) < 1e-4
    assert abs(v["holding_one"] - (400_000 - 300_000) / 256_000) < 1e-4
    assert abs(v["busy"] - 250_000 / 256_000) < 1e-4 and abs(v["idle"] - 6_000 / 256_000) < 1e-4
    assert abs(v["instructions_issued_in_pairs"] - 300_000 / 380_000) < 1e-4
    saved = json.loads((tmp_path / "profiles" / "t_ggx_reflect_refract_stalls.json").read_text())
    assert saved["valu_port"] == v and saved["other_kernels_counted"] == 0


def test_clock_record_grbm_and_stamps(tmp_path):
    """tools/summarize_workload.py clock_record: GRBM_GUI_ACTIVE rows of one dispatch add up (rocprofv3 may split them by XCD),
    / 8 / the dispatch's own duration, by exact kernel name; the sustained stamps of the bench line take precedence"""
    spec = importlib.util.spec_from_file_location("summarize_workload", ROOT / "tools" / "summarize_workload.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src, dst = tmp_path / "prof", tmp_path / "profiles"
    dst.mkdir()
    rows = []
    for disp in (1, 2, 3):
        for x in range(8):                                   # 8 ms dispatch at 1.9 GHz: 1.52e7 cycles per XCD
            rows.append((disp, EXACT, "GRBM_GUI_ACTIVE", 1.9e9 * 8e-3, 1_000_000, 1_000_000 + 8_000_000))
    rows.append((4, FAST, "GRBM_GUI_ACTIVE", 8 * 2.4e9 * 8e-3, 0, 8_000_000))
    rows.append((5, STAMPED, "GRBM_GUI_ACTIVE", 8 * 1.0e9 * 8e-3, 0, 8_000_000))
    _write(src / "clock_grbm" / "x" / "1_counter_collection.csv", rows)
    mod.clock_record(str(src), str(dst), "t", "ggx_reflect_refract", "ggx_kernel<5, 0, 1>", "exact")
    rec = json.loads((dst / "t_ggx_reflect_refract_clock.json").read_text())
    assert abs(rec["grbm_clock_ghz"] - 1.9) < 1e-3 and rec["grbm_dispatches"] == 3 and abs(rec["grbm_dispatch_ms"] - 8.0) < 1e-6
    assert rec["effective_clock_ghz"] == rec["grbm_clock_ghz"] and "sustained_clock_ghz" not in rec
    (src / "clock_sustained.json").write_text(json.dumps({"record": "headline_detail", "roofline": {
        "kernel_ms": 2.0, "clock": {"effective_clock_ghz": 1.876, "workgroups": 262144}}}) + "\n")
    mod.clock_record(str(src), str(dst), "t", "ggx_reflect_refract", "ggx_kernel<5, 0, 1>", "exact")
    rec = json.loads((dst / "t_ggx_reflect_refract_clock.json").read_text())
    assert rec["sustained_clock_ghz"] == 1.876 and rec["effective_clock_ghz"] == 1.876 and abs(rec["grbm_clock_ghz"] - 1.9) < 1e-3
