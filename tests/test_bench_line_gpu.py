"""bench.py's default invocation (what the driver runs: no flags beyond --steps / --warmup) on one GPU: the contract fields, the
roofline / cpu_baseline / end_to_end blocks, and the short device-resident C4 / C5 measurements under `other_workloads` (same
definitions as `value` / `roofline`) -- so that the driver-recorded line carries all three GPU configs of BASELINE.json."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_default_bench_line_has_contract_fields_and_other_workloads():
    env = dict(os.environ)
    env.pop("INFERA_DEVICES", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "3", "--e2e-reps", "2"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["unit"] == "rows/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("C2:") and d["config"]["rows_per_gpu"] == 10_000_000
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.5 < r["frac"] < 1.0
    assert abs(d["value"] - 10_000_000 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    e = d["end_to_end"]
    assert e["rows_per_s"] > 0 and 0 < e["frac_of_pcie"] < 1.2 and "vs_cpu_baseline" in e
    o = d["other_workloads"]
    c4, c5 = o["C4"], o["C5"]
    assert c4["roofline"]["bound"] == "hbm" and c4["rows"] == 50_000_000 and 0.3 < c4["roofline"]["frac"] < 1.0, c4
    # C5 on the default plan: the tiled convolutions on the bf16 matrix cores (three exact parts per operand, six MFMAs per product), priced
    # against the dense bf16 peak / 6; the exact-fp32 plan beside it (C5_fp32) against the fp32 MFMA peak
    assert c5["roofline"]["bound"] == "mfma" and c5["rows"] == 1024 and c5["roofline"]["peak"] == 2500.0 / 6.0 and 0.2 < c5["roofline"]["frac"] < 1.0, c5
    assert c5["dtype"].startswith("bf16x6") and "conv_split_bf16x6" in c5["kernel"] and c5["speedup_over_fp32"] > 1.15
    c5f = o["C5_fp32"]
    assert c5f["dtype"] == "f32" and "conv_tiled_cq" in c5f["kernel"] and 0.3 < c5f["roofline"]["frac"] < 1.0 and c5f["roofline"]["peak"] == 157.3, c5f
    assert abs(c4["rows_per_s"] - c4["rows"] / (c4["ms_per_pass"] / 1e3)) / c4["rows_per_s"] < 1e-9
    assert c4["passes_timed"] >= 20
    # round 3: `value` says what it is; the CPU baseline carries a best-CPU leg and ratios are taken against the faster one
    assert d["value_is"].startswith("device_resident")
    b = c["best_cpu"]
    assert b["value"] > 0 and b["gflops_per_cpu"] > c["gflops_per_cpu"] > 0 and c["host_cpu"]["fma_peak_gflops_per_cpu"] > 0
    assert abs(e["vs_cpu_baseline"] - e["rows_per_s"] / max(c["value"], b["value"])) / e["vs_cpu_baseline"] < 1e-9
    assert e["vs_cpu_reference_shaped"] >= e["vs_cpu_baseline"]
    # ... the host CPU cost per chunk, the 8-GPU prediction it implies and the CPU count that 6x would need
    h = e["host_cpu_cost"]
    assert 5 < h["cpu_us_per_chunk"] < 2000 and abs(h["rows_per_cpu_second"] - 2048e6 / h["cpu_us_per_chunk"]) / h["rows_per_cpu_second"] < 1e-6
    assert h["predicted_rows_per_s_at_8_gpus"] <= 8 * e["rows_per_s"] * (1 + 1e-9) and h["cpus_needed_for_6x"] > 0
    # ... the link-elided 8-slot probe of the host side (child process)
    pr = e["host_ceiling_probe"]
    assert "error" not in pr, pr
    assert pr["device_slots"] == 8 and pr["rows_per_s"] > 0 and pr["host_cpu_cost"]["cpu_us_per_chunk"] > 0
    # ... and C4 / C5 end to end beside their CPU baselines in the same line
    for w in (c4, c5):
        assert w["end_to_end"]["rows_per_s"] > 0 and w["cpu_baseline"]["value"] > 0 and w["cpu_baseline"]["best_cpu"]["value"] > 0, w.keys()
        assert w["end_to_end"]["vs_cpu_baseline"] > 0
    assert c5["end_to_end"]["rows_per_s"] > 0.5 * c5["rows_per_s"]  # C5 stays kernel-bound end to end
    # ... and C5 in the opt-in split-fp16 convolution mode: its own ceiling (dense fp16 / 3), well ahead of the exact-fp32 kernels
    s5 = o["C5_f16x3"]
    assert "error" not in s5, s5
    assert s5["dtype"] == "f16x3" and "conv_split_f16x3" in s5["kernel"] and s5["roofline"]["peak"] == 2500.0 / 3.0 and 0.1 < s5["roofline"]["frac"] < 1.0
    assert s5["speedup_over_fp32"] > 1.3 and s5["speedup_over_fp32"] > c5["speedup_over_fp32"] and "conv_split_f16x3" not in c5["kernel"]
    assert s5["end_to_end"]["rows_per_s"] > 1.1 * c5["end_to_end"]["rows_per_s"] and s5["end_to_end"]["vs_cpu_baseline"] > c5["end_to_end"]["vs_cpu_baseline"]
