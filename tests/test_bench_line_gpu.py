"""bench.py's default invocation (what the driver runs: no flags beyond --steps / --warmup) on one GPU.  The ONE stdout line must fit the
driver's capture window (<= 4 KB: round 3's 20.9 KB line parsed as null) and carry the contract fields + roofline + cpu_baseline (three legs:
reference-shaped port, register-blocked port, torch-CPU) + end_to_end + the other two GPU configs of BASELINE.json; the full object goes to
--detail / stderr."""
import pytest


@pytest.mark.gpu
def test_default_bench_line_is_small_and_has_contract_fields_and_other_workloads(default_bench_run):
    d, full = default_bench_run  # (conftest: ONE line, <= 4096 bytes, asserted there)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["unit"] == "rows/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("C2:") and d["config"]["rows_per_gpu"] == 10_000_000 and d["value_is"] == "device_resident"
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.0 < r["frac"] < 1.0
    assert "traffic" in r and r["kernel_ms"] > 0 and r["algorithmic_bytes"] == 5.16e9
    assert abs(d["value"] - 10_000_000 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-5
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] == c["threads"] >= c["cpus"] >= 1 and "sample" in c  # threads used, and the CPUs they can occupy
    b, t = c["best_cpu"], c["torch_cpu"]
    assert b["value"] > 0 and b["gflops_per_cpu"] > c["gflops_per_cpu"] > 0
    assert "error" not in t and t["value"] > 0 and t["gflops_per_cpu"] > 0, t
    e = d["end_to_end"]
    assert e["rows_per_s"] > 0 and 0 < e["frac_of_pcie"] < 1.2 and d["value_end_to_end"] == pytest.approx(e["rows_per_s"], rel=1e-4)
    assert e["callers_per_gpu"] == e["threads"] >= 1 and e["host_read_gbs"] == pytest.approx(e["rows_per_s"] * 512 / 1e9, rel=1e-3)
    best = max(c["value"], b["value"], t["value"])
    assert abs(e["vs_cpu_baseline"] - e["rows_per_s"] / best) / e["vs_cpu_baseline"] < 1e-3  # against the FASTEST of the three CPU legs
    assert e["vs_cpu_reference_shaped"] >= e["vs_cpu_baseline"]
    assert 5 < e["cpu_us_per_chunk"] < 2000 and 0 < e["predicted_scaling_at_8_gpus"] <= 8.0 and e["cpus_needed_for_6x"] > 0
    # round 6: the SECOND bound of the 8-GPU model (what the quota's CPUs gather per second with no GPU behind them), and which bound binds
    assert e["host_copy_gbs_at_threads"] > 1.0 and e["binding_bound"] in ("cpu time per chunk", "host copy rate", "8 x the 1-GPU rate")
    bg = e["bounds_in_gpus"]
    assert e["predicted_scaling_at_8_gpus"] == pytest.approx(min(8.0, bg["cpu_time"], bg["host_copy"]), rel=2e-2)
    g = d["end_to_end_registered"]
    assert "error" not in g and g["rows_per_s"] > 0 and g["zero_copy_calls"] > 0 and g["cpu_us_per_chunk"] > 0
    # round 6: the registered scan on DuckDB's block layout (128 unrelated 256 KiB blocks per chunk: the pulling kernel only) at 2 / 4 / 8 callers,
    # [M rows/s, CPU us per chunk]; the 8-GPU prediction of the opt-in path is read from THIS row
    blk = g["duckdb_blocks"]
    assert "error" not in blk and all(blk[t][0] > 1.0 and 1.0 < blk[t][1] < 500 for t in ("2", "4", "8")), blk
    assert all(blk["in_order"][t][0] > 1.0 for t in ("4", "8")), blk  # a row group's blocks allocated back to back: the arena puts them at one stride
    assert g["prediction_from"] == "duckdb_blocks" and g["predicted_8_gpus_vs_staged_1_gpu"] > 0
    fb = full["end_to_end_registered"]["duckdb_blocks"]
    assert fb["blocks"] > 20_000 and fb["assembled_chunks_per_scan_set"] > 0 and fb["predicted_at_callers_per_gpu"] in (2, 4, 8)
    assert full["end_to_end_registered"]["predicted_8_gpus_vs_staged_1_gpu"] == pytest.approx(fb["predicted_rows_per_s_at_8_gpus"] / e["rows_per_s"], rel=1e-3)
    o = d["other_workloads"]
    c4, c5, c5f = o["C4"], o["C5"], o["C5_fp32"]
    assert c4["roofline"]["bound"] == "hbm" and c4["rows"] == 50_000_000 and 0.0 < c4["roofline"]["frac"] < 1.0, c4
    # C5 on the default plan: the stem and the tiled convolutions on the bf16 matrix cores (three exact parts per operand, six MFMAs per
    # product), priced against the dense bf16 peak / 6; the exact-fp32 plan beside it (C5_fp32) against the fp32 MFMA peak
    assert c5["roofline"]["bound"] == "mfma" and c5["rows"] == 1024 and abs(c5["roofline"]["peak"] - 2500.0 / 6.0) < 0.01 and 0.0 < c5["roofline"]["frac"] < 1.0, c5
    assert c5["dtype"] == "bf16x6" and c5["speedup_over_fp32"] > 0
    assert c5f["dtype"] == "f32" and 0.0 < c5f["roofline"]["frac"] < 1.0 and c5f["roofline"]["peak"] == 157.3, c5f
    assert abs(c4["rows_per_s"] - c4["rows"] / (c4["ms_per_pass"] / 1e3)) / c4["rows_per_s"] < 1e-3
    for w in (c4, c5):  # ... end to end beside their three CPU legs
        assert w["end_to_end"]["rows_per_s"] > 0 and w["end_to_end"]["vs_cpu_baseline"] > 0
        assert w["cpu"]["port"] > 0 and w["cpu"]["best_port"] > 0 and w["cpu"]["torch"] > 0, w["cpu"]
    # the full object: kernels by name, the phase breakdown, the prose
    fo = full["other_workloads"]
    assert "conv_split_bf16x6" in fo["C5"]["kernel"] and "conv_tiled_cq" in fo["C5_fp32"]["kernel"] and fo["C4"]["passes_timed"] >= 20
    assert full["value_is"].startswith("device_resident") and "us_per_chunk_per_thread" in full["end_to_end"]
    assert full["cpu_baseline"]["host_cpu"]["fma_peak_gflops_per_cpu"] > 0


@pytest.mark.gpu
@pytest.mark.perf
def test_default_bench_rates_are_where_they_were_measured(default_bench_run):
    """The rate thresholds of the same run (collected LAST, tests/conftest.py: a noisy box must not stop `-m gpu -x` before the parity files).
    Loose lower bounds around what five rounds of boxes measured: C2 0.92 of the fp32 MFMA peak, C4 0.63-0.69 of 8 TB/s, C5 0.49 of bf16 / 6 and
    1.55x its exact-fp32 plan, C5 end to end 0.91-0.93 of its resident rate, C2 end to end 0.87 of the link."""
    d, _ = default_bench_run
    o = d["other_workloads"]
    assert d["roofline"]["frac"] > 0.5 and o["C4"]["roofline"]["frac"] > 0.3 and o["C5"]["roofline"]["frac"] > 0.2 and o["C5_fp32"]["roofline"]["frac"] > 0.3
    assert o["C5"]["speedup_over_fp32"] > 1.15
    assert o["C5"]["end_to_end"]["rows_per_s"] > 0.5 * o["C5"]["rows_per_s"]  # C5 stays kernel-bound end to end
    assert 0.5 < d["end_to_end"]["frac_of_pcie"] < 1.0  # 0.87 of the link's raw 64 GB/s on every box so far
