"""Stale-constant guard for bench.py's `roofline.traffic`: the HBM bytes per launch come from committed
rocprofv3 PMC passes (profiles/traffic_<workload>.json), not from the run itself -- so each file must name the kernel that
`infera_hip_get_plan` reports for that workload TODAY, and the row count bench.py uses.  If a kernel is renamed or the schedule
picks another one, this test fails until the PMC passes are re-collected (tools/profile_bench.sh).  Runs without a GPU: the plan
is made at infera_load_model."""
import json
import os

import pytest

from infera_amd import capi
from infera_amd import onnx_writer as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(tmp_path, name, blob):
    path = W.write(str(tmp_path / f"{name}.onnx"), blob)
    capi.load_model("traffic_" + name, path)
    try:
        return capi.get_plan("traffic_" + name)
    finally:
        capi.unload_model("traffic_" + name)


def test_traffic_mlp_names_the_planned_kernel(tmp_path):
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_mlp.json")))
    plan = _plan(tmp_path, "mlp", W.mlp((128, 256, 64, 1)))
    assert t["rows"] == 10_000_000 and t["workload"] == "mlp"
    assert t["kernel"] in plan["fused_kernel"], (t["kernel"], plan["fused_kernel"])
    assert abs(t["traffic_bytes_per_launch"] / (516.0 * t["rows"]) - 1.0) < 0.05  # one pass over the table, results once


def test_traffic_logreg_names_the_planned_kernel(tmp_path):
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_logreg.json")))
    plan = _plan(tmp_path, "logreg", W.logreg_softmax(128, 10))
    assert t["rows"] == 50_000_000 and t["workload"] == "logreg"
    assert plan["exec"][0] == "dense_softmax" and len(plan["dense_kernels"]) == 1
    assert plan["dense_kernels"][0].startswith(t["kernel"]), (t["kernel"], plan["dense_kernels"])
    assert abs(t["traffic_bytes_per_launch"] / (552.0 * t["rows"]) - 1.0) < 0.05


def test_traffic_resnet18_lists_the_planned_kernel_families(tmp_path):
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_resnet18.json")))
    plan = _plan(tmp_path, "resnet18", W.resnet18(in_hw=224))
    assert t["rows"] == 1024
    kinds = set(plan["exec"]) - {"skipped"}
    named = " ".join(t["read_bytes_by_kernel"].keys())
    # (default plan since round 3: the stem + max-pool and the tiled convolutions in bf16 x three exact parts)
    want = {"conv_patch_pool_bf16x6": "conv2d_stem_split6_kernel", "conv_split_bf16x6": "conv2d_split6_kernel"}
    for kind, stem in want.items():
        assert kind in kinds, kinds
        assert stem in named, (stem, named[:300])
