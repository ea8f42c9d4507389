"""INFERA_PROFILE (hip/profile.{hpp,cpp}; SURVEY.md 5 "tracing"): 1 = roctx ranges around every stage of a host-ABI call + the per-stage exit
report; 2 (round 6) = the fine-grained SECTION clocks of a call's CPU work (no roctx ranges) + the same exit report.  Both must leave results
untouched and print their report at process exit; unset, nothing is printed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer, synth
path = onnx_writer.write(os.path.join(%(tmp)r, "mlp.onnx"), onnx_writer.mlp((128, 256, 64, 1)))
capi.load_model("p", path)
x = synth.table(3, 0, 2048, 128)
h = hashlib.sha256()
for _ in range(20):
    h.update(capi.predict_columns("p", [np.ascontiguousarray(x[:, c]) for c in range(128)]).tobytes())
print("RESULT " + h.hexdigest())
"""


def run(tmp_path, mode):
    env = dict(os.environ)
    env.pop("INFERA_PROFILE", None)
    if mode:
        env["INFERA_PROFILE"] = mode
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tmp": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1], p.stderr


def test_profile_modes_report_and_do_not_change_results(built, tmp_path):
    plain, err0 = run(tmp_path, None)
    stages, err1 = run(tmp_path, "1")
    sections, err2 = run(tmp_path, "2")
    assert plain == stages == sections
    assert "[infera profile]" not in err0
    assert "host-ABI calls 20" in err1 and "enqueue" in err1 and "sections, ns per host-ABI call" not in err1
    assert "host-ABI calls 20" in err2 and "sections, ns per host-ABI call (20 calls" in err2
    for name in ("capi: registry read + validation", "enqueue: H2D copy", "enqueue: model launch", "wait: event record", "copy out"):
        assert name in err2, (name, err2[-2000:])
