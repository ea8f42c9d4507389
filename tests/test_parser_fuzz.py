"""Corrupted model files must fail with an error, never crash or hang: `infera_load_model` takes paths from SQL, so
the protobuf wire decoder and the lowering see whatever bytes a user points them at (the reference gets the same
guarantee from tract's prost decoder, engine.rs:49-56).  Mutations run in a child process so a crash is observable."""
from __future__ import annotations

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, random
sys.path.insert(0, %(root)r)
import numpy as np
from infera_amd import capi, onnx_writer as W
from oracle import oracle

seeds = [W.linear_dyn(), W.mlp((6, 5, 3), final_softmax=True), W.resnet18(in_hw=32), W.unary_zoo(8), W.exporter_reshape(),
         W.sklearn_pipeline(9, 3), W.se_net()]
rng = random.Random(%(seed)d)
d = %(tmp)r
loaded = failed = 0
for it in range(%(iters)d):
    blob = bytearray(rng.choice(seeds))
    kind = rng.randrange(6)
    if kind == 0:    # truncate
        blob = blob[: rng.randrange(len(blob))]
    elif kind == 1:  # flip bits
        for _ in range(rng.randrange(1, 8)):
            blob[rng.randrange(len(blob))] ^= 1 << rng.randrange(8)
    elif kind == 2:  # overwrite a run with random bytes
        a = rng.randrange(len(blob)); n = rng.randrange(1, 64)
        blob[a:a + n] = bytes(rng.randrange(256) for _ in range(n))
    elif kind == 3:  # huge varints: lengths / dims that point far outside the file
        a = rng.randrange(len(blob))
        blob[a:a + 1] = b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01"
    elif kind == 4:  # delete a run
        a = rng.randrange(len(blob)); n = rng.randrange(1, 200)
        del blob[a:a + n]
    else:            # duplicate a run
        a = rng.randrange(len(blob)); n = rng.randrange(1, 200)
        blob[a:a] = blob[a:a + n]
    p = os.path.join(d, "m%%d.onnx" %% (it %% 8))
    with open(p, "wb") as f:
        f.write(bytes(blob))
    for loader in ("product", "oracle"):
        try:
            if loader == "product":
                capi.load_model("fz", p)
                capi.get_model_info("fz")
                capi.unload_model("fz")
            else:
                oracle.Model(p)
            loaded += 1
        except (capi.InferaError, oracle.OracleError):
            failed += 1
print("OK", loaded, failed)
"""


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_corrupted_models_fail_cleanly(built, tmp_path, seed):
    code = CHILD % {"root": ROOT, "seed": seed, "tmp": str(tmp_path), "iters": 250}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, f"child died (rc={out.returncode}): {out.stderr[-3000:]}"
    last = out.stdout.strip().splitlines()[-1].split()
    assert last[0] == "OK" and int(last[2]) > 100, out.stdout[-500:]  # most mutations are rejected, some still load
