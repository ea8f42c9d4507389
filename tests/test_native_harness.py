"""the reference's concurrency harness (test/concurrency/test_concurrency.py:25-50, 71-78) as a C++
program against the C ABI -- tests/native/concurrency_harness.cpp, std::threads calling include/infera.h directly, no Python
in the process.  On a box without a GPU the predict call must fail loudly (no CPU fallback) and everything else -- 80
loads / unloads from 8 threads, the -1 of an unknown unload, the empty registry at the end -- is still checked."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def _run(*extra):
    subprocess.run(["make", "-C", NATIVE], check=True, capture_output=True)
    p = subprocess.run([os.path.join(NATIVE, "concurrency_harness"), os.path.join(ROOT, "tests", "golden", "linear.onnx"), *extra],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_native_harness_without_gpu_fails_loudly_and_keeps_the_registry_clean():
    from infera_amd import capi

    if capi.device_count() > 0:
        pytest.skip("a GPU is visible: the full harness runs in the gpu test")
    r = _run("--no-predict")
    assert r == {"threads": 8, "iterations": 10, "predictions_ok": 0, "failures": 0}


@pytest.mark.gpu
def test_native_harness_like_reference():
    r = _run()
    assert r == {"threads": 8, "iterations": 10, "predictions_ok": 80, "failures": 0}
