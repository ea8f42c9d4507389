import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build libinfera.so + the oracle once per session (no-ops when up to date)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def models(tmp_path_factory, built):
    """Benchmark/parity ONNX models written by infera_amd.onnx_writer (deterministic, seed 1234)."""
    from infera_amd import onnx_writer as W

    d = tmp_path_factory.mktemp("models")
    out = {
        "linear": os.path.join(ROOT, "tests", "golden", "linear.onnx"),
        "multi_output": os.path.join(ROOT, "tests", "golden", "multi_output.onnx"),
        "linear_dyn": W.write(str(d / "linear_dyn.onnx"), W.linear_dyn()),
        "mlp": W.write(str(d / "mlp.onnx"), W.mlp((128, 256, 64, 1))),
        "logreg": W.write(str(d / "logreg.onnx"), W.logreg_softmax(128, 10)),
        "identity_dyn": W.write(str(d / "identity_dyn.onnx"), W.identity(4)),
    }
    return out


@pytest.fixture(scope="session")
def gpu_api(built):
    """The ctypes mirror of the C ABI, on a box where the HIP backend sees a GPU (for `-m gpu` tests)."""
    from infera_amd import capi

    assert capi.device_count() >= 1, capi.get_devices()
    return capi


def run_bench(args, env=None, launcher=None, timeout=1200):
    """Runs bench.py (optionally under `launcher`, e.g. torch.distributed.run) and returns (compact stdout line, full detail object).
    The driver's contract: exactly ONE JSON line on stdout, small enough for its 8 KB capture window; everything else in --detail."""
    import json
    import subprocess
    import tempfile

    detail = tempfile.NamedTemporaryFile(prefix="bench_detail_", suffix=".json", delete=False).name
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + list(args) + ["--detail", detail]
    e = dict(os.environ if env is None else env)
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) <= 4096, len(lines[0])
    try:
        with open(detail) as fh:
            full = json.load(fh)
    finally:
        os.unlink(detail)
    return json.loads(lines[0]), full
