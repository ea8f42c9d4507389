import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: asserts a timing RATIO or rate threshold (always also `gpu`); collected LAST, so that under `-x` a noisy "
                                       "box can stop the run only after every parity / functional test has been reached")


def pytest_collection_modifyitems(config, items):
    """Timing assertions run after everything else: `-m gpu -x` stops at the first failure, and a rate that dips on one noisy
    box must not leave test_parity_gpu.py, test_sql_surface.py, ... unreached.  Stable sort: the order inside each group stays pytest's."""
    items.sort(key=lambda it: 1 if it.get_closest_marker("perf") else 0)


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


@pytest.fixture(scope="session")
def default_bench_run():
    """bench.py's default invocation, run ONCE per session: the contract test reads its fields, the perf test (collected last) its ratios."""
    env = dict(os.environ)
    env.pop("INFERA_DEVICES", None)
    return run_bench(["--steps", "5", "--warmup", "2", "--cpu-seconds", "3", "--e2e-reps", "2"], env=env)  # (asserts ONE line, <= 4096 bytes)


@pytest.fixture(scope="session")
def scan_stress_run(tmp_path_factory, built):
    """tests/native/scan_stress (16 scanners + registration churn + model churn), run ONCE per session: functional asserts in
    test_native_harness.py, the rate-under-churn ratio in its perf test."""
    import json
    import subprocess

    from infera_amd import onnx_writer as W

    native = os.path.join(ROOT, "tests", "native")
    subprocess.run(["make", "-C", native, "scan_stress"], check=True, capture_output=True)
    mlp = W.write(str(tmp_path_factory.mktemp("stress") / "mlp128.onnx"), W.mlp((128, 256, 64, 1)))
    # (INFERA_ZERO_COPY_MAX_INFLIGHT=0: every chunk whose blocks are registered is fetched in place -- the registry under maximum pressure)
    p = subprocess.run([os.path.join(native, "scan_stress"), mlp, os.path.join(ROOT, "tests", "golden", "linear.onnx"), "2", "16"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, INFERA_ZERO_COPY_MAX_INFLIGHT="0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


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
    # ... and NOTHING else on stdout (round 6: gloo announced its peers there from C++; bench.py points fd 1 at stderr for everything but the line)
    assert [l for l in p.stdout.splitlines() if l.strip()] == lines, p.stdout[-2000:]
    assert len(lines[0]) <= 4096, len(lines[0])
    try:
        with open(detail) as fh:
            full = json.load(fh)
    finally:
        os.unlink(detail)
    return json.loads(lines[0]), full
