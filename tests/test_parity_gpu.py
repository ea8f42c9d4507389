"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerance (BASELINE.json north_star): 1e-4 relative in fp32.  Written here as
|got - want| <= 1e-4 * |want| + 1e-6 -- the absolute floor only matters for outputs within 1e-2 of
zero, where the relative measure of an fp32 dot product of 128..256 O(1) terms is ill-conditioned.
Observed error is ~1e-6 (summation order is the only difference: exact-fp32 MFMA).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-6


def assert_close(got, want):
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = RTOL * np.abs(want.astype(np.float64)) + ATOL
    bad = err > tol
    assert not bad.any(), f"{bad.sum()} / {bad.size} elements out of tolerance; worst err {err.max():.3e}"


@pytest.fixture(scope="module")
def api(built):
    from infera_amd import capi

    assert capi.device_count() >= 1, capi.get_devices()
    return capi


def test_reference_golden_linear(api, models):
    # test/sql/test_core_functionality.test:48-56, docs/examples/e3_integration_and_errors.sql:14-24
    api.load_model("linear", models["linear"])
    assert api.predict("linear", np.array([[1, 2, 3]], np.float32)).tolist() == [[1.75]]
    assert api.predict("linear", np.zeros((1, 3), np.float32)).tolist() == [[0.25]]
    assert api.predict_from_blob("linear", np.zeros(3, np.float32).tobytes()).tolist() == [[0.25]]
    api.unload_model("linear")


def test_reference_golden_multi_output(api, models):
    # test/sql/test_multi_output.test:23-26
    api.load_model("multi_output", models["multi_output"])
    out = api.predict("multi_output", np.array([[1, 2, 3, 4]], np.float32))
    assert out.tolist() == [[1.0, 2.0, 3.0, 4.0]]
    api.unload_model("multi_output")


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 2048, 4096 + 17])
def test_mlp_fused_vs_oracle(api, models, rows):
    from infera_amd import synth
    from oracle import oracle

    api.load_model("mlp", models["mlp"])
    assert api.get_plan("mlp")["exec"][0] == "mlp3_fused"
    x = synth.table(42, 1000, rows, 128)
    assert_close(api.predict("mlp", x), oracle.Model(models["mlp"]).predict(x))


@pytest.mark.parametrize("rows", [1, 5, 2048, 5000])
def test_logreg_softmax_vs_oracle(api, models, rows):
    from infera_amd import synth
    from oracle import oracle

    api.load_model("logreg", models["logreg"])
    x = synth.table(7, 0, rows, 128)
    got = api.predict("logreg", x)
    assert_close(got, oracle.Model(models["logreg"]).predict(x))
    np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-5)


def test_linear_dyn_rows(api, models):
    from oracle import oracle

    api.load_model("lin", models["linear_dyn"])
    x = np.array([[1, 2, 3], [0.5, 1, 1.5], [-1, 0, 2]], np.float32)
    got = api.predict("lin", x)
    assert got.ravel().tolist() == [1.75, 1.0, -0.75]
    assert_close(got, oracle.Model(models["linear_dyn"]).predict(x))


def test_device_path_and_synth_fill(api, models):
    from infera_amd import synth

    api.load_model("mlp", models["mlp"])
    rows = 10_000
    dev = api.device_ordinal(0)
    d_in = api.DeviceBuffer(dev, rows * 128 * 4)
    d_out = api.DeviceBuffer(dev, rows * 4)
    api.synth_fill(d_in, 42, 123, rows, 128)
    x = synth.table(42, 123, rows, 128)
    assert np.array_equal(d_in.download((rows, 128)), x)  # generator is bit-exact across numpy / HIP
    assert api.predict_device("mlp", d_in, rows, 128, d_out) == (rows, 1)
    assert np.array_equal(d_out.download((rows, 1)), api.predict("mlp", x))
