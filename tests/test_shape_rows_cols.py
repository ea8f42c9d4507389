"""The PRODUCT's shape_rows_cols (infera_amd/csrc/host/engine.cpp) pinned on the reference's own unit-test table
(/root/reference infera/src/engine.rs:321-328), through the C ABI -- round 1 only ran the oracle's copy against it.
GPU part: real models whose served output has rank 1 / 3 / 4 report rows/cols by that rule."""
import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth

# engine.rs:321-328, verbatim cases
TABLE = [([], (1, 1)), ([5], (5, 1)), ([2, 3], (2, 3)), ([2, 3, 4], (2, 12)), ([1, 1, 1, 1], (1, 1))]
# the `.max(1)` arm (engine.rs:25) and plain N-D products
EXTRA = [([7, 0], (7, 1)), ([0], (0, 1)), ([3, 1], (3, 1)), ([4, 2, 5, 3], (4, 30)), ([2048, 1000], (2048, 1000))]


@pytest.mark.parametrize("shape,want", TABLE + EXTRA)
def test_product_shape_rows_cols_matches_reference_table(built, shape, want):
    from infera_amd import capi
    from oracle import oracle

    assert capi.shape_rows_cols(shape) == want
    assert oracle.shape_rows_cols(shape) == want  # the checker agrees with the product on every case


def _conv_model(out_kind):
    ws = W._WeightStream(5)
    w, b = ws.take((4, 3, 3, 3), 27), ws.take((4,), 27)
    nodes = [W.node("Conv", ["X", "w", "b"], ["c"], [W.attr_ints("pads", [1, 1, 1, 1])])]
    inits = [W.tensor("w", w), W.tensor("b", b)]
    if out_kind == "rank4":
        nodes.append(W.node("Relu", ["c"], ["Y"]))
        out = W.value_info("Y", ["N", 4, 6, 6])
    elif out_kind == "rank3":
        nodes += [W.node("Relu", ["c"], ["r"]), W.node("Reshape", ["r", "shp"], ["Y"])]
        inits.append(W.tensor("shp", np.array([0, 4, 36], np.int64)))
        out = W.value_info("Y", ["N", 4, 36])
    else:  # rank 1: a label per row, keepdims=0 -> [N]
        nodes += [W.node("GlobalAveragePool", ["c"], ["g"]), W.node("Flatten", ["g"], ["f"]),
                  W.node("ArgMax", ["f"], ["Y"], [W.attr_i("axis", 1), W.attr_i("keepdims", 0)])]
        out = W.value_info("Y", ["N"], elem_type=W.INT64)
    return W.model("shape_" + out_kind, nodes, inits, [W.value_info("X", ["N", 3, 6, 6])], [out])


@pytest.mark.parametrize("kind,shape_tail", [("rank4", [4, 6, 6]), ("rank3", [4, 36]), ("rank1", [])])
def test_output_shape_metadata(built, tmp_path, kind, shape_tail):
    from infera_amd import capi

    p = W.write(str(tmp_path / "m.onnx"), _conv_model(kind))
    capi.load_model("shp", p)
    try:
        assert capi.get_model_info("shp")["output_shape"] == [-1] + shape_tail
    finally:
        capi.unload_model("shp")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,cols", [("rank4", 144), ("rank3", 144), ("rank1", 1)])
@pytest.mark.parametrize("rows", [1, 7])
def test_gpu_results_report_rows_cols_by_the_rule(gpu_api, tmp_path, kind, cols, rows):
    from oracle import oracle

    p = W.write(str(tmp_path / "m.onnx"), _conv_model(kind))
    x = synth.table(3, 0, rows, 3 * 6 * 6)
    want = oracle.Model(p).predict_blob(x.tobytes())
    gpu_api.load_model("shp", p)
    try:
        got = gpu_api.predict_from_blob("shp", x.tobytes())
        # (infera_predict hands over a rank-2 tensor, which a rank-4 input rejects -- as in the reference, engine.rs:139-141)
        with pytest.raises(gpu_api.InferaError, match="rank"):
            gpu_api.predict("shp", x)
    finally:
        gpu_api.unload_model("shp")
    assert got.shape == (rows, cols) == want.shape
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-6)
