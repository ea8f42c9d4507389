"""Models with several graph outputs (SURVEY.md 8f-3 "named / multi outputs"): the reference always serves output 0
(/root/reference infera/src/engine.rs:146-149), which stays the default; `infera_load_model(name, "<path>#<output>")`
registers the same file with another output (by name or index) as the served one, e.g. a classifier's probabilities
beside its label.  Dead branches are dropped per selection."""
import numpy as np
import pytest

from infera_amd import onnx_writer as W
from infera_amd import synth


def _classifier(outputs):
    ws = W._WeightStream(7)
    w1, b1 = ws.take((12, 16), 12), ws.take((16,), 12)
    w2, b2 = ws.take((16, 5), 16), ws.take((5,), 16)
    nodes = [W.node("Gemm", ["X", "w1", "b1"], ["h"]), W.node("Relu", ["h"], ["hidden"]),
             W.node("Gemm", ["hidden", "w2", "b2"], ["logits"]), W.node("Softmax", ["logits"], ["probabilities"], [W.attr_i("axis", 1)]),
             W.node("ArgMax", ["probabilities"], ["label"], [W.attr_i("axis", 1), W.attr_i("keepdims", 0)])]
    info = {"label": W.value_info("label", ["N"], elem_type=W.INT64), "probabilities": W.value_info("probabilities", ["N", 5]),
            "hidden": W.value_info("hidden", ["N", 16])}
    inits = [W.tensor("w1", w1), W.tensor("b1", b1), W.tensor("w2", w2), W.tensor("b2", b2)]
    return W.model("clf", nodes, inits, [W.value_info("X", ["N", 12])], [info[o] for o in outputs])


ALL = ["label", "probabilities", "hidden"]
SHAPES = {"label": [-1], "probabilities": [-1, 5], "hidden": [-1, 16]}


def test_output_selection_metadata_and_errors(built, tmp_path):
    from infera_amd import capi

    path = W.write(str(tmp_path / "clf.onnx"), _classifier(ALL))
    try:
        for sel, name in [("", "label"), ("#label", "label"), ("#0", "label"), ("#probabilities", "probabilities"), ("#1", "probabilities"),
                          ("#hidden", "hidden"), ("#2", "hidden")]:
            capi.load_model("clf", path + sel)
            assert capi.get_model_info("clf")["output_shape"] == SHAPES[name], sel
            steps = [s["kind"] for s in capi.get_plan("clf")["plan"]["steps"]]
            assert ("ArgMax" in steps) == (name == "label") and ("Softmax" in steps) == (name != "hidden"), (sel, steps)  # dead branches dropped
        with pytest.raises(capi.InferaError, match=r"model has no output 'nope' \(outputs: label, probabilities, hidden\)"):
            capi.load_model("clf", path + "#nope")
        with pytest.raises(capi.InferaError, match="no output '3'"):
            capi.load_model("clf", path + "#3")
        # a file whose name really contains '#' is taken as written
        odd = W.write(str(tmp_path / "v#2.onnx"), _classifier(["hidden"]))
        capi.load_model("clf", odd)
        assert capi.get_model_info("clf")["output_shape"] == [-1, 16]
        # a mistyped path containing '#' is reported exactly as it was typed, not truncated at the '#'
        typo = str(tmp_path / "no_such_dir" / "m.onnx") + "#probabilities"
        with pytest.raises(capi.InferaError) as exc:
            capi.load_model("typo", typo)
        assert typo in str(exc.value), str(exc.value)
    finally:
        capi.unload_model("clf")


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 2048])
def test_gpu_each_output_matches_its_single_output_twin(gpu_api, tmp_path, rows):
    from oracle import oracle

    path = W.write(str(tmp_path / "clf.onnx"), _classifier(ALL))
    x = synth.table(11, 0, rows, 12)
    for name in ALL:
        twin = W.write(str(tmp_path / f"twin_{name}.onnx"), _classifier([name]))
        want = oracle.Model(twin).predict(x)
        gpu_api.load_model("clf_" + name, path + "#" + name)
        try:
            got = gpu_api.predict("clf_" + name, x)
        finally:
            gpu_api.unload_model("clf_" + name)
        assert got.shape == want.shape == (rows, {"label": 1, "probabilities": 5, "hidden": 16}[name])
        if name == "label":
            assert np.array_equal(got, want)
        else:
            assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-6)
