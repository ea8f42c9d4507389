"""dense_narrow16s_kernel (C4: Gemm(128->10)+Softmax over a resident table) since round 6 parks a full tile's 32 x M results in the wave's LDS
tile and writes them as one run of 16-byte pieces, with non-temporal table loads and result stores (-5 % per 50M rows,
profiles/r06_c4_store_ab.txt).  Both switches are read per launch; whatever they are set to, the results are the SAME BITS -- for every output
width 1..16, the three row lengths the kernel is built for, ragged last tiles, and with and without the softmax epilogue."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [64, 128, 256])
def test_parked_nontemporal_stores_are_bit_identical(gpu_api, tmp_path, k):
    from infera_amd import onnx_writer as W
    from infera_amd import synth
    from oracle import oracle

    capi = gpu_api
    saved = {v: os.environ.get(v) for v in ("INFERA_DENSE16S_MODE", "INFERA_DENSE16S_NT")}
    try:
        for m, head in ((1, "none"), (2, "softmax"), (7, "softmax"), (10, "softmax"), (10, "none"), (13, "softmax"), (16, "none")):
            name = f"d16s_{k}_{m}_{head}"
            graph = W.mlp((k, m), acts=[""], final_softmax=head == "softmax")
            path = W.write(str(tmp_path / (name + ".onnx")), graph)
            capi.load_model(name, path)
            try:
                assert capi.get_plan(name)["dense_kernels"][0].startswith("dense_narrow16s_kernel"), capi.get_plan(name)["dense_kernels"]
                ref_model = oracle.Model(path)
                for rows in (4096, 4096 + 17, 40_000 + 31):
                    dev = capi.device_ordinal(0)
                    d_in, d_out = capi.DeviceBuffer(dev, rows * k * 4), capi.DeviceBuffer(dev, rows * m * 4)
                    capi.synth_fill(d_in, 9, 0, rows, k)
                    got = {}
                    for park, nt in ((0, 0), (1, 0), (0, 1), (1, 1)):
                        os.environ["INFERA_DENSE16S_MODE"], os.environ["INFERA_DENSE16S_NT"] = str(park), str(nt)
                        capi.predict_device(name, d_in, rows, k, d_out)
                        got[(park, nt)] = d_out.download((rows, m))
                    for key, y in got.items():
                        assert np.array_equal(y, got[(0, 0)]), (name, rows, key)
                    want = ref_model.predict(synth.table(9, 0, rows, k))
                    assert np.all(np.abs(got[(1, 1)] - want) <= 1e-4 * np.abs(want) + 1e-6), (name, rows)
            finally:
                capi.unload_model(name)
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val


@pytest.mark.parametrize("k", [13, 30, 52, 100])
def test_generic_row_length_kernel_with_the_same_switches_is_bit_identical(gpu_api, tmp_path, k):
    """dense_narrow16g_kernel (tables of 8..128 columns of any alignment) got the non-temporal loads and stores only (parking its results LOSES
    5-20 %: profiles/r06_dense16g_ab.txt): every switch setting gives the same bits, ragged tiles included, and the oracle agrees."""
    from infera_amd import onnx_writer as W
    from infera_amd import synth
    from oracle import oracle

    capi = gpu_api
    saved = {v: os.environ.get(v) for v in ("INFERA_DENSE16S_MODE", "INFERA_DENSE16S_NT")}
    try:
        for m, softmax in ((3, True), (4, False), (10, True), (16, False)):
            name = f"d16g_{k}_{m}_{int(softmax)}"
            path = W.write(str(tmp_path / (name + ".onnx")), W.mlp((k, m), acts=[""], final_softmax=softmax))
            capi.load_model(name, path)
            try:
                assert capi.get_plan(name)["dense_kernels"][0].startswith("dense_narrow16g_kernel"), capi.get_plan(name)["dense_kernels"]
                ref_model = oracle.Model(path)
                for rows in (8192, 8192 + 5, 30_000 + 31):
                    dev = capi.device_ordinal(0)
                    d_in, d_out = capi.DeviceBuffer(dev, rows * k * 4), capi.DeviceBuffer(dev, rows * m * 4)
                    capi.synth_fill(d_in, 9, 0, rows, k)
                    got = {}
                    for park, nt in ((0, 0), (1, 0), (0, 1), (1, 1)):
                        os.environ["INFERA_DENSE16S_MODE"], os.environ["INFERA_DENSE16S_NT"] = str(park), str(nt)
                        capi.predict_device(name, d_in, rows, k, d_out)
                        got[(park, nt)] = d_out.download((rows, m))
                    for key, y in got.items():
                        assert np.array_equal(y, got[(0, 0)]), (name, rows, key)
                    want = ref_model.predict(synth.table(9, 0, rows, k))
                    assert np.all(np.abs(got[(1, 1)] - want) <= 1e-4 * np.abs(want) + 1e-6), (name, rows)
            finally:
                capi.unload_model(name)
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
