"""The split-fp16 convolution mode's host arithmetic (csrc/hip/f16_split.hpp, through the `infera_hip_f16_split` test hook; no GPU): the
fp32 -> fp16 round-to-nearest-even conversion against numpy's on every class of value (normal, subnormal, ties, overflow, inf, NaN), the
power-of-two scale pair, and what the split is for: hi + lo carries >= 22 significant bits of the scaled weight."""
import numpy as np

from infera_amd import capi


def _bits16(x):
    return int(np.float16(x).view(np.uint16))


def test_f16_conversion_matches_numpy_on_every_class_of_value(built):
    rng = np.random.default_rng(5)
    vals = list((rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 6, 4000)).astype(np.float32))
    # ties and boundaries of binary16: halfway points between neighbours, the subnormal range, the overflow threshold, signed zeros
    vals += [np.float32(v) for v in (0.0, -0.0, 1.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 2.0 ** -14, 2.0 ** -14 - 2.0 ** -25, 2.0 ** -24, 2.0 ** -25,
                                      2.0 ** -25 * 1.0000001, 2.0 ** -26, 3 * 2.0 ** -25, 65504.0, 65519.99, 65520.0, 70000.0, -65520.0, 6.1e-5, 5.96e-8)]
    for v in vals:
        # amax = 2^14 -> scale 1.0: the hook's hi is the plain conversion of v
        hi, lo, sc, inv = capi.f16_split(float(v), 16384.0)
        assert sc == 1.0 and inv == 1.0
        with np.errstate(over="ignore"):
            assert hi == _bits16(v), (v, hex(hi), hex(_bits16(v)))
            hf = np.float32(np.uint16(hi).view(np.float16))
            if np.isfinite(hf):
                assert lo == _bits16(np.float32(v) - hf), v
    for special in (np.inf, -np.inf):
        assert capi.f16_split(float(special), 16384.0)[0] == _bits16(special)
    assert (capi.f16_split(float("nan"), 16384.0)[0] & 0x7c00) == 0x7c00 and (capi.f16_split(float("nan"), 16384.0)[0] & 0x3ff) != 0


def test_scale_brings_the_maximum_into_range_and_the_split_keeps_22_bits(built):
    rng = np.random.default_rng(6)
    for amax in (1.0, 0.3, 7e-5, 123456.0, 1e-20, 1e20, 3.0e38, 1e-38, 0.0):
        _, _, sc, inv = capi.f16_split(0.0, amax)
        assert sc * inv == 1.0 and np.log2(sc) == int(np.log2(sc))
        if 1e-30 < amax < 1e30:
            assert 2.0 ** 14 <= amax * sc < 2.0 ** 15
        for v in (rng.uniform(-1, 1, 50) * amax).astype(np.float32):
            hi, lo, sc, _ = capi.f16_split(float(v), amax)
            back = float(np.uint16(hi).view(np.float16)) + float(np.uint16(lo).view(np.float16))
            x = float(np.float32(v) * np.float32(sc))
            if abs(x) >= 2.0 ** -3:  # lo is a normal half: 11 + 11 bits
                assert abs(back - x) <= 2.0 ** -22 * abs(x), (v, amax)
            else:                    # lo subnormal: absolute 2^-25 of the scaled range
                assert abs(back - x) <= 2.0 ** -25
