"""K2's division by wave-uniform constants (monorun_pnp.hip::div_by_uniform): float32(float64(x) * RN64(1 / c)) must be the IEEE float32
quotient x / c for EVERY x — the decoded istd feeds a bit-exact threshold and the fused kernel keeps the plain division.  The argument is
in the kernel's comment (a float32 quotient is never within 2^-49 of a rounding boundary; the float64 product is within 2^-52); this is
the arithmetic itself, restated in numpy, over wide-exponent samples, the edge values and exhaustively over one binade of x."""
import numpy as np


def _same(c, x):
    with np.errstate(all='ignore'):
        ref = x / c
        got = (x.astype(np.float64) * (np.float64(1.0) / np.float64(c))).astype(np.float32)
    return (ref.view(np.uint32) == got.view(np.uint32)) | (np.isnan(ref) & np.isnan(got))


def test_fp64_product_rounds_to_the_ieee_float32_quotient():
    rng = np.random.default_rng(7)
    edge = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754942e-38, 1.17549435e-38, 3.4028235e38, -3.4028235e38, 1.0, -1.0], np.float32)
    consts = [np.float32(v) for v in (0.04, 1.0, 3.0, 0.1, 7.3e-5, 1.7e9, 1.0 / 3.0, 2.5, 1e-38, 3e38, 1e-45, 0.0)] + list(rng.uniform(1e-3, 1e3, 24).astype(np.float32))
    for c in consts:
        mant = rng.standard_normal(100_000).astype(np.float32)
        x = np.concatenate([mant * np.float32(2.0) ** rng.integers(-140, 120, mant.size).astype(np.float32), edge])
        assert _same(c, x).all(), f'c = {c!r}'
    # every float32 of one binade (2^23 significands) against the constants the decode chain uses in the tests (sd_sq, std_scale) and a few others
    x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3f800000)).view(np.float32)
    for c in (np.float32(0.04), np.float32(0.2), np.float32(3.0), np.float32(0.7071068), np.float32(1.9999999)):
        assert _same(c, x).all(), f'c = {c!r} (exhaustive binade)'
