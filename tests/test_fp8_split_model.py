"""CPU statement of the operand split the e4m3 cross-attention uses on the fp8 matrix cores (csrc/attention.hip: split3_e4m3,
attn_cross_mfma8_kernel): the f32 query (scaled to amax = 448) and the probabilities (x 256) ride through
v_mfma_f32_16x16x32_fp8_fp8 as three e4m3 terms, x = t0 + t1 / 16 + t2 / 256, each residual computed exactly in f32.  This test
pins what that representation is worth -- the "12 significant bits" DESIGN.md 6b' quotes -- so the 2e-4 tolerance of the GPU test
(tests/test_gpu_kernels.py::test_cross_attention_over_the_e4m3_cache) has a derivation and not just a measurement."""
import numpy as np

from tests import helpers as Hh


def _recombine(x):
    t0, t1, t2 = Hh.e4m3_split3(x)
    return t0 + t1 / 16.0 + t2 / 256.0, (t0, t1, t2)


def test_three_e4m3_terms_carry_twelve_to_fourteen_bits():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-448, 448, 200000), rng.standard_normal(200000) * 30, rng.standard_normal(200000) * 0.5,
                        rng.standard_normal(100000) * 1e-3, 256 * np.exp(-rng.uniform(0, 20, 200000)), [448.0, -448.0, 0.0, 2.0 ** -9]])
    x = x[np.abs(x) <= 448].astype(np.float32).astype(np.float64)
    y, terms = _recombine(x)
    for t in terms:                                             # every term is on the e4m3 grid and in range
        assert np.array_equal(Hh.e4m3_round(t), t) and np.abs(t).max() <= 448
    err = np.abs(y - x)
    a = np.abs(x)
    assert (err[a >= 0.125] / a[a >= 0.125]).max() < 2.0 ** -14        # normal range of all three terms: 3 x 4 bits + rounding
    m = (a >= 2.0 ** -6) & (a < 0.125)
    assert (err[m] / a[m]).max() < 2.0 ** -12                          # the third term is subnormal here
    assert err[a < 2.0 ** -6].max() <= 2.0 ** -18                      # below the first term's normal range: absolute 2^-10 / 256
    # probabilities: p in (0, 1] scaled by 256 -- relative 2^-12 where the first term is normal, else absolute 2^-26 per key
    p = np.exp(-rng.uniform(0, 30, 100000))
    xp = (256.0 * p).astype(np.float32).astype(np.float64)
    yp, _ = _recombine(xp)
    assert (np.abs(yp - xp) / 256.0 <= np.maximum(2.0 ** -12 * xp / 256.0, 2.0 ** -26)).all()


def test_split_is_exact_on_what_three_terms_can_hold():
    # integers up to 2^12 scaled into range: 12 bits are representable without error
    k = np.arange(-4095, 4096, dtype=np.float64)
    x = k / 4096.0 * 256.0
    y, _ = _recombine(x)
    assert np.array_equal(y, x.astype(np.float32).astype(np.float64))
