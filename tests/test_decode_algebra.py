"""The two identities the fused / matrix-core decode kernels stand on, checked on the CPU in float64 / numpy with the engine's
16-bit rounding restated here (test infrastructure, independent of the HIP code):

* csrc/decfuse.hip: a GEMV over LayerNorm-folded weights commutes with the LayerNorm up to two per-row scalars,
      W' LN(x) + b'  =  rstd (W' x - mean W'1) + b'        (W' = W diag(gamma), b' = W beta + b)
  and with x = x0 + Wo a + bo the product matrix (W' Wo) moves the out-projection in front of it (7 launches per layer).
* csrc/attention.hip attn_cross_mfma_kernel: an f32 number is EXACTLY the sum of three bfloat16 numbers (3 x 8 mantissa bits),
  so an MFMA over the three halves multiplies by the f32 operand; three IEEE binary16 halves reach the f16 subnormal grid.
"""
import numpy as np


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def test_layernorm_commutes_with_folded_gemv_up_to_two_row_scalars():
    rng = np.random.default_rng(0)
    D, N, rows = 320, 96, 5
    W = rng.standard_normal((N, D)) / np.sqrt(D)
    b = rng.standard_normal(N)
    gamma = 1.0 + 0.2 * rng.standard_normal(D)
    beta = 0.1 * rng.standard_normal(D)
    x = rng.standard_normal((rows, D)) * 3.0 + 0.7               # a common offset: the mean term matters
    mean = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + 1e-5)
    ref = ((x - mean) * rstd * gamma + beta) @ W.T + b
    Wf = W * gamma                                                # load-time fold
    bf = W @ beta + b
    got = rstd * (x @ Wf.T - mean * Wf.sum(-1)) + bf
    assert np.abs(got - ref).max() < 1e-11
    # variance from the two sums the producing GEMV leaves behind (sum x, sum x^2), as the kernels compute it
    s1, s2 = x.sum(-1, keepdims=True), (x * x).sum(-1, keepdims=True)
    var2 = np.maximum(s2 / D - (s1 / D) ** 2, 0.0)
    assert np.abs(var2 - var).max() < 1e-11


def test_out_projection_moves_in_front_of_the_query_projection_through_the_product_matrix():
    rng = np.random.default_rng(1)
    D, rows = 256, 4
    Wq = rng.standard_normal((D, D)) / np.sqrt(D); bq = rng.standard_normal(D)
    Wo = rng.standard_normal((D, D)) / np.sqrt(D); bo = rng.standard_normal(D)
    gamma = 1.0 + 0.1 * rng.standard_normal(D); beta = 0.1 * rng.standard_normal(D)
    x0 = rng.standard_normal((rows, D)); a = rng.standard_normal((rows, D))
    x1 = x0 + a @ Wo.T + bo
    mean = x1.mean(-1, keepdims=True); rstd = 1.0 / np.sqrt(x1.var(-1, keepdims=True) + 1e-5)
    ref = ((x1 - mean) * rstd * gamma + beta) @ Wq.T + bq
    Wf = Wq * gamma; bf = Wq @ beta + bq
    qa = x0 @ Wf.T + Wf @ bo                                      # segment 0 of the stacked matrix [W'q ; W'q Wo ; Wo]
    qb = a @ (Wf @ Wo).T                                          # segment 1: the load-time product
    got = rstd * (qa + qb - mean * Wf.sum(-1)) + bf              # finished by the cross-attention kernel
    assert np.abs(got - ref).max() < 1e-10


def test_an_f32_is_exactly_three_bfloat16_halves():
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 6, 20000).astype(np.float32),
                        np.float32([0.0, 1.0, -1.0, 1e-20, 3.4e38 / 4, np.pi, 1.0 / 3.0])])
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)                              # exact in f32 (Sterbenz-type: |r1| <= ulp_bf16(x) / 2)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    # two halves leave at most 2^-16 relative (the first variant of the kernel; tolerance of its test was 3e-5)
    two = hi.astype(np.float64) + mid.astype(np.float64)
    nz = x != 0
    assert (np.abs(two[nz] - x[nz]) / np.abs(x[nz])).max() <= 2.0 ** -16


def test_three_binary16_halves_reach_the_f16_subnormal_grid():
    """fp16 engine: the same three-way split; below 2^-14 the halves sit on the f16 subnormal grid (2^-24), so the sum is exact to
    half a grid step -- far below anything the 16-bit K / V carry."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(20000) * 4.0).astype(np.float32)    # queries / probabilities: |x| well inside the f16 range
    hi = x.astype(np.float16)
    r1 = (x - hi.astype(np.float32)).astype(np.float32)
    mid = r1.astype(np.float16)
    r2 = (r1 - mid.astype(np.float32)).astype(np.float32)
    lo = r2.astype(np.float16)
    err = np.abs(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64) - x.astype(np.float64))
    assert err.max() <= 2.0 ** -25
