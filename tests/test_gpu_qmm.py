"""GPU parity of the slimt::qmm boundary (QMM.hh:48-63), through the C ABI.
int32 accumulators: bit-exact. Floats: bit-exact against the oracle in
PORTABLE order, and within 1e-4 (north_star) of the FAITHFUL reference order --
for these ops the two orders coincide, the epilogue being pure IEEE mul/add."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def make_case(seed, M, K, N, x_scale=2.0):
    r = rng(seed)
    x = r.normal(0, x_scale, size=(M, K)).astype(np.float32)
    W = np.clip(np.rint(r.normal(0, 32, size=(N, K))), -127, 127).astype(np.int8)
    bias = r.normal(0, 0.05, size=N).astype(np.float32)
    aq = float(np.float32(127.0 / r.uniform(4, 12)))
    bq = float(np.float32(127.0 / r.uniform(0.3, 1.0)))
    return x, W, bias, aq, bq


def test_mfma_operand_layout_asymmetric(hip, oracle):
    """A = one-hot rows against an asymmetric W catches any row/col/k swap of
    the MFMA fragment maps (cdna guide: 'A=I-check with ASYMMETRIC B')."""
    M, K, N = 16, 64, 16
    x = np.zeros((M, K), dtype=np.float32)
    for i in range(M):
        x[i, (i * 5 + 3) % K] = 1.0  # q = 1 at a distinct k per row
    W = ((np.arange(N)[:, None] * 7 + np.arange(K)[None, :] * 3) % 251 - 125).astype(np.int8)
    got = hip.affine_acc_i32(x, W, 1.0)
    want = oracle.affine_acc(x, W, 1.0)
    assert np.array_equal(got, want)


# the GEMM shape census of SURVEY App. D (decoder M=B, encoder M=B*S) + edges
SHAPES = [
    (1, 64, 8), (3, 64, 16), (16, 64, 16), (17, 128, 24), (5, 256, 40),
    (64, 256, 256), (64, 256, 512), (64, 256, 1536), (64, 1536, 256), (64, 256, 4096),
    (256, 512, 512), (2048, 256, 256), (300, 256, 1536), (130, 2048, 512), (33, 512, 2048),
    # many rows: slimt_hip_affine takes the 128-row tiling (gemm_tile.hip): a ragged last row
    # block, K in one / three / four 512-deep chunks, N not a multiple of the 128-column block
    (1100, 512, 512), (1024, 1536, 256), (1300, 2048, 576), (2048, 256, 1536),
]


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_affine_accumulators_bit_exact(hip, oracle, M, K, N):
    x, W, bias, aq, bq = make_case(M + K + N, M, K, N)
    got = hip.affine_acc_i32(x, W, aq)
    want = oracle.affine_acc(x, W, aq)
    assert got.dtype == np.int32 and np.array_equal(got, want)


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_affine_and_dot_float(hip, oracle, M, K, N):
    x, W, bias, aq, bq = make_case(7 * M + K + N, M, K, N)
    for b in (bias, None):  # affine / dot
        got = hip.affine(x, W, b, aq, bq)
        for mode in (oracle.PORTABLE, oracle.FAITHFUL):
            oracle.set_mode(mode)
            want = oracle.affine(x, W, b, aq, bq)
            assert np.array_equal(got, want), np.abs(got - want).max()
    oracle.set_mode(oracle.FAITHFUL)


def test_affine_saturation_and_ties(hip, oracle):
    """activations that saturate the +-127 clamp and sit on exact .5 ties."""
    M, K, N = 32, 256, 64
    x, W, bias, aq, bq = make_case(5, M, K, N)
    aq = 8.0
    x[0, :] = 1000.0
    x[1, :] = -1000.0
    x[2, :] = (np.arange(K) - 128 + 0.5) / 8.0  # q*aq lands on k + 0.5 exactly
    x[3, :] = 0.0
    W[0, :] = 127
    W[1, :] = -127
    got_acc = hip.affine_acc_i32(x, W, aq)
    assert np.array_equal(got_acc, oracle.affine_acc(x, W, aq))
    assert got_acc[0, 0] == 254 * 127 * K and got_acc[1, 0] == 0
    got = hip.affine(x, W, bias, aq, bq)
    assert np.array_equal(got, oracle.affine(x, W, bias, aq, bq))


@pytest.mark.parametrize("M,K,N,n_idx", [(1, 64, 512, 8), (16, 256, 2048, 256), (64, 256, 32000, 4096),
                                         (7, 512, 4000, 1000)])
def test_affine_with_select(hip, oracle, M, K, N, n_idx):
    x, W, bias, aq, bq = make_case(M + n_idx, M, K, N)
    r = rng(n_idx)
    idx = np.sort(r.choice(N, size=n_idx, replace=False)).astype(np.uint32)
    got = hip.affine_with_select(x, W, bias, aq, bq, idx)
    want = oracle.affine_select(x, W, bias, aq, bq, idx)
    assert got.shape == (M, n_idx) and np.array_equal(got, want)
    # and it is the column gather of the full affine (Intgemm.inl.cc:36-87)
    if N <= 4096:
        assert np.array_equal(got, hip.affine(x, W, bias, aq, bq)[:, idx])


def test_affine_non_finite_activations(hip, oracle):
    """NaN / inf activations quantise like intgemm's PrepareA: NaN -> -127,
    +inf -> 127, -inf -> -127 (accumulators stay exact)."""
    M, K, N = 16, 64, 32
    x, W, bias, aq, bq = make_case(77, M, K, N)
    x[0, 0], x[1, 5], x[2, 9] = np.nan, np.inf, -np.inf
    got = hip.affine_acc_i32(x, W, aq)
    assert np.array_equal(got, oracle.affine_acc(x, W, aq))


def test_affine_leading_dims_flatten(hip, oracle):
    """M = x.size / x.dim(-1): leading dims flatten (Intgemm.inl.cc:101-104)."""
    x, W, bias, aq, bq = make_case(9, 6 * 5, 128, 96)
    got = hip.affine(x.reshape(6, 5, 128), W, bias, aq, bq)
    assert got.shape == (6, 5, 96)
    assert np.array_equal(got.reshape(30, 96), oracle.affine(x, W, bias, aq, bq))


def test_affine_rejects_bad_shapes(hip):
    x = np.zeros((4, 100), dtype=np.float32)
    W = np.zeros((16, 100), dtype=np.int8)
    with pytest.raises(hip.SlimtHipError):
        hip.affine(x, W, None, 1.0, 1.0)  # K % 64 != 0, like intgemm's own constraint
    with pytest.raises(hip.SlimtHipError):
        hip.affine_with_select(np.zeros((4, 64), np.float32), np.zeros((16, 64), np.int8),
                               np.zeros(16, np.float32), 1.0, 1.0, np.array([99], np.uint32))


def test_linearity_property_full_size(hip):
    """Size-independent property at the bench's full decode size: the int32
    accumulators are linear in the quantised activations, so
    acc(q1 + q2) - 127*colsum == acc(q1) + acc(q2) - 2*127*colsum, checked via
    accS(a) + accS(b) - accS(a+b) == accS(0)."""
    M, K, N = 256, 256, 4096
    r = rng(123)
    a = r.integers(-60, 61, size=(M, K)).astype(np.float32)
    b = r.integers(-60, 61, size=(M, K)).astype(np.float32)
    W = r.integers(-127, 128, size=(N, K)).astype(np.int8)
    z = np.zeros_like(a)
    sa, sb, sab, s0 = (hip.affine_acc_i32(v, W, 1.0) for v in (a, b, a + b, z))
    assert np.array_equal(sa.astype(np.int64) + sb - sab, s0.astype(np.int64))
    assert np.array_equal(s0[0], 127 * W.astype(np.int64).sum(axis=1))
