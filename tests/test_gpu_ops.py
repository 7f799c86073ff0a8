"""GPU parity of the float ops of slimt/TensorOps.hh through the C ABI:
bit-exact vs the oracle's PORTABLE order, <= 1e-4 vs its FAITHFUL order
(the reference's scalar libm path) -- tolerance from BASELINE.json north_star."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def both_modes(oracle, fn):
    oracle.set_mode(oracle.PORTABLE)
    p = fn()
    oracle.set_mode(oracle.FAITHFUL)
    f = fn()
    return p, f


@pytest.mark.parametrize("rows,cols", [(1, 64), (7, 128), (64, 256), (100, 512), (3, 96), (5, 1000)])
def test_layer_norm(hip, oracle, rows, cols):
    r = rng(rows * cols)
    x = r.normal(0.2, 3.0, size=(rows, cols)).astype(np.float32)
    s = (1 + r.normal(0, 0.05, size=cols)).astype(np.float32)
    b = r.normal(0, 0.05, size=cols).astype(np.float32)
    got = hip.layer_norm(x, s, b)
    p, f = both_modes(oracle, lambda: oracle.layer_norm(x, s, b))
    assert np.array_equal(got, p)
    assert np.max(np.abs(got - f)) <= TOL


@pytest.mark.parametrize("rows,cols", [(1, 1), (8, 5), (64, 32), (33, 64), (17, 100), (4, 128), (2, 300)])
def test_softmax(hip, oracle, rows, cols):
    r = rng(rows + cols)
    x = r.normal(0, 4.0, size=(rows, cols)).astype(np.float32)
    if cols > 3:
        x[0, cols // 2:] = -99999999.0  # masked keys
    got = hip.softmax(x)
    p, f = both_modes(oracle, lambda: oracle.softmax(x))
    assert np.array_equal(got, p)
    assert np.max(np.abs(got - f)) <= TOL
    assert np.allclose(got.sum(-1), 1.0, atol=1e-5)


def test_highway(hip, oracle):
    r = rng(3)
    x, y, g = (r.normal(0, 3, size=5000).astype(np.float32) for _ in range(3))
    g[:6] = [0.0, -100.0, 100.0, -1e-8, 87.0, -87.0]
    got = hip.highway(x, y, g)
    p, f = both_modes(oracle, lambda: oracle.highway(x, y, g))
    assert np.array_equal(got, p)
    assert np.max(np.abs(got - f)) <= TOL


@pytest.mark.parametrize("B,H,Tq,S,dh", [(2, 8, 1, 32, 32), (3, 8, 32, 32, 32), (2, 8, 16, 16, 64),
                                        (1, 8, 1, 128, 32), (2, 8, 70, 70, 8), (2, 8, 1, 5, 16),
                                        (1, 8, 128, 128, 64)])
def test_sdpa(hip, oracle, B, H, Tq, S, dh):
    r = rng(B * 1000 + Tq * 10 + S)
    q = r.normal(0, 1.5, size=(B, H, Tq, dh)).astype(np.float32)
    k = r.normal(0, 1.5, size=(B, H, S, dh)).astype(np.float32)
    v = r.normal(0, 1.5, size=(B, H, S, dh)).astype(np.float32)
    lengths = r.integers(1, S + 1, size=B).astype(np.uint32)
    lengths[0] = S
    mask = oracle.make_mask(lengths, S)
    got_out, got_attn = hip.sdpa(q, k, v, mask)
    (p_out, p_attn), (f_out, f_attn) = both_modes(oracle, lambda: oracle.sdpa(q, k, v, mask))
    assert np.array_equal(got_attn, p_attn)
    assert np.array_equal(got_out, p_out)
    assert np.max(np.abs(got_attn - f_attn)) <= TOL
    assert np.max(np.abs(got_out - f_out)) <= TOL
    for b in range(B):
        assert np.all(got_attn[b, :, :, lengths[b]:] == 0)  # pads get zero weight
