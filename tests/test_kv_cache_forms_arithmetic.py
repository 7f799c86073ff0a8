"""The three packed forms of the cross-attention K/V cache are exact by three small facts of IEEE single arithmetic; this file
checks them on the CPU with numpy (the kernels themselves are checked against the checker in tests/test_gpu_kv_narrow.py):

* tight (16-bit) form -- the cache holds r = accS - centre[d] as int16, the decoder computes float(r) + float(centre):
  both are integers a float holds exactly, their sum accS is below 2^24, so the addition is exact (decode_fused.hip, attention_row16);
* narrow (20-bit) form -- the decoder rebuilds accS << 12 and its chains run 4096 times the accumulators': scaling every
  product and every partial sum by a power of two changes no rounding, and u / 4096 takes the factor out (attention_row20);
* 24-bit form -- the same with accS << 8 and u / 256 (attention_row24).
The reference computes float(accS) * u + pb per element (qmm/Intgemm.inl.cc:146-153); the hoisted order these forms feed is
DESIGN.md section 2."""
import numpy as np

f32 = np.float32


def test_tight_form_addition_is_exact():
    rng = np.random.Generator(np.random.PCG64(16))
    r = rng.integers(-2 ** 15, 2 ** 15, size=200000, dtype=np.int64)
    centre = rng.integers(-(2 ** 24) + 2 ** 15 + 1, 2 ** 24 - 2 ** 15, size=r.size, dtype=np.int64)
    acc = r + centre
    assert np.abs(acc).max() < 2 ** 24
    got = r.astype(f32) + centre.astype(f32)  # one IEEE addition per value
    assert got.dtype == f32 and np.array_equal(got.astype(np.int64), acc)
    # the corners: int16's ends against the largest centres a float holds exactly
    for rr in (-2 ** 15, 2 ** 15 - 1, 0, -1):
        for cc in (2 ** 24 - 2 ** 15, -(2 ** 24) + 2 ** 15 + 1, 0, 8388607, -8388608):
            assert int(f32(rr) + f32(cc)) == rr + cc


def _chain(q, k, scale):
    """t = fma chain over ascending d of q_d * (k_d * scale), in float32 (np.float32 has no fma: use float64 products rounded once,
    which equals a correctly rounded fma for these magnitudes: 24 x 24 bit significands fit a double exactly, and so does the sum)."""
    t = f32(0.0)
    for qd, kd in zip(q, k):
        t = f32(np.float64(qd) * np.float64(f32(kd * scale)) + np.float64(t))
    return t


def test_scaled_chains_round_like_unscaled_ones():
    rng = np.random.Generator(np.random.PCG64(20))
    for _ in range(200):
        q = (rng.standard_normal(32) * 3).astype(f32)
        acc = rng.integers(-2 ** 19, 2 ** 19, size=32).astype(f32)  # float(accS): exact
        u = f32(rng.uniform(1e-6, 1e-3))
        c = f32(rng.standard_normal())
        plain = f32(np.float64(_chain(q, acc, f32(1.0))) * np.float64(u) + np.float64(c))  # fmaf(t, u, c_h)
        for shift, scale in ((12, f32(4096.0)), (8, f32(256.0))):
            scaled_t = _chain(q, acc, scale)
            assert scaled_t == _chain(q, acc, f32(1.0)) * scale  # every partial sum is the unscaled one times 2^shift
            u_scaled = f32(u / scale)  # exact: a power of two (no underflow at these magnitudes)
            assert f32(u_scaled * scale) == u
            assert f32(np.float64(scaled_t) * np.float64(u_scaled) + np.float64(c)) == plain, shift


def test_packed_integers_come_back():
    """pack20 / unpack20 and pack16 as device_common.h and decode_fused.hip do them, on the integers."""
    rng = np.random.Generator(np.random.PCG64(24))
    x = rng.integers(-2 ** 19, 2 ** 19, size=4096, dtype=np.int64)
    hi, lo = x >> 4, x & 15  # 16-bit plane (arithmetic shift), nibble plane
    assert (hi >= -2 ** 15).all() and (hi < 2 ** 15).all()
    back = ((hi & 0xffff) << 16 | lo << 12).astype(np.uint32).view(np.int32).astype(np.int64)  # {hi byte 1, hi byte 0, lo << 4, 0}
    assert np.array_equal(back, x << 12) and np.array_equal((x << 12).astype(f32).astype(np.int64), x << 12)
    r = rng.integers(-2 ** 15, 2 ** 15, size=4096, dtype=np.int64)
    word = (r[0::2] & 0xffff) | (r[1::2] << 16)  # value c in the low / high half of dword c / 2
    low = ((word & 0xffff) ^ 0x8000) - 0x8000  # sext(WORD_0)
    high = ((word.astype(np.uint32).view(np.int32)) >> 16).astype(np.int64)  # arithmetic shift of the 32-bit dword
    assert np.array_equal(low, r[0::2]) and np.array_equal(high, r[1::2])
