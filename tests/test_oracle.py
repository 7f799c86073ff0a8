"""CPU tests of the oracle (the checker itself): every op is re-derived
independently in numpy / torch fp32 / float64 and compared; the two accumulator
formulations must agree exactly; FAITHFUL (reference's libm + sequential
sums) and PORTABLE (GPU-reproducible order) modes must agree to float
rounding. PARITY UNPINNED: the reference has no golden vectors for this path
(SURVEY 8c), so these independent re-derivations are what pins the oracle."""
import numpy as np
import pytest

from conftest import ulp_diff


def rng(seed=0):
    return np.random.Generator(np.random.PCG64(seed))


def test_quantize_rne_and_clamp(oracle):
    # ties go to even (intgemm cvtps_epi32), clamp to +-127 (never -128)
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, 127.5, 300.0, -300.0, -127.5, 0.49999],
                 dtype=np.float32)
    q = oracle.quantize(x, 1.0)
    assert q.tolist() == [0, 2, 2, 0, -2, -2, 126, 127, 127, -127, -127, 0]
    # NaN / inf: intgemm's convert (INT_MIN) + saturating packs + max(-127)
    special = np.array([np.nan, np.inf, -np.inf], dtype=np.float32)
    assert oracle.quantize(special, 1.0).tolist() == [-127, 127, -127]
    r = rng(1)
    x = r.normal(0, 3, size=4096).astype(np.float32)
    aq = np.float32(17.3)
    want = np.clip(np.rint(x * aq), -127, 127).astype(np.int8)
    assert np.array_equal(oracle.quantize(x, float(aq)), want)


@pytest.mark.parametrize("M,K,N", [(1, 64, 8), (5, 128, 24), (16, 256, 256), (33, 1536, 40)])
def test_gemm_signed_vs_shifted_identity(oracle, M, K, N):
    r = rng(M * 1000 + N)
    q = r.integers(-127, 128, size=(M, K)).astype(np.int8)
    W = r.integers(-127, 128, size=(N, K)).astype(np.int8)
    acc = oracle.gemm_i8(q, W, shifted=False)
    accS = oracle.gemm_i8(q, W, shifted=True)
    ref = q.astype(np.int64) @ W.astype(np.int64).T
    colsum = W.astype(np.int64).sum(axis=1)
    assert np.array_equal(acc.astype(np.int64), ref)
    # accS = acc + 127 * colsum
    assert np.array_equal(accS.astype(np.int64), ref + 127 * colsum[None, :])


def test_gemm_extreme_values_no_overflow(oracle):
    K = 2048
    q = np.full((2, K), 127, dtype=np.int8)
    W = np.full((3, K), -127, dtype=np.int8)
    W[1] = 127
    accS = oracle.gemm_i8(q, W, shifted=True)
    assert accS[0, 0] == 254 * -127 * K and accS[0, 1] == 254 * 127 * K


def _affine_numpy(x, W, bias, aq, bq):
    """Independent float32 re-derivation of SURVEY App. A.4."""
    f = np.float32
    q = np.clip(np.rint(x * f(aq)), -127, 127).astype(np.int64)
    Wl = W.astype(np.int64)
    accS = (q + 127) @ Wl.T
    colsum = Wl.sum(axis=1)
    a_alpha = f(127.0) / f(aq)
    b_alpha = f(127.0) / f(bq)
    mult = (f(-1.0) * (a_alpha * b_alpha)) / f(127.0)
    pb = colsum.astype(np.float32) * mult
    if bias is not None:
        pb = pb + bias.astype(np.float32)
    else:
        pb = pb + f(0.0)
    u = f(1.0) / (f(aq) * f(bq))
    return accS.astype(np.float32) * u + pb[None, :]


@pytest.mark.parametrize("M,K,N,with_bias", [(3, 64, 16, True), (16, 256, 256, True),
                                             (7, 256, 48, False), (4, 1536, 256, True)])
def test_affine_matches_numpy(oracle, M, K, N, with_bias):
    r = rng(K + N)
    x = r.normal(0, 2.0, size=(M, K)).astype(np.float32)
    W = np.clip(np.rint(r.normal(0, 32, size=(N, K))), -127, 127).astype(np.int8)
    bias = r.normal(0, 0.05, size=N).astype(np.float32) if with_bias else None
    aq, bq = np.float32(127 / 6.0), np.float32(127 / 0.7)
    y = oracle.affine(x, W, bias, float(aq), float(bq))
    want = _affine_numpy(x, W, bias, aq, bq)
    assert np.array_equal(y, want)
    # the Ruy provider's float order agrees to rounding
    y_ruy = oracle.affine(x, W, bias, float(aq), float(bq), provider="ruy")
    assert np.max(np.abs(y - y_ruy)) <= 2e-4 * max(1.0, float(np.max(np.abs(y))))


def test_affine_select_is_column_gather(oracle):
    r = rng(7)
    M, K, N = 5, 128, 200
    x = r.normal(0, 2.0, size=(M, K)).astype(np.float32)
    W = r.integers(-127, 128, size=(N, K)).astype(np.int8)
    bias = r.normal(0, 0.05, size=N).astype(np.float32)
    idx = np.sort(r.choice(N, size=48, replace=False)).astype(np.uint32)
    full = oracle.affine(x, W, bias, 20.0, 150.0)
    sel = oracle.affine_select(x, W, bias, 20.0, 150.0, idx)
    assert np.array_equal(sel, full[:, idx])


def test_exp_portable_vs_libm(oracle):
    xs = np.concatenate([np.linspace(-90, 0, 20001), -np.logspace(-8, 1.9, 2000)]).astype(np.float32)
    oracle.set_mode(oracle.PORTABLE)
    ep = np.array([oracle.lib().so_exp(float(x)) for x in xs], dtype=np.float32)
    oracle.set_mode(oracle.FAITHFUL)
    ef = np.array([oracle.lib().so_exp(float(x)) for x in xs], dtype=np.float32)
    big = ef > 1e-37
    assert ulp_diff(ep[big], ef[big]).max() <= 2
    assert np.max(np.abs(ep - ef)) <= 1.5e-7
    ref = np.exp(xs.astype(np.float64))
    ok = xs > -80
    assert np.max(np.abs(ep[ok] - ref[ok]) / ref[ok]) < 3e-7


def test_layer_norm_modes_and_float64(oracle):
    r = rng(3)
    for D in (64, 256, 512):
        x = r.normal(0.3, 2.5, size=(9, D)).astype(np.float32)
        s = (1 + r.normal(0, 0.05, size=D)).astype(np.float32)
        b = r.normal(0, 0.05, size=D).astype(np.float32)
        oracle.set_mode(oracle.FAITHFUL)
        yf = oracle.layer_norm(x, s, b)
        oracle.set_mode(oracle.PORTABLE)
        yp = oracle.layer_norm(x, s, b)
        x64 = x.astype(np.float64)
        mean = x64.mean(-1, keepdims=True)
        var = ((x64 - mean) ** 2).mean(-1, keepdims=True)
        ref = s * ((x64 - mean) / np.sqrt(var + 1e-6)) + b
        assert np.max(np.abs(yf - ref)) < 2e-5
        assert np.max(np.abs(yp - ref)) < 2e-5
        assert np.max(np.abs(yf - yp)) < 1e-5
    oracle.set_mode(oracle.FAITHFUL)


def test_layer_norm_faithful_is_sequential(oracle):
    # exact restatement of TensorOps.cc:553-578 in numpy float32 scalar loops
    r = rng(4)
    D = 96
    x = r.normal(0, 3, size=(3, D)).astype(np.float32)
    s = (1 + r.normal(0, 0.05, size=D)).astype(np.float32)
    b = r.normal(0, 0.05, size=D).astype(np.float32)
    oracle.set_mode(oracle.FAITHFUL)
    y = oracle.layer_norm(x, s, b)
    f = np.float32
    for j in range(3):
        acc = f(0)
        for i in range(D):
            acc = f(acc + x[j, i])
        mean = f(acc / f(D))
        sq = f(0)
        for i in range(D):
            v = f(x[j, i] - mean)
            sq = f(sq + f(v * v))
        sigma = f(np.sqrt(f(f(sq / f(D)) + f(1e-6))))
        for i in range(D):
            want = f(f(s[i] * f(f(x[j, i] - mean) / sigma)) + b[i])
            assert y[j, i] == want


def test_softmax_and_sdpa_vs_torch(oracle):
    torch = pytest.importorskip("torch")
    r = rng(5)
    B, H, Tq, S, dh = 2, 4, 5, 12, 16
    q = r.normal(0, 1.5, size=(B, H, Tq, dh)).astype(np.float32)
    k = r.normal(0, 1.5, size=(B, H, S, dh)).astype(np.float32)
    v = r.normal(0, 1.5, size=(B, H, S, dh)).astype(np.float32)
    lengths = np.array([12, 7], dtype=np.uint32)
    mask = oracle.make_mask(lengths, S)
    assert mask[1, 6] == 0 and mask[1, 7] == np.float32(-99999999.0)
    tq, tk, tv, tm = map(torch.from_numpy, (q, k, v, mask))
    scores = (tq @ tk.transpose(-1, -2)) * (1.0 / np.sqrt(np.float32(dh))) + tm[:, None, None, :]
    p = torch.softmax(scores, dim=-1)
    ref_out = (p @ tv).numpy()
    for mode in (oracle.FAITHFUL, oracle.PORTABLE):
        oracle.set_mode(mode)
        out, attn = oracle.sdpa(q, k, v, mask)
        assert np.max(np.abs(attn - p.numpy())) < 2e-6
        assert np.max(np.abs(out - ref_out)) < 1e-5
        assert np.all(attn[1, :, :, 7:] == 0)
    oracle.set_mode(oracle.FAITHFUL)


def test_highway_sigmoid(oracle):
    r = rng(6)
    x, y, g = (r.normal(0, 3, size=1000).astype(np.float32) for _ in range(3))
    g[:4] = [0.0, -100.0, 100.0, -1e-8]
    for mode in (oracle.FAITHFUL, oracle.PORTABLE):
        oracle.set_mode(mode)
        out = oracle.highway(x, y, g)
        sg = 1.0 / (1.0 + np.exp(-g.astype(np.float64)))
        assert np.max(np.abs(out - (sg * x + (1 - sg) * y))) < 2e-6
    oracle.set_mode(oracle.FAITHFUL)


def test_sinusoid_and_greedy(oracle):
    pos = oracle.sinusoidal_signal(0, 8, 64)
    # position 0: sin = 0 on the first half, cos = 1 on the second
    assert np.all(pos[0, :32] == 0) and np.all(pos[0, 32:] == 1)
    p = np.arange(8)[:, None].astype(np.float64)
    inc = np.log(10000.0) / (32 - 1)
    ang = p * np.exp(-inc * np.arange(32))[None, :]
    assert np.max(np.abs(pos[:, :32] - np.sin(ang))) < 1e-5
    assert np.max(np.abs(pos[:, 32:] - np.cos(ang))) < 1e-5
    logits = np.array([[1, 3, 3, 2], [5, 5, 5, 5], [-1, -2, -0.5, -0.5]], dtype=np.float32)
    assert oracle.greedy_sample(logits).tolist() == [1, 0, 2]  # first max wins
    words = np.array([10, 20, 30, 40], dtype=np.uint32)
    assert oracle.greedy_sample(logits, words).tolist() == [20, 10, 30]


def test_model_translate_semantics(oracle, synth_models):
    """Greedy-loop bookkeeping (Model.cc:111-185): lengths include EOS, finished
    sentences stop recording, the loop is capped at floor(1.5 * S)."""
    from slimt_amd import synth
    m = synth_models("micro", eos_bias=3.0)
    om = oracle.OracleModel(m)
    ids, lens = synth.make_batch(m.V, 8, 8, ragged=True)
    sl = synth.make_shortlist(m.V, 128)
    oracle.set_mode(oracle.PORTABLE)
    out, ln, al, steps = om.translate(ids, lens, sl, want_align=True)
    Tmax = 12
    assert out.shape == (8, Tmax) and steps <= Tmax
    assert ln.max() <= Tmax and ln.min() >= 1
    assert len(set(ln.tolist())) > 1, "fixture should finish at staggered steps"
    for b in range(8):
        n = int(ln[b])
        toks = out[b, :n]
        if n < steps:  # finished early => last recorded token is EOS, none before
            assert toks[-1] == 0 and not np.any(toks[:-1] == 0)
        assert np.all(out[b, n:] == 0)
        assert np.all(np.isin(toks, sl))
        # alignment rows: probabilities over the first lens[b] keys, one per token
        a = al[b]
        assert np.allclose(a[:n, : lens[b]].sum(-1), 1.0, atol=1e-5)
        assert np.all(a[n:] == 0) and np.all(a[:, lens[b]:] == 0)
    # reference-cost mode (per-step K/V recompute + PrepareBias) is identical
    om2 = oracle.OracleModel(m, reference_cost=True)
    out2, ln2, al2, _ = om2.translate(ids, lens, sl, want_align=True)
    assert np.array_equal(out, out2) and np.array_equal(ln, ln2) and np.array_equal(al, al2)
    # full vocabulary == shortlist of everything
    out3, ln3, _, _ = om.translate(ids, lens, None)
    out4, ln4, _, _ = om.translate(ids, lens, np.arange(m.V, dtype=np.uint32))
    assert np.array_equal(out3, out4) and np.array_equal(ln3, ln4)
    oracle.set_mode(oracle.FAITHFUL)


def test_faithful_vs_portable_model_level(oracle, synth_models):
    """Drift between the reference's scalar float order (FAITHFUL) and the
    GPU-reproducible order (PORTABLE). Each float op agrees to ~1e-6 (tests
    above); across a whole layer the only larger effect is a re-quantisation
    landing on the other side of a .5 boundary (1 int8 LSB), which perturbs
    that one row by ~1e-2. So: the typical element agrees to float rounding,
    and the overwhelming majority of rows agree within north_star's 1e-4."""
    from slimt_amd import synth
    m = synth_models("mini", eos_bias=1.0)
    om = oracle.OracleModel(m)
    B, S = 6, 12
    ids, lens = synth.make_batch(m.V, B, S, ragged=True)
    mask = oracle.make_mask(lens, S)
    sl = synth.make_shortlist(m.V, 256)
    oracle.set_mode(oracle.FAITHFUL)
    x = om.embed(ids)
    enc_in = om.encode(x, mask)
    res = {}
    for mode in (oracle.FAITHFUL, oracle.PORTABLE):
        oracle.set_mode(mode)
        layer = om.encoder_layer(1, x, mask)  # same input in both modes
        states = np.zeros((m.dec_layers, B, m.D), dtype=np.float32)
        logits, attn = om.decode_step(enc_in, mask, states, None, sl)  # same encoder_out
        res[mode] = (layer, logits, attn)
    oracle.set_mode(oracle.FAITHFUL)
    for a, b in zip(res[oracle.FAITHFUL], res[oracle.PORTABLE]):
        d = np.abs(a - b).reshape(-1, a.shape[-1])
        assert np.median(d) < 1e-6
        row_ok = d.max(axis=1) <= 1e-4 * max(1.0, float(np.abs(a).max()))
        assert row_ok.mean() >= 0.9, row_ok.mean()
        assert d.max() < 0.1


def test_oracle_model_is_stateless_across_batches(oracle, synth_models):
    """Regression: the oracle's per-batch cross-attention K/V cache was keyed
    on the encoder-output pointer only; a recycled malloc pointer made a second
    translate() with different input reuse the first batch's K/V."""
    from slimt_amd import synth
    m = synth_models("micro", eos_bias=3.0)
    shared = oracle.OracleModel(m)
    oracle.set_mode(oracle.PORTABLE)
    sl = synth.make_shortlist(m.V, 128)
    for seed in (1, 2, 3, 1):
        ids, lens = synth.make_batch(m.V, 6, 8, seed=seed, ragged=True)
        got = shared.translate(ids, lens, sl, want_align=True)
        fresh = oracle.OracleModel(m).translate(ids, lens, sl, want_align=True)
        for a, b in zip(got[:3], fresh[:3]):
            assert np.array_equal(a, b), seed
    oracle.set_mode(oracle.FAITHFUL)


@pytest.mark.parametrize("preset,B,S", [("mini", 5, 23), ("tiny11", 6, 32), ("tiny11", 3, 100), ("base", 4, 17)])
def test_hoisted_cross_attention_within_tolerance_of_the_literal_order(oracle, synth_models, preset, B, S):
    """The PORTABLE order of the decoder's cross-attention applies the K / V projections' unquantisation
    multiplier and prepared bias AFTER the attention's sums (oracle/slimt_oracle.c, cross_attention_portable);
    the reference dequantises every element first (Intgemm.inl.cc:146-153, Modules.cc:24-86 = FAITHFUL). Same
    real numbers, other roundings: on the same projected query both the probabilities and the context vectors
    agree within north_star's 1e-4 (measured ~1e-6 relative), masked keys are exactly 0 in both, rows sum to 1.
    (An all-masked row is compared like any other: its scores sit next to -1e8, where both orders round alike.)"""
    from slimt_amd import synth
    m = synth_models(preset, eos_bias=1.0)
    om = oracle.OracleModel(m)
    ids, lens = synth.make_batch(m.V, B, S, ragged=True, seed=77)
    lens = lens.copy()
    lens[1] = 0  # an empty sentence: everything masked, uniform weights
    mask = oracle.make_mask(lens, S)
    oracle.set_mode(oracle.FAITHFUL)
    enc = om.encode(om.embed(ids), mask)
    r = np.random.Generator(np.random.PCG64(5))
    yq = r.normal(0, 1.5, size=(B, m.D)).astype(np.float32)
    for layer in range(m.dec_layers):
        oracle.set_mode(oracle.FAITHFUL)
        f_out, f_attn = om.cross_attention(layer, yq, enc, mask)
        oracle.set_mode(oracle.PORTABLE)
        p_out, p_attn = om.cross_attention(layer, yq, enc, mask)
        oracle.set_mode(oracle.FAITHFUL)
        assert np.abs(p_attn - f_attn).max() <= 1e-4
        scale = max(1.0, float(np.abs(f_out).max()))
        assert np.abs(p_out - f_out).max() <= 1e-4 * scale, (np.abs(p_out - f_out).max(), scale)
        assert np.median(np.abs(p_out - f_out)) <= 2e-6 * scale
        for b in range(B):
            if lens[b] > 0:  # (the empty sentence: every key carries the mask, the weights are whatever survives next to -1e8)
                assert not p_attn[b, :, lens[b]:].any() and not f_attn[b, :, lens[b]:].any()
        assert np.allclose(p_attn.sum(axis=2), 1.0, atol=1e-5)


def test_end_to_end_agreement_of_the_two_orders_on_a_larger_sample(oracle, synth_models, capsys):
    """DESIGN 2's end-to-end figure (the device's output == the reference-order output for 55 of 64 sentences of one batch) on
    four times the sample, on the CPU: the device is PORTABLE bit for bit (tests/test_gpu_*), so PORTABLE against FAITHFUL --
    the reference's literal float order, slimt/Modules.cc:24-86, TensorOps.cc:282-315,542-580 -- over 256 sentences of the
    headline shape IS that comparison. Sentences differ only from a near-tie on (the per-op bounds are the tests above): the
    share that stays identical must not fall, and the tokens in front of a sentence's first difference are the bulk."""
    from slimt_amd import synth
    m = synth_models("tiny11", 6.0)
    om = oracle.OracleModel(m)
    sl = synth.make_shortlist(m.V, 4096)
    B, S = 64, 32
    same_sentences = tok_same = tok_total = sentences = 0
    try:
        for seed in (4242, 4243, 4244, 4245):
            ids, lens = synth.make_batch(m.V, B, S, seed=seed)
            res = {}
            for mode in (oracle.FAITHFUL, oracle.PORTABLE):
                oracle.set_mode(mode)
                res[mode] = om.translate(ids, lens, sl, 1.5, 0)[:2]
            (f_out, f_len), (p_out, p_len) = res[oracle.FAITHFUL], res[oracle.PORTABLE]
            for b in range(B):
                n = min(int(f_len[b]), int(p_len[b]))
                same_sentences += int(f_len[b] == p_len[b] and np.array_equal(f_out[b, :n], p_out[b, :n]))
                tok_same += int((f_out[b, :n] == p_out[b, :n]).sum())
                tok_total += max(int(f_len[b]), int(p_len[b]))
            sentences += B
    finally:
        oracle.set_mode(oracle.FAITHFUL)
    with capsys.disabled():
        print(f"\n[faithful-vs-portable] tiny11 {sentences} sentences of {S} tokens: identical sentences {same_sentences} "
              f"({same_sentences / sentences:.3f}), identical tokens {tok_same}/{tok_total} ({tok_same / tok_total:.4f})")
    assert same_sentences / sentences >= 0.80
    assert tok_same / tok_total >= 0.95
