"""GPU parity of the engine-level boundary (Encoder::forward, Decoder::step,
Model::forward) through the C ABI, against the CPU oracle on the same seeded
synthetic models. Everything is compared BIT-EXACT against the oracle's
PORTABLE float order (so every int8 activation, every int32 accumulator and
every token along the way is identical); the distance to the reference's own
scalar float order (FAITHFUL) is a float-rounding effect bounded separately
in tests/test_oracle.py and per op in tests/test_gpu_ops.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kv_forms(m, gm):
    """Every K/V cache form asked for EXPLICITLY on `gm` (a model of the caller's own: the calibration / watch state of a
    shared one would make the form a matter of collection order): yields its name after configuring the model --
    "20" (packed, the tight form never tried), "16" (packed, the tight form tried first around 127 colsum: fits most
    sentence-layers, not all), "24" (packed, 24 bits only), "f32"."""
    centres = np.zeros((m.dec_layers, 2, m.D), dtype=np.int64)
    for l in range(m.dec_layers):
        for t, name in enumerate("kv"):
            W = np.ascontiguousarray(m.params[f"decoder_l{l + 1}_context_W{name}"].data).reshape(m.D, m.D)
            centres[l, t] = 127 * W.astype(np.int64).sum(axis=1)
    for form in ("20", "16", "24", "f32"):
        gm.set_kv_cache_format({"20": 0, "16": 0, "24": 2, "f32": 1}[form])
        if form == "16":
            gm.set_kv_centres(centres.astype(np.int32))
            gm.debug_kv_tight_limit(2 ** 15)
        elif form == "20":
            gm.debug_kv_tight_limit(0)
        yield form

# (preset, eos_bias, B, S, shortlist size or None, ragged)
CONFIGS = [
    ("micro", 3.0, 8, 8, 128, True),
    ("micro", 3.0, 5, 7, None, True),      # full vocab, B not a multiple of 16
    ("mini", 1.0, 16, 12, 256, True),
    ("mini", 1.0, 33, 9, 1000, False),
    ("tiny11", 6.0, 16, 16, 1024, True),
    ("tiny11", 6.0, 19, 32, 2048, True),
    ("tiny11", 6.0, 7, 11, None, True),
    ("tiny11", 6.0, 45, 32, 2048, True),   # 32-row decoder tiles: one full, one partly filled
    ("tiny11", 6.0, 36, 20, None, True),   # ... full vocabulary, second tile only in rows 0..3
    ("tiny11", 6.0, 5, 40, 1024, True),    # S > 32: per-sentence encoder, 128-key decoder attention
    ("tiny11", 6.0, 3, 33, 512, True),     # ... just past the 32-row kernel: 3 row tiles, 2 key tiles
    ("tiny11", 6.0, 2, 64, None, True),    # ... last length of the 2-key-tile encoder variant
    ("tiny11", 6.0, 2, 65, 1024, True),    # ... first length of the 4-key-tile variant
    ("tiny11", 6.0, 3, 70, 512, True),     # S > 64: keys L and L + 64 share a lane in the row sums
    ("tiny11", 6.0, 2, 96, 512, False),    # ... three key tiles, full-length sentences
    ("tiny11", 6.0, 2, 128, 512, True),    # the reference's wrap length (Frontend.hh:27)
    ("mini", 1.0, 4, 100, 256, True),
    ("base", 6.0, 4, 8, 512, True),
    ("base", 6.0, 19, 32, 1024, True),     # d_head 64 decoder attention, cells in global memory, ragged tile
    ("base", 6.0, 3, 40, 512, True),       # ... S > 32: the generic attention path
]


@pytest.fixture(scope="module")
def engines(hip, oracle, synth_models):
    cache = {}

    def get(preset, eos_bias):
        key = (preset, eos_bias)
        if key not in cache:
            m = synth_models(preset, eos_bias)
            cache[key] = (m, hip.Model(m), oracle.OracleModel(m))
        return cache[key]

    yield get
    for _, gm, _ in cache.values():
        gm.close()


@pytest.mark.parametrize("preset,eos_bias,B,S,n_sl,ragged", CONFIGS)
def test_encoder_every_layer_bit_exact(hip, oracle, engines, preset, eos_bias, B, S, n_sl, ragged):
    from slimt_amd import synth
    m, gm, om = engines(preset, eos_bias)
    ids, lens = synth.make_batch(m.V, B, S, seed=B * 100 + S, ragged=ragged)
    ctx = hip.Context(gm, B, S)
    oracle.set_mode(oracle.PORTABLE)
    mask = oracle.make_mask(lens, S)
    want = [om.embed(ids)]
    for l in range(1, m.enc_layers + 1):
        want.append(om.encoder_layer(l, want[-1], mask))
    oracle.set_mode(oracle.FAITHFUL)
    for mode in (0, 1):  # fused persistent encoder (when supported) / layer-by-layer kernels
        ctx.set_decode_mode(mode)
        enc, emb, layers = ctx.encode(ids, lens, want_embed=True, want_layers=True)
        assert np.array_equal(emb, want[0]), mode
        for l in range(1, m.enc_layers + 1):
            assert np.array_equal(layers[l - 1], want[l]), (mode, l, np.abs(layers[l - 1] - want[l]).max())
        assert np.array_equal(enc, want[-1]), mode
    ctx.close()


@pytest.mark.parametrize("preset,eos_bias,B,S,n_sl,ragged", CONFIGS)
def test_decoder_steps_teacher_forced_bit_exact(hip, oracle, engines, preset, eos_bias, B, S, n_sl,
                                                ragged):
    """Decoder::step with random previous tokens: logits, last-layer attention
    and SSRU states after every step."""
    from slimt_amd import synth
    m, gm, om = engines(preset, eos_bias)
    ids, lens = synth.make_batch(m.V, B, S, seed=B * 100 + S + 1, ragged=ragged)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl)
    ctx = hip.Context(gm, B, S)
    enc, _, _ = ctx.encode(ids, lens)
    ctx.decode_begin(sl)
    oracle.set_mode(oracle.PORTABLE)
    mask = oracle.make_mask(lens, S)
    states = np.zeros((m.dec_layers, B, m.D), dtype=np.float32)
    r = np.random.Generator(np.random.PCG64(5))
    prev = None
    for t in range(5):
        want_logits, want_attn = om.decode_step(enc, mask, states, prev, sl)
        logits, attn, st = ctx.decode_step(prev)
        assert np.array_equal(st, states), (t, np.abs(st - states).max())
        assert np.array_equal(attn, want_attn), t
        assert np.array_equal(logits, want_logits), (t, np.abs(logits - want_logits).max())
        pool = np.arange(m.V) if sl is None else sl
        prev = r.choice(pool, size=B).astype(np.uint32)
    oracle.set_mode(oracle.FAITHFUL)
    ctx.close()


@pytest.mark.parametrize("preset,eos_bias,B,S,n_sl,ragged", CONFIGS)
def test_translate_tokens_lengths_alignments(hip, oracle, engines, preset, eos_bias, B, S, n_sl,
                                             ragged):
    """Model::forward: identical greedy tokens, lengths (EOS bookkeeping) and
    alignment rows, including sentences that finish at staggered steps."""
    from slimt_amd import synth
    m, gm, om = engines(preset, eos_bias)
    ids, lens = synth.make_batch(m.V, B, S, seed=B * 100 + S + 2, ragged=ragged)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl)
    ctx = hip.Context(gm, B, S)
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, w_al, steps = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
    oracle.set_mode(oracle.FAITHFUL)
    # mode 0: persistent fused decoder (when the shape supports it);
    # mode 1: one launch per stage and step; modes 2 / 3 / 4 / 5: the persistent decoder
    # with 16 / 32 / 8 / 4 sentences per workgroup. Same tokens every way.
    for mode in (0, 1, 2, 3, 4, 5):
        ctx.set_decode_mode(mode)
        out, ln, al = ctx.translate(ids, lens, sl, limit_factor=1.5, eos_id=0, want_align=True)
        assert np.array_equal(ln, w_ln), (mode, ln, w_ln)
        assert np.array_equal(out, w_out), mode
        assert np.array_equal(al, w_al), mode
        # a second call on the same context must not see stale state
        out2, ln2, _ = ctx.translate(ids, lens, sl, limit_factor=1.5, eos_id=0)
        assert np.array_equal(out2, out) and np.array_equal(ln2, ln)
    ctx.close()


def test_translate_limit_factor_and_reuse(hip, oracle, engines):
    from slimt_amd import synth
    m, gm, om = engines("micro", 3.0)
    ctx = hip.Context(gm, 16, 16)
    oracle.set_mode(oracle.PORTABLE)
    for (B, S, lf) in [(1, 1, 1.5), (2, 3, 1.0), (16, 16, 0.5), (3, 5, 2.5), (2, 1, 0.5), (3, 3, 0.25)]:
        ids, lens = synth.make_batch(m.V, B, S, seed=B + S, ragged=B > 2)
        w_out, w_ln, w_al, _ = om.translate(ids, lens, None, lf, 0, want_align=True)
        # limit_factor * S < 1: the first step is unconditional (Model.cc:144-157), one token
        assert w_ln.min() >= 1 and w_out.shape[1] == max(1, int(np.float32(lf) * np.float32(S)))
        for mode in (0, 1):
            ctx.set_decode_mode(mode)
            out, ln, al = ctx.translate(ids, lens, None, limit_factor=lf, want_align=True)
            assert np.array_equal(out, w_out) and np.array_equal(ln, w_ln), (B, S, lf, mode)
            assert np.array_equal(al, w_al), (B, S, lf, mode)
    ctx.set_decode_mode(0)
    oracle.set_mode(oracle.FAITHFUL)
    with pytest.raises(hip.SlimtHipError):
        ctx.translate(np.zeros((17, 4), np.uint32), np.full(17, 4, np.uint32))  # B > workspace
    with pytest.raises(hip.SlimtHipError):
        ctx.translate(np.full((2, 4), m.V, np.uint32), np.full(2, 4, np.uint32))  # bad token id
    ctx.close()


def test_decode_invariants_at_bench_size(hip, engines):
    """Size-independent properties at BASELINE's headline size (tiny11, B=256,
    S=32, shortlist 4096): sentences are independent (batch rows can be
    permuted / split without changing any token), tokens come from the
    shortlist, lengths are capped at floor(1.5*S), and two runs agree."""
    from slimt_amd import synth
    m, gm, _ = engines("tiny11", 6.0)
    B, S = 256, 32
    ids, lens = synth.make_batch(m.V, B, S, seed=77, ragged=True)
    sl = synth.make_shortlist(m.V, 4096)
    ctx = hip.Context(gm, B, S)
    out, ln, _ = ctx.translate(ids, lens, sl)
    assert out.shape == (B, 48) and ln.max() <= 48 and ln.min() >= 1
    for b in range(B):
        assert np.all(np.isin(out[b, : ln[b]], sl))
    perm = np.random.Generator(np.random.PCG64(1)).permutation(B)
    out_p, ln_p, _ = ctx.translate(ids[perm], lens[perm], sl)
    assert np.array_equal(out_p, out[perm]) and np.array_equal(ln_p, ln[perm])
    half = B // 2
    out_h, ln_h, _ = ctx.translate(ids[:half], lens[:half], sl)
    assert np.array_equal(out_h, out[:half]) and np.array_equal(ln_h, ln[:half])
    ctx.close()


def test_concurrent_contexts_share_one_model(hip, oracle, engines):
    """slimt::Async semantics (Frontend.cc:212-226): several workers call the
    re-entrant forward concurrently on one shared model. Each worker = one
    context (stream + workspace); every result must equal the oracle's."""
    import threading
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    B, S, W = 48, 24, 6
    sl = synth.make_shortlist(m.V, 2048)
    jobs = [synth.make_batch(m.V, B, S, seed=1000 + i, ragged=True) for i in range(W)]
    oracle.set_mode(oracle.PORTABLE)
    want = [om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3] for ids, lens in jobs]
    oracle.set_mode(oracle.FAITHFUL)
    ctxs = [hip.Context(gm, B, S) for _ in range(W)]
    problems = []

    def check(tag, i, got):
        out, ln, al = got
        if not (np.array_equal(out, want[i][0]) and np.array_equal(ln, want[i][1])
                and np.array_equal(al, want[i][2])):
            rows = np.nonzero((out != want[i][0]).any(axis=1) | (ln != want[i][1]))[0]
            first = [int(np.argmax(out[r] != want[i][0][r])) for r in rows]
            problems.append((tag, i, rows.tolist(), first))

    for i, (ids, lens) in enumerate(jobs):  # serial, all on context 0 (first calls)
        check("serial", i, ctxs[0].translate(ids, lens, sl, want_align=True))

    def work(i):
        for it in range(3):
            check(f"concurrent#{it}", i, ctxs[i].translate(jobs[i][0], jobs[i][1], sl, want_align=True))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in ctxs:
        c.close()
    assert not problems, problems


@pytest.mark.parametrize("preset,B,S,n_sl", [("micro", 1, 1, None), ("tiny11", 1, 1, 8), ("tiny11", 1, 32, 4096),
                                              ("mini", 17, 2, 64), ("tiny11", 33, 3, None)])
def test_translate_edge_shapes(hip, oracle, engines, preset, B, S, n_sl):
    """Smallest shapes: one sentence, one source token (floor(1.5 * 1) = 1 decode
    step), a shortlist of one MFMA half-tile, odd batch sizes."""
    from slimt_amd import synth
    m, gm, om = engines(preset, 6.0 if preset == "tiny11" else 1.0)
    ids, lens = synth.make_batch(m.V, B, S, seed=B * 31 + S, ragged=True)
    sl = synth.make_shortlist(m.V, n_sl, frequent=min(100, n_sl)) if n_sl else None
    ctx = hip.Context(gm, B, S)
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
    oracle.set_mode(oracle.FAITHFUL)
    for mode in (0, 1, 2, 3, 5):
        ctx.set_decode_mode(mode)
        out, ln, al = ctx.translate(ids, lens, sl, want_align=True)
        assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out) and np.array_equal(al, w_al), mode
    ctx.close()


@pytest.mark.parametrize("n_sl", [16, 24, 136, 264, 504, 520, 1000, 1032, 4104])
def test_output_layer_tile_counts(hip, oracle, engines, n_sl):
    """Shortlists whose tile counts are odd, not a multiple of the 16 waves or of the 32-tile period of the
    paired epilogue constants (kernels.h, PreparedWeight::cp4): tiles without a partner, a last tile of 8
    columns, fewer tiles than waves -- and one context that sees all of them in turn (the pair array of a
    larger shortlist is reused by a smaller one). Persistent decoder with 16 and 32 sentences per workgroup."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    B, S = 5, 9
    ids, lens = synth.make_batch(m.V, B, S, seed=n_sl, ragged=True)
    ctx = hip.Context(gm, B, S)
    for n in (4104, n_sl, 8):  # large first: its constants stay behind in the buffers
        sl = synth.make_shortlist(m.V, n, frequent=min(100, n))
        oracle.set_mode(oracle.PORTABLE)
        w_out, w_ln, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
        oracle.set_mode(oracle.FAITHFUL)
        for mode in (0, 2, 3, 4):
            ctx.set_decode_mode(mode)
            out, ln, al = ctx.translate(ids, lens, sl, want_align=True)
            assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out) and np.array_equal(al, w_al), (n, mode)
    ctx.close()


@pytest.mark.parametrize("poison", ["nan", "-inf", "nan-in-column-0"])
def test_logits_without_a_maximum_sample_class_zero(hip, oracle, synth_models, poison):
    """Every logit NaN (or -inf): nothing beats the arg-max's start value. The reference's scan starts at class 0
    and stays there (Transformer.cc:287-298); the kernels must do the same -- and must not index the shortlist or
    the embedding with their "no column yet" marker. And a NaN in column 0's logit ALONE: `value > NaN` is never
    true, so the reference stays at class 0 as well, whatever the other columns hold (the kernels' arg-max skips
    NaNs; the rule is applied where the token is taken). All decoder variants; EOS is another id than class 0's,
    so the loop runs its full length on class 0."""
    import copy
    from slimt_amd import synth
    m = copy.deepcopy(synth_models("tiny11", 6.0))
    bias = m.params["decoder_ff_logit_out_b"]
    B, S = 19, 11
    ids, lens = synth.make_batch(m.V, B, S, seed=4, ragged=True)
    shortlists = (synth.make_shortlist(m.V, 1024), None)
    if poison == "nan-in-column-0":
        assert shortlists[0][0] == 0  # column 0 of both output layers is vocabulary id 0
        bias.data.reshape(-1)[0] = np.float32(np.nan)
    else:
        bias.data[...] = np.float32(np.nan) if poison == "nan" else np.float32(-np.inf)
    gm, om = hip.Model(m), oracle.OracleModel(m)
    eos = 0 if poison != "nan-in-column-0" else 7
    for sl in shortlists:
        oracle.set_mode(oracle.PORTABLE)
        w_out, w_ln, _, _ = om.translate(ids, lens, sl, 1.5, eos)
        oracle.set_mode(oracle.FAITHFUL)
        if poison == "nan-in-column-0":
            assert not w_out.any() and (w_ln == int(np.float32(1.5) * np.float32(S))).all()  # class 0 at every step
        ctx = hip.Context(gm, B, S)
        for mode in (0, 1, 2, 3, 5, 6):
            ctx.set_decode_mode(mode)
            out, ln, _ = ctx.translate(ids, lens, sl, eos_id=eos)
            assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out), (poison, mode, sl is None)
        ctx.close()
    gm.close()


def test_translate_everything_finishes_at_step_one(hip, oracle, engines):
    """EOS bias so large that every sentence emits EOS first: the persistent
    decoder leaves its loop early and the remaining output stays zero."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 100.0)
    B, S = 40, 12
    ids, lens = synth.make_batch(m.V, B, S, seed=3, ragged=True)
    sl = synth.make_shortlist(m.V, 512)
    ctx = hip.Context(gm, B, S)
    out, ln, al = ctx.translate(ids, lens, sl, want_align=True)
    assert np.all(ln == 1) and np.all(out[:, 0] == 0) and np.all(out[:, 1:] == 0)
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
    oracle.set_mode(oracle.FAITHFUL)
    assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out) and np.array_equal(al, w_al)
    ctx.close()


def test_context_workspace_limits_and_errors(hip, engines):
    from slimt_amd import synth
    m, gm, _ = engines("micro", 3.0)
    ctx = hip.Context(gm, 4, 8)
    ids, lens = synth.make_batch(m.V, 5, 8, seed=1)
    with pytest.raises(hip.SlimtHipError, match="exceeds the context workspace"):
        ctx.translate(ids, lens, None)
    ids, lens = synth.make_batch(m.V, 4, 9, seed=1)
    with pytest.raises(hip.SlimtHipError, match="exceeds the context workspace"):
        ctx.translate(ids, lens, None)
    ids, lens = synth.make_batch(m.V, 4, 8, seed=1)
    bad = ids.copy(); bad[0, 0] = m.V
    with pytest.raises(hip.SlimtHipError, match="out of range"):
        ctx.translate(bad, lens, None)
    with pytest.raises(hip.SlimtHipError, match="length"):
        ctx.translate(ids, lens + 9, None)
    with pytest.raises(hip.SlimtHipError, match="shortlist id"):
        ctx.translate(ids, lens, np.array([0, 1, 2, 3, 4, 5, 6, m.V], np.uint32))
    out, ln, _ = ctx.translate(ids, lens, None)  # the context is still usable after errors
    assert out.shape[0] == 4 and ln.min() >= 1
    with pytest.raises(hip.SlimtHipError):
        hip.Context(gm, 4, 129)  # slimt wraps at 128 (Frontend.hh:27)
    ctx.close()


@pytest.mark.parametrize("B,S", [(96, 64), (40, 128)])
def test_decode_invariants_long_sentences(hip, engines, B, S):
    """The size-independent properties of test_decode_invariants_at_bench_size for
    the long-sentence kernels (per-sentence encoder, 128-key decoder attention) at
    sizes the oracle would take minutes for: rows are independent (permutation,
    split), tokens come from the shortlist, lengths are capped, runs repeat."""
    from slimt_amd import synth
    m, gm, _ = engines("tiny11", 6.0)
    ids, lens = synth.make_batch(m.V, B, S, seed=S, ragged=True)
    sl = synth.make_shortlist(m.V, 2048)
    ctx = hip.Context(gm, B, S)
    T = int(np.float32(1.5) * np.float32(S))
    out, ln, _ = ctx.translate(ids, lens, sl)
    assert out.shape == (B, T) and ln.max() <= T and ln.min() >= 1
    for b in range(B):
        assert np.all(np.isin(out[b, : ln[b]], sl))
    out2, ln2, _ = ctx.translate(ids, lens, sl)
    assert np.array_equal(out2, out) and np.array_equal(ln2, ln)
    perm = np.random.Generator(np.random.PCG64(2)).permutation(B)
    out_p, ln_p, _ = ctx.translate(ids[perm], lens[perm], sl)
    assert np.array_equal(out_p, out[perm]) and np.array_equal(ln_p, ln[perm])
    third = B // 3
    out_h, ln_h, _ = ctx.translate(ids[:third], lens[:third], sl)
    assert np.array_equal(out_h, out[:third]) and np.array_equal(ln_h, ln[:third])
    ctx.set_decode_mode(1)  # one launch per stage: same tokens
    out_s, ln_s, _ = ctx.translate(ids[:8], lens[:8], sl)
    assert np.array_equal(out_s, out[:8]) and np.array_equal(ln_s, ln[:8])
    ctx.close()


@pytest.mark.parametrize("preset,B,S", [("tiny11", 27, 14), ("tiny11", 5, 40), ("base", 19, 16)])
def test_kv_cache_policy_keeps_results(hip, oracle, engines, preset, B, S):
    """The non-temporal K/V variants of the persistent decoder (d_head 32 short and long
    sentences, d_head 64) are separate kernel instantiations that small tests would never
    reach through the per-launch choice: force each policy."""
    from slimt_amd import synth
    m, gm, om = engines(preset, 6.0)
    sl = synth.make_shortlist(m.V, 512)
    ids, lens = synth.make_batch(m.V, B, S, seed=7300 + B, ragged=True)
    oracle.set_mode(oracle.PORTABLE)
    want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    try:
        for policy in (2, 1, 0):
            gm.set_kv_cache_policy(policy)
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), policy
    finally:
        gm.set_kv_cache_policy(0)
        ctx.close()


@pytest.mark.parametrize("preset,S", [("tiny11", s) for s in (1, 2, 3, 4, 5, 6, 7, 9, 13, 16, 21, 31, 32, 33, 40, 47, 50, 63, 64,
                                                              65, 66, 71, 96, 101, 127, 128)] +
                         [("base", s) for s in (3, 4, 7, 10, 16, 29, 32)])
def test_packed_kv_cache_matches_oracle_and_f32_cache(hip, oracle, engines, preset, S):
    """The 24-bit K/V cache (default where supported: tiny11 up to S = 128 -- 33..64 through the 64-row
    encoder and the one-head-per-pass attention, 65..128 through the per-sentence encoder and the
    two-keys-per-lane attention --, base up to S = 32; base caches the signed
    accumulator and adds the column-sum term in the attention) against the oracle and
    against the f32 cache, for sentence lengths on both sides of every layout edge: several
    sentences per encoder workgroup, a last V group of 1..4 keys, S = 1 / 2 / 5 (which fall
    back to f32: a padded group of four keys would not fit their plane), both cache-load
    policies and every decoder tiling (16, 32, 8 and 4 sentences per workgroup). Alignments are
    the head-0 probabilities computed from the unpacked K, so they pin the floats too."""
    from slimt_amd import synth
    m, _, om = engines(preset, 6.0)
    # a FRESH device model: which form a sentence-layer takes depends on the model's calibration / watch state, and a model
    # shared across the module would make that a matter of collection order. Every form is asked for
    # explicitly below, and what the encoder recorded is checked.
    gm = hip.Model(m)
    B = (37 if S <= 64 else 19) if preset == "tiny11" else 21
    sl = synth.make_shortlist(m.V, 768)
    ids, lens = synth.make_batch(m.V, B, S, seed=9100 + S, ragged=True)
    ids, lens = ids.copy(), lens.copy()
    lens[3] = 0  # an empty sentence: everything masked, uniform weights over real V rows
    ids[3, :] = 0
    oracle.set_mode(oracle.PORTABLE)
    want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    seen = set()
    try:
        for form in kv_forms(m, gm):
            for policy in (2, 1):
                gm.set_kv_cache_policy(policy)
                # 16 / 32 / 8 / 4 sentences per decoder workgroup (32: tiny11's short sentences only; 8 and 4: the
                # packed-cache variants, else they fall back to 16)
                for mode in ((2, 3, 4, 5) if preset == "tiny11" else (2, 4, 5)):
                    ctx.set_decode_mode(mode)
                    got = ctx.translate(ids, lens, sl, want_align=True)
                    assert all(np.array_equal(a, b) for a, b in zip(got, want)), (form, policy, mode)
                    f = ctx.debug_kv_formats(m.dec_layers, B)
                    if f is not None:
                        seen |= {(form, int(x)) for x in np.unique(f)}
                        if form == "20":
                            assert not (f == 2).any()  # nobody tried the tight form
        # where the shape has the per-sentence forms at all (not S = 1, 2, 5, 9..12: a padded V group would not fit), the
        # narrow run recorded narrow sentence-layers and the tight run tight ones
        if ("20", 0) in seen:
            assert ("16", 2) in seen or S > 32 and preset == "base", seen
    finally:
        ctx.close()
        gm.close()


@pytest.mark.parametrize("B,S", [(2, 32), (7, 32), (70, 32), (9, 31), (5, 17), (23, 16), (11, 13), (40, 9),
                                  (19, 5), (37, 4), (100, 3), (150, 1)])
def test_tall_encoder_every_layer_and_translate_bit_exact(hip, oracle, engines, B, S):
    """encode_tall.hip (64 rows = floor(64 / S) whole sentences per workgroup; chosen by itself
    only for batches that fill the device): forced here on small batches -- every layer's output
    against the oracle for 1..64 sentences per workgroup, partly filled last workgroups and
    16-query halves (S <= 16 / > 16); then its decoder K/V cache in both storage formats
    through translate (tokens, lengths, alignments)."""
    from slimt_amd import synth
    m, _, om = engines("tiny11", 6.0)
    gm = hip.Model(m)  # its own: the cache forms below are asked for explicitly (kv_forms)
    ids, lens = synth.make_batch(m.V, B, S, seed=5200 + 64 * B + S, ragged=True)
    oracle.set_mode(oracle.PORTABLE)
    mask = oracle.make_mask(lens, S)
    want = [om.embed(ids)]
    for l in range(1, m.enc_layers + 1):
        want.append(om.encoder_layer(l, want[-1], mask))
    sl = synth.make_shortlist(m.V, 640)
    want_t = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    try:
        ctx.set_encode_rows(64)
        enc, emb, layers = ctx.encode(ids, lens, want_embed=True, want_layers=True)
        assert np.array_equal(emb, want[0])
        for l in range(1, m.enc_layers + 1):
            assert np.array_equal(layers[l - 1], want[l]), (l, np.abs(layers[l - 1] - want[l]).max())
        assert np.array_equal(enc, want[-1])
        for form in kv_forms(m, gm):
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want_t)), form
        gm.set_kv_cache_format(0)
        ctx.set_encode_rows(32)
        got = ctx.translate(ids, lens, sl, want_align=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, want_t))
    finally:
        ctx.close()
        gm.close()


@pytest.mark.parametrize("B,S", [(3, 33), (17, 40), (5, 48), (20, 49), (9, 63), (33, 64)])
def test_tall_encoder_medium_sentences_bit_exact(hip, oracle, engines, B, S):
    """Sentences of 33..64 tokens: one per workgroup of the 64-row encoder (four key tiles, three
    or four 16-query tiles; the 64-column softmax in the canonical butterfly order) against the
    oracle layer by layer and through translate; forcing 32-row tiles falls back to the
    per-sentence kernel, with the same results."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    ids, lens = synth.make_batch(m.V, B, S, seed=6100 + 64 * B + S, ragged=True)
    oracle.set_mode(oracle.PORTABLE)
    mask = oracle.make_mask(lens, S)
    want = [om.embed(ids)]
    for l in range(1, m.enc_layers + 1):
        want.append(om.encoder_layer(l, want[-1], mask))
    sl = synth.make_shortlist(m.V, 640)
    want_t = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    try:
        for rows in (0, 32):
            ctx.set_encode_rows(rows)
            enc, emb, layers = ctx.encode(ids, lens, want_embed=True, want_layers=True)
            assert np.array_equal(emb, want[0]), rows
            for l in range(1, m.enc_layers + 1):
                assert np.array_equal(layers[l - 1], want[l]), (rows, l, np.abs(layers[l - 1] - want[l]).max())
            assert np.array_equal(enc, want[-1]), rows
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want_t)), rows
    finally:
        ctx.close()


@pytest.mark.parametrize("Ld", [1, 3])
def test_decoder_depths_other_than_two(hip, oracle, Ld):
    """tiny11's shape with 1 and 3 decoder layers: the per-layer LDS tables of the persistent decoder
    (SSRU cells, packed-cache biases, LayerNorm constants) scale with the depth -- at 3 layers the
    LayerNorm constants no longer fit and are read from global memory, and 33..64-token sentences
    fall back to the long-sentence path; results stay those of the oracle."""
    from slimt_amd import synth
    m = synth.make_model("tiny11", eos_bias=6.0, dims=(256, 1536, 8, 2, Ld, 4000))
    gm, om = hip.Model(m), oracle.OracleModel(m)
    try:
        for B, S in ((21, 32), (9, 13), (7, 48)):
            ids, lens = synth.make_batch(m.V, B, S, seed=8800 + 10 * Ld + S, ragged=True)
            sl = synth.make_shortlist(m.V, 512)
            oracle.set_mode(oracle.PORTABLE)
            want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
            oracle.set_mode(oracle.FAITHFUL)
            ctx = hip.Context(gm, B, S)
            try:
                for fmt in (0, 2, 1):
                    gm.set_kv_cache_format(fmt)
                    got = ctx.translate(ids, lens, sl, want_align=True)
                    assert all(np.array_equal(a, b) for a, b in zip(got, want)), (B, S, fmt)
            finally:
                ctx.close()
    finally:
        gm.set_kv_cache_format(0)
        gm.close()


@pytest.mark.parametrize("budget", [0, 1, 3, 1000])
def test_decoder_admission_and_ticket_launches_keep_results(hip, oracle, engines, budget):
    """Concurrent contexts of one model under every decoder budget (0 = no limit, 1 = one
    decoder at a time, 3 = a launch of 2 workgroups after one of 2): the over-subscribed
    ticket launches and the admission gate only decide where and when a tile runs."""
    import threading
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    B, S, W = 27, 14, 3  # 2 tiles per batch, the second partly filled
    sl = synth.make_shortlist(m.V, 1024)
    jobs = [synth.make_batch(m.V, B, S, seed=7100 + i, ragged=True) for i in range(6)]
    oracle.set_mode(oracle.PORTABLE)
    want = [om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3] for ids, lens in jobs]
    oracle.set_mode(oracle.FAITHFUL)
    gm.set_decoder_budget(budget)
    ctxs = [hip.Context(gm, B, S) for _ in range(W)]
    bad = []

    def work(w):
        # automatic (8 or 4 sentences per workgroup while they fit the budget), then 16 / 8 / 4 forced
        for rep, mode in enumerate((0, 2, 4, 5)):
            ctxs[w].set_decode_mode(mode)
            for i in range(w, len(jobs), W):
                got = ctxs[w].translate(jobs[i][0], jobs[i][1], sl, want_align=True)
                if not all(np.array_equal(a, b) for a, b in zip(got, want[i])):
                    bad.append((w, rep, i))

    ts = [threading.Thread(target=work, args=(w,)) for w in range(W)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for c in ctxs:
        c.close()
    gm.set_decoder_budget(192)
    assert not bad, bad


def test_translate_with_an_empty_sentence(hip, oracle, engines):
    """Length 0 (nothing but padding): every key is masked, the softmax is uniform over
    the S padded positions and their real K/V rows -- the one case in which the decoder's
    padding skip must not skip."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    B, S = 18, 12
    ids, lens = synth.make_batch(m.V, B, S, seed=7400, ragged=True)
    lens = lens.copy()
    lens[3] = 0
    ids = ids.copy()
    ids[3, :] = 0
    sl = synth.make_shortlist(m.V, 512)
    oracle.set_mode(oracle.PORTABLE)
    want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    try:
        for mode in (0, 1):
            ctx.set_decode_mode(mode)
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), mode
    finally:
        ctx.close()


def test_deep_async_queues_keep_results(hip, oracle, engines):
    """Many more launches queued than the admission ring holds (64 events), on more contexts
    than decoders are admitted: every one of the 216 asynchronous translates must still
    produce the oracle's tokens (tickets, admission waits, K/V policy queries, ring reuse)."""
    from slimt_amd import synth
    from test_shortlist import _Hip
    m, gm, om = engines("tiny11", 6.0)
    B, S, W, rounds = 40, 12, 18, 12
    T = int(np.float32(1.5) * np.float32(S))
    sl = synth.make_shortlist(m.V, 512)
    jobs = [synth.make_batch(m.V, B, S, seed=7500 + i, ragged=True) for i in range(3)]
    oracle.set_mode(oracle.PORTABLE)
    want = [om.translate(ids, lens, sl, 1.5, 0)[:2] for ids, lens in jobs]
    oracle.set_mode(oracle.FAITHFUL)
    dev = _Hip()
    gm.set_decoder_budget(12)  # 3 workgroups per batch: 4 decoders admitted at a time
    ctxs = [hip.Context(gm, B, S) for _ in range(W)]
    try:
        d_in = [(dev.to_dev(ids), dev.to_dev(lens)) for ids, lens in jobs]
        d_sl = dev.to_dev(sl)
        outs = []
        for r in range(rounds):
            for w in range(W):
                j = (r + w) % len(jobs)
                d_out = dev.to_dev(np.full((B, T), 7, np.uint32))
                d_len = dev.to_dev(np.full(B, 99, np.uint32))
                ctxs[w].translate_device(d_in[j][0], d_in[j][1], B, S, d_sl, sl.size, 1.5, 0, d_out, d_len,
                                         0, steps_hint=T)
                outs.append((j, d_out, d_len))
        for c in ctxs:
            c.synchronize()
        for j, d_out, d_len in outs:
            ln = dev.from_dev(d_len, (B,), np.uint32)
            out = dev.from_dev(d_out, (B, T), np.uint32)
            assert np.array_equal(ln, want[j][1])
            assert np.array_equal(out, want[j][0][:, :T])
    finally:
        for c in ctxs:
            c.close()
        gm.set_decoder_budget(192)
        dev.free()


def test_model_create_checks_payload_sizes(hip, synth_models):
    """slimt_hip_param.bytes: a payload shorter than its shape (truncated file) is refused
    before anything is uploaded."""
    import copy
    m = copy.copy(synth_models("micro", 3.0))
    m.params = dict(m.params)
    victim = copy.copy(m.params["encoder_l1_ffn_W1"])
    full = victim.payload()

    class Short:
        name, kind, rows, cols = victim.name, victim.kind, victim.rows, victim.cols

        def payload(self):
            return full[: len(full) // 2]

    m.params["encoder_l1_ffn_W1"] = Short()
    with pytest.raises(hip.SlimtHipError, match="bytes, its shape"):
        hip.Model(m)


def test_model_from_bin_container_equals_model_from_params(hip, oracle, engines):
    """slimt_hip_model_create_from_bin (what the Transformer::Transformer hook calls with its `View model`,
    integration/): same tokens, lengths and alignment rows as the model built from the parameter list."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    gb = hip.Model.from_bin(synth.write_bin(m), m.enc_layers, m.dec_layers, m.H)
    try:
        assert (gb.D, gb.F, gb.V, gb.H) == (m.D, m.F, m.V, m.H)
        ids, lens = synth.make_batch(m.V, 21, 17, ragged=True, seed=11)
        sl = synth.make_shortlist(m.V, 1000)
        got = []
        for model in (gm, gb):
            ctx = hip.Context(model, 21, 17)
            got.append(ctx.translate(ids, lens, sl, want_align=True))
            ctx.close()
        for a, b in zip(got[0], got[1]):
            assert np.array_equal(a, b)
        oracle.set_mode(oracle.PORTABLE)
        w_out, w_len, w_al, _ = om.translate(ids, lens, sl, want_align=True)
        oracle.set_mode(oracle.FAITHFUL)
        assert np.array_equal(got[1][0], w_out) and np.array_equal(got[1][1], w_len) and np.array_equal(got[1][2], w_al)
    finally:
        gb.close()


def test_device_resident_ids_out_of_range_do_not_fault(hip, oracle, engines):
    """slimt_hip_translate_device cannot validate arrays that live on the device: a source id (or a shortlist id) past the
    vocabulary must not fault -- it reads the table's last row -- and must not disturb the OTHER sentences of the batch
    (sentences are independent: they still equal the oracle); a sentence length past the padded width S is S."""
    import torch
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    B, S = 70, 32
    ids, lens = synth.make_batch(m.V, B, S, seed=66, ragged=True)
    sl = synth.make_shortlist(m.V, 1024)
    # a length past the padded width means the padded width: sentences 17 and 63 are checked against the oracle at len = S
    full = lens.copy()
    full[[17, 63]] = S
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, _, _ = om.translate(ids, full, sl, 1.5, 0)
    oracle.set_mode(oracle.FAITHFUL)
    bad = ids.copy()
    bad[5, 0] = np.uint32(m.V + 12345)
    bad[41, min(3, int(lens[41]) - 1)] = np.uint32(0xFFFFFFF0)
    bad_lens = lens.copy()
    bad_lens[17] = np.uint32(S + 100)
    bad_lens[63] = np.uint32(0xFFFFFFFF)
    dev = torch.device("cuda", 0)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    T = int(np.float32(1.5) * np.float32(S))
    d_ids, d_len, d_sl = to_dev(bad), to_dev(bad_lens), to_dev(sl)
    d_out = torch.zeros((B, T), dtype=torch.int32, device=dev)
    d_ol = torch.zeros((B,), dtype=torch.int32, device=dev)
    ctx = hip.Context(gm, B, S)
    try:
        for mode in (0, 1, 3):
            ctx.set_decode_mode(mode)
            ctx.translate_device(d_ids.data_ptr(), d_len.data_ptr(), B, S, d_sl.data_ptr(), sl.size, 1.5, 0,
                                 d_out.data_ptr(), d_ol.data_ptr(), 0)
            torch.cuda.synchronize()
            out = d_out.cpu().numpy().view(np.uint32)
            ln = d_ol.cpu().numpy().view(np.uint32)
            ok = np.ones(B, bool)
            ok[[5, 41]] = False
            assert np.array_equal(ln[ok], w_ln[ok]) and np.array_equal(out[ok], w_out[ok]), mode
            assert (ln[~ok] >= 1).all() and (ln[~ok] <= T).all()
    finally:
        ctx.close()
