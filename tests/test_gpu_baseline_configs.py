"""GPU parity at BASELINE.json's OWN sizes (configs 2, 3, 4 and the headline),
through the C ABI, against the CPU oracle on the same seeded inputs.

The oracle's C restatement finishes these sizes in seconds (multi-threaded row
loops), so every config is compared BIT-EXACT at full size (tokens, lengths,
alignment rows; the encoder layer by layer), not only through size-independent
properties. The large-M stage-kernel encoder (M = B*S >= 2048: `run_affine_res`
+ `layer_norm_q_kernel` with and without its int8 side output, 64-/32-row GEMM
blocks; engine.cpp encode_device) is the path BASELINE's `base` config takes.

Reference semantics: slimt/Modules.cc:321-334 (EncoderLayer::forward),
slimt/TensorOps.cc:542-580 (layer_norm), slimt/Model.cc:111-204."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(hip, oracle, synth_models):
    cache = {}

    def get(preset, eos_bias):
        key = (preset, eos_bias)
        if key not in cache:
            m = synth_models(preset, eos_bias)
            cache[key] = (m, hip.Model(m), oracle.OracleModel(m, threads=8))
        return cache[key]

    yield get
    for _, gm, _ in cache.values():
        gm.close()


# (preset, B, S, ragged, modes): mode 0 = the persistent encoders (encode_fused.hip for
# D = 256, encode_wide.hip for D = 512), mode 1 = one launch per stage (the `big` M >= 2048
# branch of the stage path at these sizes)
ENCODER_CASES = [
    ("tiny11", 64, 32, True, (1,)),     # M = 2048: first size of the `big` branch, 64-row FFN1 blocks
    ("tiny11", 128, 32, False, (1,)),   # M = 4096: 32-row LayerNorm-GEMM threshold
    ("tiny11", 256, 32, True, (0, 1)),  # headline batch: fused encoder == stage kernels == oracle
    ("base", 64, 32, True, (0, 1)),     # D = 512, M = 2048
    ("base", 128, 32, True, (0, 1)),    # M = 4096, ragged lengths
    ("base", 256, 32, False, (0, 1)),   # BASELINE config 3: M = 8192
    ("base", 100, 21, True, (0, 1)),    # M = 2100: not a multiple of the 64-row block; one sentence per workgroup
    ("base", 37, 5, True, (0,)),        # wide encoder: 6 sentences per workgroup, last workgroup partly empty
    ("base", 33, 11, True, (0,)),       # 2 sentences per workgroup, 10 rows of every tile unused
    ("base", 9, 16, True, (0,)),        # 2 sentences fill the tile exactly
    ("base", 3, 1, False, (0,)),        # one-token sentences
]


@pytest.mark.parametrize("preset,B,S,ragged,modes", ENCODER_CASES)
def test_encoder_large_M_every_layer_bit_exact(hip, oracle, engines, preset, B, S, ragged, modes):
    from slimt_amd import synth
    m, gm, om = engines(preset, 6.0)
    ids, lens = synth.make_batch(m.V, B, S, seed=B * 100 + S + 9, ragged=ragged)
    oracle.set_mode(oracle.PORTABLE)
    mask = oracle.make_mask(lens, S)
    want = [om.embed(ids)]
    for l in range(1, m.enc_layers + 1):
        want.append(om.encoder_layer(l, want[-1], mask))
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    try:
        for mode in modes:
            ctx.set_decode_mode(mode)
            enc, emb, layers = ctx.encode(ids, lens, want_embed=True, want_layers=True)
            assert np.array_equal(emb, want[0]), mode
            for l in range(1, m.enc_layers + 1):
                assert np.array_equal(layers[l - 1], want[l]), (mode, l, np.abs(layers[l - 1] - want[l]).max())
            assert np.array_equal(enc, want[-1]), mode
    finally:
        ctx.close()


# (name, preset, B, S, shortlist size or None, EOS bias: chosen so that the sentences of
# the fixture finish at staggered steps -- some at once, some never)
FULL_CONFIGS = [
    ("config2", "tiny11", 64, 32, 4096, 8.0),    # BASELINE configs[1]
    ("headline", "tiny11", 256, 32, 4096, 8.0),  # BASELINE metric
    ("config3", "base", 256, 32, 4096, 12.0),    # BASELINE configs[2]
    ("config4", "tiny11", 512, 32, None, 10.0),  # BASELINE configs[3]: full 32k-vocabulary output GEMM
]


@pytest.mark.parametrize("name,preset,B,S,n_sl,eos_bias", FULL_CONFIGS)
def test_baseline_config_full_size_bit_exact(hip, oracle, engines, name, preset, B, S, n_sl, eos_bias):
    """Model::forward at the config's full size: tokens, lengths (staggered EOS) and
    alignment rows equal the oracle's; then the size-independent properties
    (permutation, split, membership, repeat)."""
    from slimt_amd import synth
    m, gm, om = engines(preset, eos_bias)
    ids, lens = synth.make_batch(m.V, B, S, seed=900 + B, ragged=True)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl)
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
    oracle.set_mode(oracle.FAITHFUL)
    assert len(set(w_ln.tolist())) >= 4, "fixture should finish at staggered steps"
    ctx = hip.Context(gm, B, S)
    try:
        out, ln, al = ctx.translate(ids, lens, sl, want_align=True)
        assert np.array_equal(ln, w_ln), name
        assert np.array_equal(out, w_out), name
        assert np.array_equal(al, w_al), name
        T = int(np.float32(1.5) * np.float32(S))
        assert out.shape == (B, T) and ln.max() <= T and ln.min() >= 1
        if sl is not None:
            for b in range(B):
                assert np.all(np.isin(out[b, : ln[b]], sl))
        else:
            assert out.max() < m.V
        perm = np.random.Generator(np.random.PCG64(3)).permutation(B)
        out_p, ln_p, _ = ctx.translate(ids[perm], lens[perm], sl)
        assert np.array_equal(out_p, out[perm]) and np.array_equal(ln_p, ln[perm])
        third = B // 3
        out_h, ln_h, _ = ctx.translate(ids[third: 2 * third], lens[third: 2 * third], sl)
        assert np.array_equal(out_h, out[third: 2 * third]) and np.array_equal(ln_h, ln[third: 2 * third])
        out2, ln2, _ = ctx.translate(ids, lens, sl)
        assert np.array_equal(out2, out) and np.array_equal(ln2, ln)
        if name in ("config2", "config3"):
            ctx.set_decode_mode(1)  # one launch per stage and step: same tokens
            out_s, ln_s, al_s = ctx.translate(ids, lens, sl, want_align=True)
            assert np.array_equal(out_s, out) and np.array_equal(ln_s, ln) and np.array_equal(al_s, al)
        if name != "config3":
            ctx.set_decode_mode(3)  # 32 sentences per decoder workgroup
            out_3, ln_3, _ = ctx.translate(ids, lens, sl)
            assert np.array_equal(out_3, out) and np.array_equal(ln_3, ln)
        # mode 0 above chose by occupancy (one context alone: 4 sentences per workgroup where the variant exists);
        # every tiling forced: 16, 8, 4 sentences per workgroup -- tokens, lengths AND alignment rows
        for mode in (2, 4, 5, 6):  # (6: the output layer shared by clusters of four 16-sentence workgroups, tiny11)
            ctx.set_decode_mode(mode)
            out_m, ln_m, al_m = ctx.translate(ids, lens, sl, want_align=True)
            assert np.array_equal(out_m, w_out) and np.array_equal(ln_m, w_ln) and np.array_equal(al_m, w_al), (name, mode)
        gm.set_adaptive_decoder_rows(False)  # mode 0 with the adaptive choice off: always 16
        ctx.set_decode_mode(0)
        out_m, ln_m, al_m = ctx.translate(ids, lens, sl, want_align=True)
        gm.set_adaptive_decoder_rows(True)
        assert np.array_equal(out_m, w_out) and np.array_equal(ln_m, w_ln) and np.array_equal(al_m, w_al), name
    finally:
        ctx.close()


def test_sharded_4096_sentences_equal_one_stream(hip, oracle, engines):
    """BASELINE config 5's unit of work on ONE device: 4096 sentences cut by
    `plan_shards` into 8 x 512 (one shard per GPU), the shards run as B=512 batches on
    four contexts concurrently -- every sentence's tokens equal those of
    the same sentence translated in a single B=256 pass (sentences are independent:
    the shard plan changes where a sentence runs, never its result), and shard 0
    equals the oracle."""
    import threading
    from slimt_amd import synth
    from slimt_amd.sharding import plan_shards
    m, gm, om = engines("tiny11", 6.0)
    N, S, Bq = 4096, 32, 256
    ids, lens = synth.make_batch(m.V, N, S, seed=4096, ragged=True)
    sl = synth.make_shortlist(m.V, 4096)
    plan = plan_shards(N, 512, 8)  # per rank: [(start, count)]
    assert [len(p) for p in plan] == [1] * 8 and all(p[0][1] == 512 for p in plan)
    shards = [(p[0][0], p[0][0] + p[0][1]) for p in plan]
    assert sorted(shards) == [(i * 512, (i + 1) * 512) for i in range(8)]
    ref = hip.Context(gm, Bq, S)
    want_out = np.zeros((N, 48), np.uint32)
    want_len = np.zeros(N, np.uint32)
    for lo in range(0, N, Bq):
        o, l, _ = ref.translate(ids[lo: lo + Bq], lens[lo: lo + Bq], sl)
        want_out[lo: lo + Bq], want_len[lo: lo + Bq] = o, l
    ref.close()
    oracle.set_mode(oracle.PORTABLE)
    o_out, o_len, _, _ = om.translate(ids[:512], lens[:512], sl, 1.5, 0)
    oracle.set_mode(oracle.FAITHFUL)
    assert np.array_equal(want_out[:512], o_out) and np.array_equal(want_len[:512], o_len)
    ctxs = [hip.Context(gm, 512, S) for _ in range(4)]
    got_out = np.zeros_like(want_out)
    got_len = np.zeros_like(want_len)

    def work(w):
        for r in range(w, 8, 4):  # "rank" r's shard: one batch of 512 sentences
            lo, hi = shards[r]
            o, l, _ = ctxs[w].translate(ids[lo:hi], lens[lo:hi], sl)
            got_out[lo:hi], got_len[lo:hi] = o, l

    ts = [threading.Thread(target=work, args=(w,)) for w in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for c in ctxs:
        c.close()
    assert np.array_equal(got_len, want_len) and np.array_equal(got_out, want_out)


def test_gpu_vs_reference_float_order_token_agreement(hip, oracle, engines, capsys):
    """How far is the GPU's translation from the reference's OWN float order
    (oracle FAITHFUL: libm exp, sequential row sums -- what slimt's scalar
    TensorOps compute)? The GPU is bit-identical to the PORTABLE order; the two
    orders differ by float rounding, which flips an int8 re-quantisation by one LSB
    here and there, and on a random-weight model (flat logits) such a flip can
    change an argmax. Measured here, end to end on tiny11 B=64: the fraction of
    identical sentences / tokens, and for every sentence at its FIRST differing step
    (same history on both sides) the reference-order logit margin between the two
    candidates -- a near-tie, never a different distribution."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    B, S = 64, 32
    ids, lens = synth.make_batch(m.V, B, S, seed=11, ragged=True)
    sl = synth.make_shortlist(m.V, 4096)
    ctx = hip.Context(gm, B, S)
    g_out, g_len, _ = ctx.translate(ids, lens, sl)
    ctx.close()
    oracle.set_mode(oracle.FAITHFUL)
    f_out, f_len, _, _ = om.translate(ids, lens, sl, 1.5, 0)
    mask = oracle.make_mask(lens, S)
    enc = om.encode(om.embed(ids), mask)
    states = np.zeros((m.dec_layers, B, m.D), np.float32)
    prev = None
    undiverged = np.ones(B, bool)
    margins = []
    for t in range(int(f_len.max())):
        logits, _ = om.decode_step(enc, mask, states, prev, sl)
        tok = sl[np.argmax(logits, axis=1)]
        for b in range(B):
            if undiverged[b] and t < min(g_len[b], f_len[b]):
                assert tok[b] == f_out[b, t]
                if g_out[b, t] != tok[b]:
                    gi = int(np.searchsorted(sl, g_out[b, t]))
                    spread = float(logits[b].max() - logits[b].min())
                    margins.append((b, t, float(logits[b].max() - logits[b, gi]) / spread))
                    undiverged[b] = False
        prev = tok.astype(np.uint32)
    same_sentences = sum(
        1 for b in range(B) if g_len[b] == f_len[b] and np.array_equal(g_out[b, : g_len[b]], f_out[b, : f_len[b]]))
    tok_same = sum(int((g_out[b, : min(g_len[b], f_len[b])] == f_out[b, : min(g_len[b], f_len[b])]).sum())
                   for b in range(B))
    tok_total = int(np.maximum(g_len, f_len).sum())
    with capsys.disabled():
        print(f"\n[faithful-vs-gpu] tiny11 B={B} S={S}: identical sentences {same_sentences}/{B} "
              f"({same_sentences / B:.3f}), identical tokens {tok_same}/{tok_total} ({tok_same / tok_total:.4f}); "
              f"first-divergence margins (fraction of the logit spread): "
              f"{[round(x[2], 4) for x in margins]}")
    # measured (deterministic inputs): 55 of 64 sentences, 96.5 % of the tokens, margins <= 0.033 (56 / 96.9 % before the
    # cached cross-attention took the hoisted order, round 4) -- synthetic N(0, 32)
    # weights; parity on trained weights is unpinned (no real model exists here)
    assert same_sentences / B >= 0.85
    assert tok_same / tok_total >= 0.95
    assert all(x[2] <= 0.04 for x in margins), margins  # near-ties only
