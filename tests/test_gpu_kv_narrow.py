"""The narrow (20-bit) form of the packed cross-attention K/V cache and its 24-bit fallback
(include/slimt_hip.h, slimt_hip_model_set_kv_cache_format; kernels.h, FusedDecodeArgs::kv_fmt).

What the cache holds is the K / V projections' shifted accumulators accS of the encoder output
(/root/reference/slimt/Modules.cc:248-249 recomputes them every step; qmm/Intgemm.inl.cc:146-153 is
where an accumulator becomes a float). The encoder stores a workgroup's sentences of a layer in 20
bits per value when EVERY accumulator of their K and V lies in [-2^19, 2^19), else in 24 bits. Either
form gives back the same integers, so every result must stay what the checker (oracle/, PORTABLE order)
computes -- and WHICH form each sentence got is itself predictable from the checker's own accumulators,
which pins the boundary: a sentence with one accumulator at or past the limit must take the 24-bit
form next to narrow ones in the same decoder workgroup."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kv_accumulators(oracle, m, om, ids, lens):
    """accS of every decoder layer's K and V projection for every row of the batch: [Ld][2][B][S][D] (int32), from the
    checker's encoder output (bit-identical to the device's, tests/test_gpu_engine.py)."""
    B, S = ids.shape
    oracle.set_mode(oracle.PORTABLE)
    enc = om.encode(om.embed(ids), oracle.make_mask(lens, S))
    oracle.set_mode(oracle.FAITHFUL)
    rows = np.ascontiguousarray(enc.reshape(B * S, m.D), dtype=np.float32)
    acc = np.zeros((m.dec_layers, 2, B, S, m.D), dtype=np.int32)
    for l in range(m.dec_layers):
        for t, name in enumerate("kv"):
            W = m.params[f"decoder_l{l + 1}_context_W{name}"]
            aq = float(np.asarray(m.params[f"decoder_l{l + 1}_context_W{name}_QuantMultA"].data).ravel()[0])
            acc[l, t] = oracle.affine_acc(rows, np.ascontiguousarray(W.data).reshape(m.D, m.D), aq).reshape(B, S, m.D)
    return acc


def colsum_centres(m, jitter_seed=None):
    """Centres for the tight (16-bit) form (include/slimt_hip.h, slimt_hip_model_set_kv_centres), int32 [Ld][2][D]: 127 colsum
    -- the form then holds the signed accumulator -- plus, for tests of the arithmetic, arbitrary offsets: any integers
    must give the same results."""
    out = np.zeros((m.dec_layers, 2, m.D), dtype=np.int64)
    for l in range(m.dec_layers):
        for t, name in enumerate("kv"):
            W = np.ascontiguousarray(m.params[f"decoder_l{l + 1}_context_W{name}"].data).reshape(m.D, m.D)  # payload [N][K]
            out[l, t] = 127 * W.astype(np.int64).sum(axis=1)
    if jitter_seed is not None:
        out += np.random.Generator(np.random.PCG64(jitter_seed)).integers(-6000, 6001, size=out.shape)
    return out.astype(np.int32)


def centred(acc, centres):
    """accS - centre: what the tight form holds (kernels.h, FusedDecodeArgs::kv_centre)."""
    return acc.astype(np.int64) - centres.astype(np.int64)[:, :, None, None, :]


def expected_forms(acc, limit, group, signed=None, tight_limit=0):
    """1 = 24-bit: some accumulator of the workgroup's sentences (`group` consecutive ones; every row, padding included)
    outside [-limit, limit); 2 = 16-bit, where the tight form is tried (tight_limit > 0): every SIGNED accumulator of
    them in [-tight_limit, tight_limit); else 0 = 20-bit."""
    Ld, _, B = acc.shape[:3]
    outside = ((acc < -limit) | (acc >= limit)).any(axis=(1, 3, 4))  # [Ld][B]
    want = np.zeros((Ld, B), dtype=np.uint8)
    for s0 in range(0, B, group):
        want[:, s0:s0 + group] = outside[:, s0:s0 + group].any(axis=1, keepdims=True)
    if tight_limit > 0:
        out16 = ((signed < -tight_limit) | (signed >= tight_limit)).any(axis=(1, 3, 4))
        for s0 in range(0, B, group):
            fits = ~out16[:, s0:s0 + group].any(axis=1)
            want[fits, s0:s0 + group] = 2
    return want


@pytest.mark.parametrize("preset,B,S,rows", [("tiny11", 37, 32, 64), ("tiny11", 37, 32, 32), ("tiny11", 41, 16, 64),
                                              ("tiny11", 23, 21, 32), ("tiny11", 50, 7, 64), ("tiny11", 19, 29, 64),
                                              ("tiny11", 9, 8, 32), ("tiny11", 30, 13, 64),
                                              ("tiny11", 17, 40, 64), ("tiny11", 9, 64, 64), ("tiny11", 20, 33, 64), ("tiny11", 11, 57, 64),
                                              ("tiny11", 9, 128, 0), ("tiny11", 7, 65, 0), ("tiny11", 5, 100, 0), ("tiny11", 6, 121, 0),
                                              ("base", 21, 32, 32), ("base", 19, 16, 32), ("base", 26, 7, 32), ("base", 7, 25, 32)])
def test_forms_follow_the_accumulators_and_results_do_not(hip, oracle, synth_models, preset, B, S, rows):
    """tiny11: both encoders (64- and 32-row tiles) write the narrow form of the shifted accumulator -- sentences of 33..64
    tokens one per 64-row workgroup, read by the one-head-per-pass attention --; base (D = 512, one 32-row encoder): the
    narrow form holds the shifted accumulator too, its 24-bit form the signed one."""
    from slimt_amd import synth
    m = synth_models(preset, 6.0)
    gm, om = hip.Model(m), oracle.OracleModel(m)
    ctx = hip.Context(gm, B, S)
    try:
        ids, lens = synth.make_batch(m.V, B, S, seed=4400 + 64 * B + S, ragged=True)
        sl = synth.make_shortlist(m.V, 640)
        oracle.set_mode(oracle.PORTABLE)
        want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
        oracle.set_mode(oracle.FAITHFUL)
        acc = kv_accumulators(oracle, m, om, ids, lens)
        peak = np.abs(acc.astype(np.int64)).max(axis=(1, 3, 4))  # [Ld][B]
        assert peak.max() < 2 ** 19  # the synthetic model: every sentence fits the narrow form
        ctx.set_encode_rows(rows)
        group = max(1, rows // S)  # (sentences of more than 32 tokens: one per workgroup, whichever encoder takes them)
        # limits: the real one, and three that split this batch's sentences (the median peak, its neighbours)
        order = np.sort(peak.ravel())
        limits = [2 ** 19, int(order[len(order) // 2]), int(order[len(order) // 4]) + 1, int(order[-1]), int(order[-1]) + 1, 1]
        # the tight (16-bit) form: written by the 64-row encoder (tiny11: sentences of up to 64 tokens), the per-sentence one
        # (65..128) and the D = 512 one (base: up to 32), read by the tilings of 16 / 8 / 4 sentences and, up to 32 tokens, of 32 (mode 3);
        # its limits likewise: int16's, two that split the batch, none
        tight_here = True  # (every encoder writes the tight form: 64- and 32-row tiles, one sentence per workgroup, D = 512)
        centres = colsum_centres(m, jitter_seed=B * 100 + S)
        if tight_here:
            gm.set_kv_centres(centres)
            assert np.array_equal(gm.debug_kv_centres(m.dec_layers, m.D), centres)
        signed = centred(acc, centres)
        order16 = np.sort(np.abs(signed).max(axis=(1, 3, 4)).ravel())
        tights = [2 ** 15, int(order16[len(order16) // 2]), int(order16[len(order16) // 4]) + 1, int(order16[-1]) + 1, 0, 1]
        tights = [min(t, 2 ** 15) for t in tights]
        seen_forms = set()
        for limit, tight in zip(limits, tights):
            gm.debug_kv_narrow_limit(limit)
            gm.debug_kv_tight_limit(tight)
            # 16 / 32 / 8 / 4 sentences per decoder workgroup (32: tiny11's sentences of up to 32 tokens only)
            for mode in ((2, 3, 4, 5, 0) if preset == "tiny11" and S <= 32 else (2, 3, 4, 5) if preset == "base" else (2, 4, 5)):
                forms = expected_forms(acc, limit, group, signed, tight if tight_here else 0)
                ctx.set_decode_mode(mode)
                got = ctx.translate(ids, lens, sl, want_align=True)
                assert all(np.array_equal(a, b) for a, b in zip(got, want)), (limit, tight, mode)
                seen = ctx.debug_kv_formats(m.dec_layers, B)
                assert seen is not None and np.array_equal(seen, forms), (limit, tight, mode, seen, forms)
                seen_forms |= set(np.unique(seen).tolist())
        assert seen_forms == ({0, 1, 2} if tight_here else {0, 1}), seen_forms
        assert expected_forms(acc, limits[1], group).any() and not expected_forms(acc, 2 ** 19, group).any()
        # the other cache formats record nothing and give the same results
        for fmt in (2, 1):
            gm.set_kv_cache_format(fmt)
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), fmt
            assert ctx.debug_kv_formats(m.dec_layers, B) is None
    finally:
        ctx.close()
        gm.close()


@pytest.mark.parametrize("rows", [64, 32])
def test_an_accumulator_past_2_19_sends_its_sentences_to_the_24_bit_form(hip, oracle, rows):
    """A model whose first K column of decoder layer 1 and first V column of layer 2 sum weights of one sign: 127 colsum
    alone is close to 2^19 there, and the data-dependent part decides sentence by sentence. The checker's accumulators say
    which sentences cross the limit; the device must agree, keep the others narrow, and translate all of them exactly."""
    from slimt_amd import synth
    B, S = 41, 32
    m = synth.make_model("tiny11", seed=1234, eos_bias=6.0)
    ids, lens = synth.make_batch(m.V, B, S, seed=777, ragged=True)
    om0 = oracle.OracleModel(m)
    base = kv_accumulators(oracle, m, om0, ids, lens)
    del om0
    # column 0 <- `k1` weights of +127 in front (the rest as drawn): accS[.., 0] ~ 127 * 127 * k1 + data; pick k1 so that
    # some but not all groups of sentences cross 2^19 (the encoder output does not depend on these weights)
    group = rows // S
    picked = {}
    oracle.set_mode(oracle.PORTABLE)
    om_enc = oracle.OracleModel(m)
    enc = np.ascontiguousarray(om_enc.encode(om_enc.embed(ids), oracle.make_mask(lens, S)).reshape(B * S, m.D), dtype=np.float32)
    del om_enc
    oracle.set_mode(oracle.FAITHFUL)
    for l, t, name in ((0, 0, "k"), (1, 1, "v")):
        W = m.params[f"decoder_l{l + 1}_context_W{name}"]
        data = np.ascontiguousarray(W.data).reshape(m.D, m.D).copy()  # payload [N][K]
        aq = float(np.asarray(m.params[f"decoder_l{l + 1}_context_W{name}_QuantMultA"].data).ravel()[0])
        best = None
        for k1 in range(16, 64):
            col = data[0].copy()
            col[:k1] = 127
            a0 = oracle.affine_acc(enc, col[None, :].copy(), aq).reshape(B, S).astype(np.int64)
            over = ((a0 < -2 ** 19) | (a0 >= 2 ** 19)).any(axis=1)
            frac = np.mean([over[s:s + group].any() for s in range(0, B, group)])
            if 0.15 <= frac <= 0.85 and (best is None or abs(frac - 0.5) < abs(best[1] - 0.5)):
                best = (k1, frac, col)
        assert best is not None, "no prefix length splits this batch"
        picked[(l, t)] = best[:2]
        data[0] = best[2]
        W.data = data.astype(np.int8)
    gm, om = hip.Model(m), oracle.OracleModel(m)
    ctx = hip.Context(gm, B, S)
    try:
        acc = kv_accumulators(oracle, m, om, ids, lens)
        assert np.array_equal(acc[:, :, :, :, 1:], base[:, :, :, :, 1:])  # only column 0 moved
        forms = expected_forms(acc, 2 ** 19, group)
        assert forms.any() and not forms.all(), picked
        # Where the tight form is tried (64-row encoder) it holds the accumulator less its column's
        # centre -- here 127 colsum, i.e. the SIGNED accumulator: a sentence whose shifted accumulator passes 2^19 through
        # 127 colsum alone still fits int16 -- the 24-bit form is then only for those whose data-dependent part is large too
        centres = colsum_centres(m)
        gm.set_kv_centres(centres)
        forms16 = expected_forms(acc, 2 ** 19, group, centred(acc, centres), 2 ** 15)
        assert (forms16[forms == 0] != 1).all()
        assert np.abs(acc.astype(np.int64)).max() < 2 ** 23
        sl = synth.make_shortlist(m.V, 640)
        oracle.set_mode(oracle.PORTABLE)
        want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
        oracle.set_mode(oracle.FAITHFUL)
        ctx.set_encode_rows(rows)
        for mode in (2, 3, 4, 5):
            ctx.set_decode_mode(mode)
            for policy in (1, 2):
                gm.set_kv_cache_policy(policy)
                got = ctx.translate(ids, lens, sl, want_align=True)
                assert all(np.array_equal(a, b) for a, b in zip(got, want)), (mode, policy)
                assert np.array_equal(ctx.debug_kv_formats(m.dec_layers, B), forms16), (mode, policy)
    finally:
        ctx.close()
        gm.close()


def test_narrow_limit_is_bounded(hip, synth_models):
    m = synth_models("tiny11", 6.0)
    gm = hip.Model(m)
    try:
        for bad in (0, -1, 2 ** 19 + 1, 2 ** 23):
            with pytest.raises(hip.SlimtHipError):
                gm.debug_kv_narrow_limit(bad)
        gm.debug_kv_narrow_limit(2 ** 19)
        for bad in (-1, 2 ** 15 + 1):
            with pytest.raises(hip.SlimtHipError):
                gm.debug_kv_tight_limit(bad)
        gm.debug_kv_tight_limit(0)
        gm.debug_kv_tight_limit(2 ** 15)
        n = m.dec_layers * 2 * m.D
        for bad in (np.full(n, 2 ** 24, dtype=np.int32), np.full(n, -2 ** 24, dtype=np.int32), np.zeros(n - 1, dtype=np.int32)):
            with pytest.raises(hip.SlimtHipError):
                gm.set_kv_centres(bad)
        assert gm.debug_kv_centres(m.dec_layers, m.D) is None
        gm.set_kv_centres(np.full(n, 2 ** 24 - 1, dtype=np.int32))
        assert (gm.debug_kv_centres(m.dec_layers, m.D) == 2 ** 24 - 1).all()
    finally:
        gm.close()


@pytest.mark.parametrize("B,S", [(20, 9), (33, 4), (18, 12), (25, 3)])
def test_lengths_whose_narrow_v_block_would_not_fit_keep_the_24_bit_form(hip, oracle, synth_models, B, S):
    """V is cached in groups of eight keys in the narrow form and of four in the 24-bit one: for S = 1..4 and 9..12 a
    sentence's narrow block would be larger than its slot, so those lengths keep the 24-bit form (nothing is recorded)."""
    from slimt_amd import synth
    m = synth_models("tiny11", 6.0)
    gm, om = hip.Model(m), oracle.OracleModel(m)
    ctx = hip.Context(gm, B, S)
    try:
        ids, lens = synth.make_batch(m.V, B, S, seed=5100 + S, ragged=True)
        sl = synth.make_shortlist(m.V, 640)
        oracle.set_mode(oracle.PORTABLE)
        want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
        oracle.set_mode(oracle.FAITHFUL)
        for rows in (64, 32):
            ctx.set_encode_rows(rows)
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), rows
            assert ctx.debug_kv_formats(m.dec_layers, B) is None
    finally:
        ctx.close()
        gm.close()


@pytest.mark.parametrize("preset", ["tiny11", "base"])
def test_a_model_that_mostly_needs_24_bits_is_switched_to_them(hip, oracle, synth_models, preset):
    """Format 0 watches its own fallback: a sentence in the 24-bit form is read through an out-of-line call that its whole
    decoder workgroup waits for, so a model whose accumulators mostly do not fit 20 bits (here: every sentence, by a
    narrow-form limit of 1) is switched to the 24-bit form for every batch once 1024 sentence-layers have shown it
    (include/slimt_hip.h, slimt_hip_debug_kv_watch). Results never change; a model that fits stays narrow."""
    from slimt_amd import synth
    m = synth_models(preset, 6.0)
    gm, om = hip.Model(m), oracle.OracleModel(m)
    B, S = 48, 24
    ctx = hip.Context(gm, B, S)
    try:
        gm.debug_kv_tight_limit(0)  # (this test is about the 20- / 24-bit pair: no tight form, no calibration batch)
        ids, lens = synth.make_batch(m.V, B, S, seed=808, ragged=True)
        sl = synth.make_shortlist(m.V, 640)
        oracle.set_mode(oracle.PORTABLE)
        want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
        oracle.set_mode(oracle.FAITHFUL)
        # a model that fits: stays narrow however long it runs
        for _ in range(14):
            got = ctx.translate(ids, lens, sl, want_align=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        switched, wide, total = gm.debug_kv_watch()
        assert not switched and wide == 0 and total == 14 * B * m.dec_layers
        assert ctx.debug_kv_formats(m.dec_layers, B) is not None
        # nothing fits: the watch trips after 1024 sentence-layers (the device's count lags a batch or two)
        gm.debug_kv_narrow_limit(1)
        n = 0
        while not gm.debug_kv_watch()[0]:
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), n
            n += 1
            assert n <= 40, gm.debug_kv_watch()
        assert n * B * m.dec_layers >= 1024
        switched, wide, total = gm.debug_kv_watch()
        assert switched and wide * 32 > total
        for _ in range(3):  # from now on every batch is cached in the 24-bit form, nothing is recorded
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want))
            assert ctx.debug_kv_formats(m.dec_layers, B) is None
        # choosing a format (or a limit) starts the watch afresh
        gm.set_kv_cache_format(0)
        gm.debug_kv_narrow_limit(2 ** 19)
        got = ctx.translate(ids, lens, sl, want_align=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        assert not gm.debug_kv_watch()[0] and ctx.debug_kv_formats(m.dec_layers, B) is not None
    finally:
        ctx.close()
        gm.close()


def test_centres_are_calibrated_from_the_first_large_batch_and_a_layer_that_mostly_misses_stops_trying(hip, oracle, synth_models):
    """Without centres from the caller the library calibrates them (include/slimt_hip.h, slimt_hip_model_set_kv_centres): the
    first batch of >= 1024 rows that could take the tight form is cached as f32 and its column means -- floor(sum / rows +
    1/2) over every row of the batch, in integers -- become the centres. On the synthetic model the column means carry most
    of the accumulators' spread (7.5 k against 3.2 k around them), so every sentence fits int16 afterwards.
    The tight form has its own watch, per decoder layer (slimt_hip_debug_kv_tight_watch): a sentence that misses is read
    through an out-of-line call, so a layer where more than one in 32 of 1024 sentences missed (here: all of them, by a tight
    limit of 1) goes back to starting with the 20-bit form. Results never change."""
    from slimt_amd import synth
    m = synth_models("tiny11", 6.0)
    gm, om = hip.Model(m), oracle.OracleModel(m)
    B, S = 64, 32  # (2048 rows; the 64-row encoder: 32 workgroups)
    ctx = hip.Context(gm, B, S)
    try:
        ids, lens = synth.make_batch(m.V, B, S, seed=909, ragged=True)
        sl = synth.make_shortlist(m.V, 640)
        oracle.set_mode(oracle.PORTABLE)
        want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
        oracle.set_mode(oracle.FAITHFUL)
        acc = kv_accumulators(oracle, m, om, ids, lens)
        assert gm.debug_kv_centres(m.dec_layers, m.D) is None
        got = ctx.translate(ids, lens, sl, want_align=True)  # the calibration batch: an f32 cache
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        assert ctx.debug_kv_formats(m.dec_layers, B) is None
        centres = gm.debug_kv_centres(m.dec_layers, m.D)
        sums = acc.astype(np.int64).sum(axis=(2, 3))  # [Ld][2][D]
        assert centres is not None and np.array_equal(centres, (2 * sums + B * S) // (2 * B * S))
        forms = expected_forms(acc, 2 ** 19, 2, centred(acc, centres), 2 ** 15)
        assert (forms == 2).all()  # the synthetic model around its column means
        assert (expected_forms(acc, 2 ** 19, 2, centred(acc, colsum_centres(m)), 2 ** 15) != 2).any()  # ... not around 127 colsum
        for _ in range(19):
            got = ctx.translate(ids, lens, sl, want_align=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        off, missed, tried = gm.debug_kv_tight_watch()
        assert off == 0 and tried[:2] == [19 * B, 19 * B] and missed[:2] == [0, 0]
        assert np.array_equal(ctx.debug_kv_formats(m.dec_layers, B), forms)
        with pytest.raises(hip.SlimtHipError):
            gm.set_kv_centres(np.zeros(5, dtype=np.int32))
        gm.debug_kv_recalibrations(0)  # (no new generation of centres: the end of the road is what this part is about; the
        gm.debug_kv_tight_limit(1)     #  re-calibration has its own test below)
        n = 0
        while gm.debug_kv_tight_watch()[0] != 3:
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), n
            n += 1
            assert n <= 40, gm.debug_kv_tight_watch()
        assert n * B >= 1024
        for _ in range(2):
            got = ctx.translate(ids, lens, sl, want_align=True)
            assert all(np.array_equal(a, b) for a, b in zip(got, want))
            assert not ctx.debug_kv_formats(m.dec_layers, B).any()  # every sentence-layer in the 20-bit form, first try
        _, missed, tried = gm.debug_kv_tight_watch()
        assert missed[:2] == tried[:2]  # (nobody tried any more)
    finally:
        ctx.close()
        gm.close()


def test_centres_that_do_not_fit_the_traffic_are_replaced_not_fatal(hip, oracle, synth_models):
    """VERDICT r05 item 4(ii). The tight form's centres are the column means of the FIRST batch of >= 1024 rows (or set by
    the caller); when they do not fit the traffic behind them -- here the caller sets 127 colsum, under which the form
    holds the signed accumulator and about one sentence in sixteen misses it -- the layer's watch trips (more than 1 in
    32 of >= 1024 sentences). Round 5 then switched the layer off for good. Now the first trips start a new GENERATION
    of centres, calibrated from the next large batch (slimt_hip_debug_kv_recalibrations): the form comes back for
    (nearly) every sentence, batches in flight keep the generation they were encoded with, results stay the checker's
    throughout, and only trips past the limit switch a layer off."""
    from slimt_amd import synth
    m = synth_models("tiny11", 6.0)
    om = oracle.OracleModel(m)
    S, B = 32, 64
    sl = synth.make_shortlist(m.V, 1024)

    def run(gm, n_batches, check_every):
        ctx = hip.Context(gm, B, S)
        shares, gens = [], []
        oracle.set_mode(oracle.PORTABLE)
        for i in range(n_batches):
            ids, lens = synth.make_batch(m.V, B, S, seed=9000 + i, ragged=True)
            out, ln, al = ctx.translate(ids, lens, sl, want_align=True)
            if check_every and (i % check_every == 0 or i >= n_batches - 3):
                w_out, w_ln, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
                assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out) and np.array_equal(al, w_al), i
            f = ctx.debug_kv_formats(m.dec_layers, B)
            shares.append(0.0 if f is None else float((f == 2).mean()))
            gens.append(gm.debug_kv_recalibrations())
        oracle.set_mode(oracle.FAITHFUL)
        ctx.close()
        return shares, gens

    gm = hip.Model(m)
    try:
        gm.set_kv_centres(colsum_centres(m))
        assert gm.debug_kv_recalibrations() == 0
        shares, gens = run(gm, 44, 8)
        off, missed, sub = gm.debug_kv_tight_watch()
        assert gens[-1] == 1, (gens, shares)               # noticed once, replaced once
        first = gens.index(1)
        assert 16 <= first <= 26, gens                     # 1024 sentences = 16 batches of 64 (+ the device counters' lag)
        assert 0.3 < np.mean(shares[:first]) < 0.97, shares  # under the set centres a good share of the workgroups missed
        assert off == 0 and np.mean(shares[-8:]) > 0.97, (off, shares, gens)  # calibrated from the traffic: the form is back
        got = gm.debug_kv_centres(m.dec_layers, m.D)
        assert got is not None and not np.array_equal(got, colsum_centres(m))
    finally:
        gm.close()
    # a model that may not re-calibrate switches the layers off instead (round 5's behaviour: still the end of the road)
    gm2 = hip.Model(m)
    try:
        gm2.set_kv_centres(colsum_centres(m))
        gm2.debug_kv_recalibrations(0)
        shares2, gens2 = run(gm2, 30, 0)
        off2, _, _ = gm2.debug_kv_tight_watch()
        # (per layer: a layer whose sentences mostly fit under these centres goes on trying)
        assert off2 != 0 and gens2[-1] == 0 and shares2[-1] <= 0.5
    finally:
        gm2.close()
