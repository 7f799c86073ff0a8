"""GPU parity of slimt_hip_translate_many_* (include/slimt_hip.h): several batches of one padded source length in ONE
encoder and ONE decoder launch -- what `workers` concurrent Model::forward calls are in the reference
(slimt/Frontend.cc:207-227, Batcher.cc:95-120). Every batch's tokens, lengths and alignment rows must equal those of its
own unmerged call AND the CPU oracle's (PORTABLE order): bit for bit, with staggered EOS, a short last batch, batch sizes
that leave holes between the sub-batches, one shared or several distinct shortlists, the full vocabulary, every decoder
tiling and every encoder."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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


def _want(oracle, om, ids, lens, sl):
    oracle.set_mode(oracle.PORTABLE)
    out = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    return out


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def _run_many_device(hip, gm, batches, S, sls, mode=0, enc_rows=0, want_align=True, max_rows=None):
    """batches: [(ids, lens)] -- each padded to its own length <= S --, sls: per batch a shortlist array or None.
    Returns per batch (out, len, align)."""
    T_launch = max(int(np.float32(1.5) * np.float32(S)), 1)
    rows = hip.translate_many_rows([b[0].shape[0] for b in batches])
    ctx = hip.Context(gm, max_rows or rows, S)
    ctx.set_decode_mode(mode)
    if enc_rows:
        ctx.set_encode_rows(enc_rows)
    keep, args, outs = [], [], []
    sl_dev = {}
    for (ids, lens), sl in zip(batches, sls):
        B, Sj = ids.shape
        T = max(int(np.float32(1.5) * np.float32(Sj)), 1)
        d_ids, d_len = _dev(ids), _dev(lens)
        if sl is not None and id(sl) not in sl_dev:
            sl_dev[id(sl)] = _dev(sl)
        d_sl = sl_dev.get(id(sl))
        d_out = torch.full((B, T), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
        d_ol = torch.full((B,), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
        d_al = torch.full((B, T, Sj), 7.25, dtype=torch.float32, device="cuda") if want_align else None
        keep.append((d_ids, d_len, d_sl))
        outs.append((d_out, d_ol, d_al))
        args.append((d_ids.data_ptr(), d_len.data_ptr(), B, d_sl.data_ptr() if d_sl is not None else 0,
                     0 if sl is None else sl.size, d_out.data_ptr(), d_ol.data_ptr(), d_al.data_ptr() if want_align else 0, Sj))
    ctx.translate_many_device(args, S, 1.5, 0, steps_hint=T_launch)
    ctx.synchronize()
    res = [(o.cpu().numpy().view(np.uint32), l.cpu().numpy().view(np.uint32), None if a is None else a.cpu().numpy())
           for o, l, a in outs]
    ctx.close()
    return res


def _check(oracle, om, batches, sls, res):
    for j, ((ids, lens), sl, (out, ln, al)) in enumerate(zip(batches, sls, res)):
        w_out, w_ln, w_al = _want(oracle, om, ids, lens, sl)
        assert np.array_equal(ln, w_ln), (j, ln, w_ln)
        assert np.array_equal(out, w_out), j
        if al is not None:
            assert np.array_equal(al, w_al), j


# sizes: multiples of 32 (no holes), sizes that leave holes, a short last batch, one batch alone (k = 1: the plain call)
@pytest.mark.parametrize("preset,S,sizes,n_sl,mode,enc_rows", [
    ("tiny11", 32, [64, 64, 64, 64], 2048, 0, 0),      # BASELINE config 2's batches, four per launch
    ("tiny11", 32, [64, 64, 64, 64], 2048, 2, 64),     # 16 sentences per decoder workgroup, 64-row encoder
    ("tiny11", 32, [64, 64, 10], 1024, 0, 0),          # a short last batch
    ("tiny11", 24, [33, 5, 47, 16, 1, 20, 31, 64], 1024, 0, 0),  # eight batches, holes everywhere
    ("tiny11", 13, [19, 45], 1024, 4, 32),             # 8 sentences per workgroup, 32-row encoder
    ("tiny11", 13, [19, 45], 1024, 5, 0),              # 4 sentences per workgroup
    ("tiny11", 20, [40, 24], 0, 3, 0),                 # the full vocabulary, 32 sentences per workgroup
    ("tiny11", 40, [7, 9, 3], 1024, 0, 0),             # 33..64 tokens: one sentence per encoder workgroup
    ("tiny11", 32, [50], 1024, 0, 0),                  # k = 1
    ("base", 32, [19, 32, 6], 1024, 0, 0),             # D = 512
    ("base", 10, [6, 40], 512, 0, 0),
])
def test_merged_batches_equal_their_own_calls_and_the_oracle(hip, oracle, engines, preset, S, sizes, n_sl, mode, enc_rows):
    from slimt_amd import synth
    m, gm, om = engines(preset, 6.0)  # eos_bias 6: sentences end at different steps
    sl = synth.make_shortlist(m.V, n_sl) if n_sl else None
    batches = [synth.make_batch(m.V, B, S, seed=1000 * j + 7 * B + S, ragged=True) for j, B in enumerate(sizes)]
    sls = [sl] * len(sizes)
    res = _run_many_device(hip, gm, batches, S, sls, mode, enc_rows)
    _check(oracle, om, batches, sls, res)
    # ... and the unmerged call of each batch gives the same (the oracle check above implies it; this is the direct statement)
    ctx = hip.Context(gm, max(sizes), S)
    ctx.set_decode_mode(mode)
    for (ids, lens), (out, ln, al) in zip(batches, res):
        o1, l1, a1 = ctx.translate(ids, lens, sl, want_align=True)
        assert np.array_equal(o1, out) and np.array_equal(l1, ln) and np.array_equal(a1, al)
    ctx.close()


@pytest.mark.parametrize("preset,S,shapes,n_sl,mode", [
    ("tiny11", 32, [(40, 32), (33, 27), (64, 26), (9, 32)], 1024, 0),   # padded to fewer tokens than the launch
    ("tiny11", 24, [(20, 19), (50, 24), (31, 22)], 1024, 2),
    ("tiny11", 16, [(64, 13), (64, 16)], 0, 0),
    ("tiny11", 48, [(5, 40), (7, 48), (4, 35)], 512, 0),               # 33..64 tokens
    ("base", 32, [(19, 32), (12, 25)], 512, 0),
])
def test_merged_batches_with_their_own_padded_lengths(hip, oracle, engines, preset, S, shapes, n_sl, mode):
    """A batch padded to fewer tokens than the launch keeps the step limit and the alignment width of ITS length
    (Model.cc:159-161, 84-108): outputs == the oracle on that batch at its own padded length, and == its own call."""
    from slimt_amd import synth
    m, gm, om = engines(preset, 6.0)
    sl = synth.make_shortlist(m.V, n_sl) if n_sl else None
    batches = [synth.make_batch(m.V, B, Sj, seed=400 + 13 * j + B, ragged=True) for j, (B, Sj) in enumerate(shapes)]
    sls = [sl] * len(shapes)
    res = _run_many_device(hip, gm, batches, S, sls, mode)
    _check(oracle, om, batches, sls, res)
    ctx = hip.Context(gm, max(b for b, _ in shapes), S)
    ctx.set_decode_mode(mode)
    for (ids, lens), (out, ln, al) in zip(batches, res):
        o1, l1, a1 = ctx.translate(ids, lens, sl, want_align=True)
        assert np.array_equal(o1, out) and np.array_equal(l1, ln) and np.array_equal(a1, al)
    ctx.close()


def test_merged_batches_with_their_own_shortlists(hip, oracle, engines):
    """Model.cc:117-120: a batch's shortlist is ITS shortlist -- three batches, three lists of different sizes (two batches
    share one: one packed output layer for both)."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    S = 16
    slA, slB, slC = synth.make_shortlist(m.V, 1024, seed=1), synth.make_shortlist(m.V, 1536, seed=2), synth.make_shortlist(m.V, 520, seed=3)
    sizes = [33, 20, 64, 12]
    sls = [slA, slB, slA, slC]
    batches = [synth.make_batch(m.V, B, S, seed=50 + j, ragged=True) for j, B in enumerate(sizes)]
    for mode in (0, 2):
        res = _run_many_device(hip, gm, batches, S, sls, mode)
        _check(oracle, om, batches, sls, res)


def test_merged_launch_falls_back_batch_by_batch(hip, oracle, engines):
    """More than eight batches, sources past 64 tokens, a context too small for the merged rows: translated one by one,
    in order, on the same stream -- same results."""
    from slimt_amd import synth
    m, gm, om = engines("tiny11", 6.0)
    sl = synth.make_shortlist(m.V, 1024)
    batches = [synth.make_batch(m.V, 5 + j, 12, seed=90 + j, ragged=True) for j in range(10)]
    res = _run_many_device(hip, gm, batches, 12, [sl] * 10, max_rows=32)
    _check(oracle, om, batches, [sl] * 10, res)
    batches = [synth.make_batch(m.V, 3, 70, seed=70 + j, ragged=True) for j in range(2)]
    res = _run_many_device(hip, gm, batches, 70, [sl] * 2)
    _check(oracle, om, batches, [sl] * 2, res)
    batches = [synth.make_batch(m.V, 30, 12, seed=10 + j, ragged=True) for j in range(3)]
    res = _run_many_device(hip, gm, batches, 12, [sl] * 3, max_rows=40)  # 96 merged rows do not fit 40
    _check(oracle, om, batches, [sl] * 3, res)


@pytest.mark.parametrize("preset,S,sizes,n_sl", [("tiny11", 32, [64, 40, 64], 2048), ("tiny11", 13, [7, 33], None),
                                                   ("base", 32, [19, 19], 1024), ("tiny11", 30, [(20, 26), (33, 30), (8, 25)], 1024)])
def test_merged_pinned_async(hip, oracle, engines, preset, S, sizes, n_sl):
    """slimt_hip_translate_many_async: pinned host arrays per batch, read and written in place by the two launches;
    alignment rows staged in device memory and copied out per sentence (Model.cc:84-108)."""
    from slimt_amd import synth
    m, gm, om = engines(preset, 6.0)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl)
    sizes = [x if isinstance(x, tuple) else (x, S) for x in sizes]  # (B, this batch's own padded length)
    batches = [synth.make_batch(m.V, B, Sj, seed=300 + 11 * j + B, ragged=True) for j, (B, Sj) in enumerate(sizes)]
    ctx = hip.Context(gm, hip.translate_many_rows([b for b, _ in sizes]), S)
    pins, bufs = [], []
    for ids, lens in batches:
        B, Sj = ids.shape
        T = max(int(np.float32(1.5) * np.float32(Sj)), 1)
        ps = [hip._Pinned() for _ in range(5)]
        b = (ps[0].array(np.uint32, (B, Sj)), ps[1].array(np.uint32, (B,)), ps[2].array(np.uint32, (B, T)),
             ps[3].array(np.uint32, (B,)), ps[4].array(np.float32, (B, T, Sj)))
        b[0][...] = ids
        b[1][...] = lens
        pins.append(ps)
        bufs.append(b)
    for rep in range(2):
        for b in bufs:
            b[2][...] = 0x5a5a5a5a
            b[3][...] = 0x5a5a5a5a
            b[4][...] = np.float32(7.25)
        ctx.translate_many_async(bufs, sl)
        ctx.synchronize()
        res = [(b[2].copy(), b[3].copy(), b[4].copy()) for b in bufs]
        _check(oracle, om, batches, [sl] * len(sizes), res)
    ctx.close()
    for ps in pins:
        for p in ps:
            p.free()


def test_merged_headline_shape_full_size(hip, oracle, engines):
    """BASELINE config 2 at full size: four batches of 64 sentences of 32 tokens, shortlist 4096, merged -- against the
    unmerged calls (bit for bit) with nobody emitting EOS (T = 48 for all) and with staggered endings."""
    from slimt_amd import synth
    for eos_bias in (-100.0, 6.0):
        m, gm, om = engines("tiny11", eos_bias)
        sl = synth.make_shortlist(m.V, 4096)
        batches = [synth.make_batch(m.V, 64, 32, seed=4321 + j) for j in range(4)]
        res = _run_many_device(hip, gm, batches, 32, [sl] * 4)
        ctx = hip.Context(gm, 64, 32)
        for (ids, lens), (out, ln, al) in zip(batches, res):
            o1, l1, a1 = ctx.translate(ids, lens, sl, want_align=True)
            assert np.array_equal(o1, out) and np.array_equal(l1, ln) and np.array_equal(a1, al)
        ctx.close()
        if eos_bias > 0:
            _check(oracle, om, batches[:1], [sl], res[:1])


@pytest.mark.parametrize("preset,S,shapes,mode", [
    ("tiny11", 24, [(33, 24), (20, 20), (47, 24), (5, 19)], 0),
    ("tiny11", 16, [(64, 16), (64, 16), (10, 13)], 2),
    ("base", 32, [(19, 32), (12, 27)], 0),
    ("tiny11", 2, [(3, 2)] * 8, 0),   # eight shortlists, one encoder tile: the workgroup of tile 0 generates all of them
])
def test_merged_batches_generate_their_own_lexical_shortlists(hip, oracle, engines, preset, S, shapes, mode):
    """Model.cc:117-120 per batch, merged: every batch's output vocabulary is ShortlistGenerator::generate of ITS source
    words, produced by one of the first workgroups of the ONE encoder launch; tokens, lengths and alignment rows == the
    checker's on that batch with the checker's own shortlist for it -- and the lists really differ between the batches."""
    from slimt_amd import synth
    m, gm, om = engines(preset, 6.0)
    blob = synth.make_lexical_shortlist(m.V, m.V, 100, 2, seed=21, empty_fraction=0.3, min_count=1)
    osl = oracle.OracleShortlist(blob, m.V, m.V)
    gen = hip.ShortlistGenerator(blob, m.V, m.V)
    batches = [synth.make_batch(m.V, B, Sj, seed=700 + 17 * j + B, ragged=True) for j, (B, Sj) in enumerate(shapes)]
    sls = [osl.generate(ids, lens) for ids, lens in batches]
    assert len({tuple(s_) for s_ in sls}) > 1
    rows = hip.translate_many_rows([b for b, _ in shapes])
    ctx = hip.Context(gm, rows, S)
    ctx.set_decode_mode(mode)
    keep, args, outs = [], [], []
    for ids, lens in batches:
        B, Sj = ids.shape
        T = max(int(np.float32(1.5) * np.float32(Sj)), 1)
        d_ids, d_len = _dev(ids), _dev(lens)
        d_out = torch.full((B, T), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
        d_ol = torch.full((B,), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
        d_al = torch.full((B, T, Sj), 7.25, dtype=torch.float32, device="cuda")
        keep.append((d_ids, d_len))
        outs.append((d_out, d_ol, d_al))
        args.append((d_ids.data_ptr(), d_len.data_ptr(), B, 0, 0, d_out.data_ptr(), d_ol.data_ptr(), d_al.data_ptr(), Sj))
    for rep in range(2):
        ctx.translate_many_device(args, S, 1.5, 0, steps_hint=max(int(np.float32(1.5) * np.float32(S)), 1), generator=gen)
        ctx.synchronize()
        res = [(o.cpu().numpy().view(np.uint32), l.cpu().numpy().view(np.uint32), a.cpu().numpy()) for o, l, a in outs]
        _check(oracle, om, batches, sls, res)
    ctx.close()
    gen.close()
