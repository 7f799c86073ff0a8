"""The C++ host mirror of the reference interface (slimt_amd/host: slimt::qmm
provider, Marian .bin loader, Model/Worker::forward) against the oracle.
This is what a slimt maintainer's `WITH_HIP` build would exercise: C++ code
linked against libslimt_hip.so, no Python in the data path."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest


def _build_host():
    from slimt_amd import build
    return build.build_host()


def test_host_driver_builds_and_links():
    exe = _build_host()
    assert os.path.exists(exe) and os.access(exe, os.X_OK)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr


def test_cpp_bin_loader_rejects_garbage():
    exe = _build_host()
    with tempfile.TemporaryDirectory() as d:
        bad = os.path.join(d, "bad.bin")
        open(bad, "wb").write(struct.pack("<QQ", 7, 1))
        case = os.path.join(d, "case.bin")
        open(case, "wb").write(struct.pack("<9I3f", 1, 1, 4, 1, 1, 0, 1, 64, 8, 1.5, 1.0, 1.0) +
                               b"\0" * 4096)
        r = subprocess.run([exe, bad, case, os.path.join(d, "o.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "version" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("preset,eos_bias,B,S,n_sl", [("micro", 3.0, 6, 8, 128), ("tiny11", 6.0, 9, 14, 1024),
                                                      ("mini", 1.0, 5, 9, 0)])
def test_cpp_model_forward_and_qmm(hip, oracle, synth_models, preset, eos_bias, B, S, n_sl):
    from slimt_amd import synth
    exe = _build_host()
    m = synth_models(preset, eos_bias)
    ids, lens = synth.make_batch(m.V, B, S, seed=B + S, ragged=True)
    sl = synth.make_shortlist(m.V, n_sl) if n_sl else np.zeros(0, np.uint32)
    r = np.random.Generator(np.random.PCG64(3))
    M, K, N = 7, 128, 96
    x = r.normal(0, 2, size=(M, K)).astype(np.float32)
    W = r.integers(-127, 128, size=(N, K)).astype(np.int8)
    bias = r.normal(0, 0.05, size=N).astype(np.float32)
    aq, bq = np.float32(21.5), np.float32(180.25)
    idx = np.sort(r.choice(N, size=40, replace=False)).astype(np.uint32)
    with tempfile.TemporaryDirectory() as d:
        mb, cb, ob = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin"))
        open(mb, "wb").write(synth.write_bin(m))
        with open(cb, "wb") as f:
            f.write(struct.pack("<9I3f", m.enc_layers, m.dec_layers, m.H, B, S, sl.size, M, K, N,
                                1.5, aq, bq))
            for a in (ids, lens, sl, x, W, bias):
                f.write(np.ascontiguousarray(a).tobytes())
            f.write(struct.pack("<I", idx.size) + idx.tobytes())
        res = subprocess.run([exe, mb, cb, ob], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        raw = open(ob, "rb").read()
    oracle.set_mode(oracle.PORTABLE)
    om = oracle.OracleModel(m)
    w_out, w_ln, w_al, _ = om.translate(ids, lens, sl if n_sl else None, 1.5, 0, want_align=True)
    off = 0
    for b in range(B):
        (n,) = struct.unpack_from("<I", raw, off)
        off += 4
        toks = np.frombuffer(raw, np.uint32, n, off)
        off += 4 * n
        L = int(lens[b])
        al = np.frombuffer(raw, np.float32, n * L, off).reshape(n, L)
        off += 4 * n * L
        assert n == w_ln[b] and np.array_equal(toks, w_out[b, :n])
        assert np.array_equal(al, w_al[b, :n, :L])
    y1 = np.frombuffer(raw, np.float32, M * N, off).reshape(M, N)
    off += 4 * M * N
    y2 = np.frombuffer(raw, np.float32, M * N, off).reshape(M, N)
    off += 4 * M * N
    y3 = np.frombuffer(raw, np.float32, M * idx.size, off).reshape(M, idx.size)
    off += 4 * M * idx.size
    assert off == len(raw)
    assert np.array_equal(y1, oracle.affine(x, W, bias, float(aq), float(bq)))
    assert np.array_equal(y2, oracle.affine(x, W, None, float(aq), float(bq)))
    assert np.array_equal(y3, oracle.affine_select(x, W, bias, float(aq), float(bq), idx))
    oracle.set_mode(oracle.FAITHFUL)


@pytest.mark.gpu
def test_cpp_shortlist_generator_then_forward(hip, oracle, synth_models):
    """The C++ mirror of slimt::ShortlistGenerator feeding Worker::forward, as
    Model::forward does (Model.cc:117-120): generated ids and translation both
    equal the oracle's."""
    from slimt_amd import synth
    exe = _build_host()
    m = synth_models("micro", 3.0)
    B, S = 6, 8
    ids, lens = synth.make_batch(m.V, B, S, seed=21, ragged=True)
    blob = synth.make_lexical_shortlist(m.V, m.V, frequent=16, best=5, seed=4)
    with tempfile.TemporaryDirectory() as d:
        mb, cb, ob, sb = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin", "lex.bin"))
        open(mb, "wb").write(synth.write_bin(m))
        open(sb, "wb").write(blob)
        x = np.zeros((1, 64), np.float32)
        W = np.zeros((16, 64), np.int8)
        bias = np.zeros(16, np.float32)
        idx = np.arange(8, dtype=np.uint32)
        with open(cb, "wb") as f:
            f.write(struct.pack("<9I3f", m.enc_layers, m.dec_layers, m.H, B, S, 0, 1, 64, 16, 1.5, 1.0, 1.0))
            for a in (ids, lens, np.zeros(0, np.uint32), x, W, bias):
                f.write(np.ascontiguousarray(a).tobytes())
            f.write(struct.pack("<I", idx.size) + idx.tobytes())
        res = subprocess.run([exe, mb, cb, ob, sb], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        raw = open(ob, "rb").read()
    want_sl = oracle.OracleShortlist(blob, m.V, m.V).generate(ids, lens)
    (n,) = struct.unpack_from("<I", raw, 0)
    got_sl = np.frombuffer(raw, np.uint32, n, 4)
    assert np.array_equal(got_sl, want_sl)
    off = 4 + 4 * n
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, _, _ = oracle.OracleModel(m).translate(ids, lens, want_sl, 1.5, 0)
    oracle.set_mode(oracle.FAITHFUL)
    for b in range(B):
        (k,) = struct.unpack_from("<I", raw, off)
        off += 4
        toks = np.frombuffer(raw, np.uint32, k, off)
        off += 4 * k + 4 * k * int(lens[b])
        assert k == w_ln[b] and np.array_equal(toks, w_out[b, :k])



@pytest.mark.gpu
@pytest.mark.parametrize("preset,eos_bias,B,S,n_sl", [("micro", 3.0, 6, 8, 128), ("tiny11", 6.0, 9, 14, 1024),
                                                      ("tiny11", 8.0, 20, 32, 0), ("base", 6.0, 5, 11, 512)])
def test_cpp_transformer_classes_drive_model_forward(hip, oracle, synth_models, preset, eos_bias, B, S, n_sl):
    """host/Transformer.hh: Encoder::forward(embedding, mask), Decoder::start_states / step(encoder_out,
    mask, states, previous, shortlist), greedy_sample*, transform_embedding, index_select with the
    reference's signatures (Transformer.hh:15-72), driven like Model::forward + Model::decode
    (Model.cc:111-204): encoder output, tokens and alignment rows equal the oracle's."""
    from slimt_amd import synth
    exe = _build_host()
    m = synth_models(preset, eos_bias)
    ids, lens = synth.make_batch(m.V, B, S, seed=B + S + 40, ragged=True)
    sl = synth.make_shortlist(m.V, n_sl) if n_sl else np.zeros(0, np.uint32)
    with tempfile.TemporaryDirectory() as d:
        mb, cb, ob = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin"))
        open(mb, "wb").write(synth.write_bin(m))
        with open(cb, "wb") as f:
            f.write(struct.pack("<9I3f", m.enc_layers, m.dec_layers, m.H, B, S, sl.size, 0, 0, 0, 1.5, 1.0, 1.0))
            for a in (ids, lens, sl):
                f.write(np.ascontiguousarray(a).tobytes())
        res = subprocess.run([exe, "--transformer", mb, cb, ob], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr
        raw = open(ob, "rb").read()
    oracle.set_mode(oracle.PORTABLE)
    om = oracle.OracleModel(m)
    mask = oracle.make_mask(lens, S)
    want_enc = om.encode(om.embed(ids), mask)
    w_out, w_ln, w_al, _ = om.translate(ids, lens, sl if n_sl else None, 1.5, 0, want_align=True)
    oracle.set_mode(oracle.FAITHFUL)
    enc = np.frombuffer(raw, np.float32, B * S * m.D, 0).reshape(B, S, m.D)
    assert np.array_equal(enc, want_enc)
    off = 4 * B * S * m.D
    for b in range(B):
        (n,) = struct.unpack_from("<I", raw, off)
        off += 4
        toks = np.frombuffer(raw, np.uint32, n, off)
        off += 4 * n
        L = int(lens[b])
        al = np.frombuffer(raw, np.float32, n * L, off).reshape(n, L)
        off += 4 * n * L
        assert n == w_ln[b] and np.array_equal(toks, w_out[b, :n]), b
        assert np.array_equal(al, w_al[b, :n, :L]), b
    assert off == len(raw)


@pytest.mark.gpu
@pytest.mark.parametrize("preset,eos_bias,lexical,threads", [("micro", 3.0, False, 1), ("tiny11", 6.0, True, 4),
                                                             ("tiny11", 6.0, False, 3), ("base", 6.0, True, 2)])
def test_cpp_model_forward_const_is_reentrant(hip, oracle, synth_models, preset, eos_bias, lexical, threads):
    """slimt::Model::forward(const Input &) const (slimt/Model.hh:56, Model.cc:187-204) as the
    reference's Async workers call it: several threads on ONE const Model, batches of growing and
    shrinking shapes (a pooled context is rebuilt when a batch outgrows it), the shortlist generated
    per batch when the model was given a lexical shortlist -- every sentence's tokens and alignment
    rows equal the oracle's, and no more contexts were built than threads ran."""
    import re
    from slimt_amd import synth
    exe = _build_host()
    m = synth_models(preset, eos_bias)
    shapes = [(5, 9), (12, 21), (3, 40), (20, 16), (1, 1), (7, 33), (16, 8), (2, 70 if preset != "base" else 30)]
    batches = [synth.make_batch(m.V, B, S, seed=600 + i, ragged=True) for i, (B, S) in enumerate(shapes)]
    blob = synth.make_lexical_shortlist(m.V, m.V, 100 if m.V > 200 else 16, 30 if m.V > 200 else 6, seed=8) if lexical else None
    with tempfile.TemporaryDirectory() as d:
        mb, cb, ob, sb = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin", "lex.bin"))
        open(mb, "wb").write(synth.write_bin(m))
        with open(cb, "wb") as f:
            f.write(struct.pack("<5If", m.enc_layers, m.dec_layers, m.H, threads, len(batches), 1.5))
            for ids, lens in batches:
                f.write(struct.pack("<2I", *ids.shape) + ids.tobytes() + lens.tobytes())
        args = [exe, "--model-forward", mb, cb, ob]
        if lexical:
            open(sb, "wb").write(blob)
            args.append(sb)
        res = subprocess.run(args, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr
        raw = open(ob, "rb").read()
    built = int(re.search(r"contexts-built: (\d+)", res.stderr).group(1))
    assert 1 <= built <= threads + len(shapes)  # one per concurrent caller + rebuilds for larger batches
    osl = oracle.OracleShortlist(blob, m.V, m.V) if lexical else None
    om = oracle.OracleModel(m)
    oracle.set_mode(oracle.PORTABLE)
    off = 0
    try:
        for ids, lens in batches:
            sl = osl.generate(ids, lens) if lexical else None
            w_out, w_ln, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
            for b in range(ids.shape[0]):
                (n,) = struct.unpack_from("<I", raw, off)
                off += 4
                toks = np.frombuffer(raw, np.uint32, n, off)
                off += 4 * n
                L = int(lens[b])
                al = np.frombuffer(raw, np.float32, n * L, off).reshape(n, L)
                off += 4 * n * L
                assert n == w_ln[b] and np.array_equal(toks, w_out[b, :n])
                assert np.array_equal(al, w_al[b, :n, :L])
    finally:
        oracle.set_mode(oracle.FAITHFUL)
    assert off == len(raw)
