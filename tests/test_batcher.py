"""Batching pipeline (what slimt does in Batcher.{hh,cc} and
Frontend.cc:207-227; here slimt_amd/host/Service.{hh,cc}, an own multi-device queue).

CPU: the C++ LengthQueue forms the same batches as a Python restatement of the
reference's batch-forming rule (Batcher::generate) -- padding is visible in results.
GPU: the Service's double-buffered workers (token-budget batches, pinned staging,
optional shortlist) -- every sentence's translation equals the oracle's for that
sentence padded to the length of the batch it travelled in (rows never interact)."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from slimt_amd import build as B, synth


def _exe():
    return B.build_host()


def _case(path, dims, max_words, wrap, workers, limit, requests, shortlist=None):
    with open(path, "wb") as f:
        f.write(struct.pack("<7If", *dims, max_words, wrap, workers, len(requests), limit))
        for segs in requests:
            f.write(struct.pack("<I", len(segs)))
            for s in segs:
                f.write(struct.pack("<I", len(s)) + np.asarray(s, np.uint32).tobytes())
        if shortlist is not None:
            f.write(struct.pack("<I", len(shortlist)) + np.asarray(shortlist, np.uint32).tobytes())


def _requests(V, n_req, seed, max_len):
    r = np.random.Generator(np.random.PCG64(seed))
    reqs = []
    for _ in range(n_req):
        segs = []
        for _ in range(int(r.integers(1, 9))):
            n = int(r.integers(1, max_len + 1))
            segs.append(np.concatenate([r.integers(2, V, size=n - 1), [0]]).astype(np.uint32))
        reqs.append(segs)
    return reqs


@pytest.mark.parametrize("max_words,wrap,limit,seed", [(64, 16, 1.5, 1), (1024, 128, 1.5, 2), (40, 32, 1.0, 3)])
def test_batcher_matches_restatement(oracle, max_words, wrap, limit, seed):
    reqs = _requests(1000, 25, seed, min(wrap, max_words))
    with tempfile.TemporaryDirectory() as d:
        cb, ob = os.path.join(d, "case.bin"), os.path.join(d, "out.bin")
        _case(cb, (1, 1, 1), max_words, wrap, 1, limit, reqs)
        res = subprocess.run([_exe(), "--batcher", cb, ob], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr
        raw = open(ob, "rb").read()
    got, off = [], 0
    while off < len(raw):
        n, ml = struct.unpack_from("<2I", raw, off)
        off += 8
        refs = [tuple(struct.unpack_from("<2I", raw, off + 8 * i)) for i in range(n)]
        off += 8 * n
        assert ml == max(len(reqs[r][i]) for r, i in refs)
        assert n * ml <= max_words  # Batcher.cc:103-104
        got.append(refs)
    want = oracle.batcher_generate([[len(s) for s in segs] for segs in reqs], max_words, wrap, limit)
    assert got == want
    flat = sorted(x for b in got for x in b)
    assert flat == sorted((r, i) for r, segs in enumerate(reqs) for i in range(len(segs)))


def test_batcher_rejects_wrap_longer_than_budget():
    with tempfile.TemporaryDirectory() as d:
        cb, ob = os.path.join(d, "case.bin"), os.path.join(d, "out.bin")
        _case(cb, (1, 1, 1), 8, 16, 1, 1.5, [[np.zeros(3, np.uint32)]])
        res = subprocess.run([_exe(), "--batcher", cb, ob], capture_output=True, text=True, timeout=120)
        assert res.returncode == 1 and "wrap_length > max_words" in res.stderr  # Batcher.cc:89-91


@pytest.mark.gpu
@pytest.mark.parametrize("workers,n_sl", [(1, None), (3, None), (4, 128)])
def test_async_workers_translate_every_sentence(hip, oracle, synth_models, workers, n_sl):
    m = synth_models("micro", 3.0)
    reqs = _requests(m.V, 12 if workers < 4 else 40, 7 + workers, 20)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl, frequent=16)
    with tempfile.TemporaryDirectory() as d:
        mb, cb, ob = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin"))
        open(mb, "wb").write(synth.write_bin(m))
        _case(cb, (m.enc_layers, m.dec_layers, m.H), 96, 24, workers, 1.5, reqs, sl)
        env = dict(os.environ, SLIMT_SERVICE_REPEAT="1")  # a second, warm pass over the same workers
        res = subprocess.run([_exe(), "--async", mb, cb, ob], capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stderr
        assert "async-warm" in res.stderr
        raw = open(ob, "rb").read()
    oracle.set_mode(oracle.PORTABLE)
    om = oracle.OracleModel(m)
    off = 0
    for segs in reqs:
        for s in segs:
            S, n = struct.unpack_from("<2I", raw, off)
            off += 8
            toks = np.frombuffer(raw, np.uint32, n, off)
            off += 4 * n
            assert S >= len(s)
            ids = np.zeros((1, S), np.uint32)
            ids[0, : len(s)] = s
            w_out, w_ln, _, _ = om.translate(ids, np.array([len(s)], np.uint32), sl, 1.5, 0)
            assert n == w_ln[0] and np.array_equal(toks, w_out[0, :n])
    oracle.set_mode(oracle.FAITHFUL)
    assert off == len(raw)


@pytest.mark.gpu
def test_service_rejects_bad_requests_and_survives(hip, synth_models):
    """An empty or overlong sentence is refused at translate() (the engine would fail the
    whole batch); the workers keep running and later requests are translated."""
    m = synth_models("micro", 3.0)
    with tempfile.TemporaryDirectory() as d:
        mb = os.path.join(d, "model.bin")
        open(mb, "wb").write(synth.write_bin(m))
        res = subprocess.run([_exe(), "--service-errors", mb, str(m.enc_layers), str(m.dec_layers), str(m.H)],
                             capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr + res.stdout
    assert "rejected empty" in res.stdout and "rejected overlong" in res.stdout
    assert "worker failure reported" in res.stdout and "survived: 3 sentences" in res.stdout


def _run_async(m, reqs, workers, max_words=96, wrap=24, env_extra=None, sl=None, expect_rc=0):
    with tempfile.TemporaryDirectory() as d:
        mb, cb, ob = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin"))
        open(mb, "wb").write(synth.write_bin(m))
        _case(cb, (m.enc_layers, m.dec_layers, m.H), max_words, wrap, workers, 1.5, reqs, sl)
        env = dict(os.environ, SLIMT_SERVICE_DUMP_FULL="1")
        extra = dict(env_extra or {})
        if "SLIMT_SERVICE_LEXICAL_BLOB" in extra:  # bytes -> a file next to the case
            lb = os.path.join(d, "lex.bin")
            open(lb, "wb").write(extra.pop("SLIMT_SERVICE_LEXICAL_BLOB"))
            extra["SLIMT_SERVICE_LEXICAL"] = lb
        env.update(extra)
        res = subprocess.run([_exe(), "--async", mb, cb, ob], capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == expect_rc, res.stderr
        raw = open(ob, "rb").read() if os.path.exists(ob) else b""
    return res, raw


def _parse_full(raw, reqs):
    """-> list of (sentence, S, batch serial, tokens, alignment rows [n][len]) in request order."""
    out, off = [], 0
    for segs in reqs:
        for s in segs:
            S, n, batch, rows, ln = struct.unpack_from("<5I", raw, off)
            off += 20
            toks = np.frombuffer(raw, np.uint32, n, off)
            off += 4 * n
            al = np.frombuffer(raw, np.float32, rows * ln, off).reshape(rows, ln)
            off += 4 * rows * ln
            out.append((s, S, batch, toks, al))
    assert off == len(raw)
    return out


def _check_batches(oracle, om, rows, shortlist_of):
    """Rebuild every batch the Service formed (sentences grouped by batch serial, in the order it
    padded them: ascending length, arrival order inside a length == request order here) and compare
    tokens, lengths and alignment rows with the oracle on that batch and ITS shortlist."""
    batches = {}
    for i, (s, S, batch, toks, al) in enumerate(rows):
        batches.setdefault(batch, []).append(i)
    oracle.set_mode(oracle.PORTABLE)
    for members in batches.values():
        members.sort(key=lambda i: (len(rows[i][0]), i))
        S = rows[members[0]][1]
        assert all(rows[i][1] == S for i in members) and S == max(len(rows[i][0]) for i in members)
        ids = np.zeros((len(members), S), np.uint32)
        lens = np.zeros(len(members), np.uint32)
        for b, i in enumerate(members):
            ids[b, : len(rows[i][0])] = rows[i][0]
            lens[b] = len(rows[i][0])
        sl = shortlist_of(ids, lens)
        w_out, w_ln, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
        for b, i in enumerate(members):
            _, _, _, toks, al = rows[i]
            n = int(w_ln[b])
            assert len(toks) == n and np.array_equal(toks, w_out[b, :n])
            assert al.shape == (n, int(lens[b])) and np.array_equal(al, w_al[b, :n, : int(lens[b])])
    oracle.set_mode(oracle.FAITHFUL)
    return len(batches)


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 4])
def test_service_generates_a_lexical_shortlist_per_batch(hip, oracle, synth_models, workers):
    """ServiceConfig::lexical_shortlist: every batch's output vocabulary is ShortlistGenerator::generate
    of ITS source words (Model.cc:117-120), on the device; tokens and alignment rows == oracle with
    OracleShortlist.generate of the rebuilt batch."""
    m = synth_models("micro", 3.0)
    blob = synth.make_lexical_shortlist(m.V, m.V, 16, 6, seed=21)
    osl = oracle.OracleShortlist(blob, m.V, m.V)
    reqs = _requests(m.V, 30, 50 + workers, 20)
    res, raw = _run_async(m, reqs, workers, env_extra={"SLIMT_SERVICE_LEXICAL_BLOB": blob, "SLIMT_SERVICE_REPEAT": "1"})
    rows = _parse_full(raw, reqs)
    n_batches = _check_batches(oracle, oracle.OracleModel(m), rows, lambda ids, lens: osl.generate(ids, lens))
    assert n_batches >= 3
    # the shortlists really differ between batches (else the test would not see a frozen one)
    seen = {tuple(osl.generate(np.asarray([s], np.uint32), np.asarray([len(s)], np.uint32))) for s, *_ in rows[:8]}
    assert len(seen) > 1


@pytest.mark.gpu
@pytest.mark.parametrize("preset,workers,merge,n_sl,max_len", [("micro", 2, "8", 128, 20), ("micro", 3, "3", None, 20),
                                                               ("tiny11", 2, "8", 1024, 12), ("micro", 2, "1", 128, 20)])
def test_service_merges_batches_of_one_length_into_one_launch(hip, oracle, synth_models, preset, workers, merge, n_sl, max_len):
    """ServiceConfig::merge_batches: a worker takes up to k consecutive batches of ONE padded length -- each formed by the
    reference's rule under max_words (Batcher.cc:95-120), each with its own arrays, results and serial number -- into one
    encoder + one decoder launch (slimt_hip_translate_many_async). Many sentences of few lengths, a small word budget: most
    launches are merged ones (the Service says how many). Every rebuilt batch == the oracle on that batch, alignment
    rows included; staggered EOS; the short last batch of a length travels with the others."""
    m = synth_models(preset, 3.0 if preset == "micro" else 6.0)
    r = np.random.Generator(np.random.PCG64(5 + workers))
    reqs = []
    for _ in range(10):  # ten requests of 40 sentences, lengths drawn from four values: long runs of one padded length
        segs = []
        for _ in range(40):
            n = int(r.choice([3, max_len // 2, max_len - 1, max_len]))
            segs.append(np.concatenate([r.integers(2, m.V, size=n - 1), [0]]).astype(np.uint32))
        reqs.append(segs)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl, frequent=16)
    res, raw = _run_async(m, reqs, workers, max_words=6 * max_len, wrap=max_len + 4, sl=sl,
                          env_extra={"SLIMT_SERVICE_MERGE": merge, "SLIMT_SERVICE_REPEAT": "1", "SLIMT_SERVICE_STATS": "1"})
    rows = _parse_full(raw, reqs)
    n_batches = _check_batches(oracle, oracle.OracleModel(m), rows, lambda ids, lens: sl)
    assert n_batches >= 40  # 400 sentences, at most 5..11 per batch
    import re
    mm = re.findall(r"merged launches: (\d+) of (\d+) launches, (\d+) batches", res.stderr)
    assert mm, res.stderr
    merged, launches, batches = (int(x) for x in mm[-1])
    if merge == "1":
        assert merged == 0 and launches == batches
    else:
        assert merged >= 5 and batches > launches


@pytest.mark.gpu
def test_service_two_replicas_on_one_device(hip, oracle, synth_models):
    """Two Model replicas (both on device 0), three workers each: the replica / worker assignment of
    Service (slimt/Frontend.cc:207-227's workers, one set per GPU) runs before an 8-GPU node does."""
    m = synth_models("micro", 3.0)
    sl = synth.make_shortlist(m.V, 128, frequent=16)
    reqs = _requests(m.V, 40, 91, 20)
    res, raw = _run_async(m, reqs, 3, env_extra={"SLIMT_SERVICE_REPLICAS": "2", "SLIMT_SERVICE_REPEAT": "1"}, sl=sl)
    rows = _parse_full(raw, reqs)
    assert _check_batches(oracle, oracle.OracleModel(m), rows, lambda ids, lens: sl) >= 3


@pytest.mark.gpu
def test_service_survives_workers_that_cannot_start(hip, oracle, synth_models):
    """Two of four workers fail their set-up: they retire, the other two translate everything
    (a failed worker used to fail most of the queue). All four failing: every
    request fails with the worker's error instead of hanging."""
    m = synth_models("micro", 3.0)
    reqs = _requests(m.V, 20, 17, 20)
    res, raw = _run_async(m, reqs, 4, env_extra={"SLIMT_SERVICE_FAIL_WORKERS": "2"})
    rows = _parse_full(raw, reqs)
    assert _check_batches(oracle, oracle.OracleModel(m), rows, lambda ids, lens: None) >= 2
    res, _ = _run_async(m, reqs, 4, env_extra={"SLIMT_SERVICE_FAIL_WORKERS": "4"}, expect_rc=1)
    assert "injected worker set-up failure" in res.stderr


@pytest.mark.gpu
def test_service_c_abi_returns_flat_results(hip, oracle, synth_models):
    """include/slimt_hip_service.h through capi.BatchService (what a binding calls): one request of
    tokenised sentences in, flat target ids / alignment rows / batch serials out == the oracle on
    every rebuilt batch; two replicas of the model, a lexical shortlist generated per batch."""
    m = synth_models("micro", 3.0)
    blob = synth.make_lexical_shortlist(m.V, m.V, 16, 6, seed=33)
    osl = oracle.OracleShortlist(blob, m.V, m.V)
    om = oracle.OracleModel(m)
    gm1, gm2 = hip.Model(m), hip.Model(m)
    svc = hip.BatchService([gm1, gm2], max_words=96, wrap_length=24, limit_factor=1.5, workers_per_device=2,
                           eos_id=0, alignments=True, lexical_shortlist=blob, source_vocab=m.V, target_vocab=m.V,
                           check=True)
    try:
        sents = [s for segs in _requests(m.V, 25, 77, 20) for s in segs]
        res = svc.translate(sents)
        assert res.n == len(sents)
        rows = [(sents[i], int(res.padded_length[i]), int(res.batch[i]), res.target(i).copy(), res.alignment(i).copy())
                for i in range(res.n)]
        res.close()
        assert _check_batches(oracle, om, rows, lambda ids, lens: osl.generate(ids, lens)) >= 3
        empty = svc.translate([])
        assert empty.n == 0
        with pytest.raises(hip.SlimtHipError, match="longer than"):
            svc.translate([np.ones(30, np.uint32)])
    finally:
        svc.close()
        gm1.close()
        gm2.close()
