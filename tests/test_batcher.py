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
