"""Lexical shortlist (slimt/Shortlist.{hh,cc}).

CPU: the oracle's restatement of ShortlistGenerator::generate / load against an
independent numpy set construction, the blob checks of load(). GPU: the device
kernel behind slimt_hip_shortlist_generate against the oracle, id for id."""
import struct

import numpy as np
import pytest

from slimt_amd import synth


def numpy_generate(blob, V_tgt, ids, lengths, shared):
    """Set-based re-derivation of Shortlist.cc:115-175 straight from the blob."""
    frequent, best, n_off, n_ids = struct.unpack_from("<4Q", blob, 16)
    off = np.frombuffer(blob, dtype=np.uint64, count=n_off, offset=48)
    lists = np.frombuffer(blob, dtype=np.uint32, count=n_ids, offset=48 + 8 * n_off)
    chosen = set(range(min(frequent, V_tgt)))
    for b in range(ids.shape[0]):
        for w in ids[b, : int(lengths[b])]:
            if shared:
                chosen.add(int(w))
            chosen.update(int(t) for t in lists[int(off[w]): int(off[w + 1])])
    i = frequent
    while len(chosen) % 8 and i < V_tgt:
        if i not in chosen:
            chosen.add(i)
        i += 1
    return np.array(sorted(chosen), dtype=np.uint32)


CASES = [  # V_src, V_tgt, frequent, best, B, S, shared, seed
    (512, 512, 100, 20, 4, 9, False, 1),
    (512, 500, 100, 20, 7, 16, True, 2),     # target vocabulary not a multiple of 32
    (2048, 2048, 8, 100, 33, 12, False, 3),
    (300, 2048, 100, 3, 2, 5, False, 4),     # few aligned words: the x8 patch adds ids
    (2048, 300, 100, 50, 16, 32, False, 5),  # nearly everything selected
    (32000, 32000, 100, 100, 64, 32, False, 6),
    (4096, 4096, 100, 100, 256, 32, False, 7),  # nearly full table: the x8 patch has to search far
    (640, 640, 64, 2, 3, 4, False, 8),          # frequent on a bitmap word boundary
    (640, 640, 700, 2, 3, 4, False, 9),         # frequent > vocabulary: no room for the patch
]


@pytest.mark.parametrize("Vs,Vt,frequent,best,B,S,shared,seed", CASES)
def test_oracle_generate_matches_set_construction(oracle, Vs, Vt, frequent, best, B, S, shared, seed):
    blob = synth.make_lexical_shortlist(Vs, Vt, frequent, best, seed=seed)
    ids, lens = synth.make_batch(min(Vs, Vt) if shared else Vs, B, S, seed=seed, ragged=True)
    sl = oracle.OracleShortlist(blob, Vs, Vt, shared=shared, check=True)
    got = sl.generate(ids, lens)
    want = numpy_generate(blob, Vt, ids, lens, shared)
    assert np.array_equal(got, want)
    assert np.all(np.diff(got.astype(np.int64)) > 0)
    assert got.size % 8 == 0 or got.size == Vt  # Shortlist.cc:158-165
    assert np.array_equal(got[: min(frequent, Vt)], np.arange(min(frequent, Vt)))


def test_oracle_load_checks(oracle):
    blob = bytearray(synth.make_lexical_shortlist(64, 64, 8, 4, seed=9))
    assert oracle.shortlist_checksum(bytes(blob)) == struct.unpack_from("<Q", blob, 8)[0]
    oracle.OracleShortlist(bytes(blob), 64, 64, check=True)
    bad = bytearray(blob); bad[0] ^= 1  # magic, Shortlist.cc:56
    with pytest.raises(ValueError):
        oracle.OracleShortlist(bytes(bad), 64, 64)
    with pytest.raises(ValueError):  # size mismatch, Shortlist.cc:58-64
        oracle.OracleShortlist(bytes(blob[:-4]), 64, 64)
    bad = bytearray(blob); bad[24] ^= 1  # header.best changed: checksum, Shortlist.cc:66-76
    with pytest.raises(ValueError):
        oracle.OracleShortlist(bytes(bad), 64, 64, check=True)
    oracle.OracleShortlist(bytes(bad), 64, 64, check=False)  # unchecked load accepts it
    with pytest.raises(ValueError):  # header too short, Shortlist.cc:49-51
        oracle.OracleShortlist(bytes(blob[:40]), 64, 64)


@pytest.mark.gpu
@pytest.mark.parametrize("Vs,Vt,frequent,best,B,S,shared,seed", CASES)
def test_gpu_generate_matches_oracle(hip, oracle, Vs, Vt, frequent, best, B, S, shared, seed):
    blob = synth.make_lexical_shortlist(Vs, Vt, frequent, best, seed=seed)
    ids, lens = synth.make_batch(min(Vs, Vt) if shared else Vs, B, S, seed=seed, ragged=True)
    want = oracle.OracleShortlist(blob, Vs, Vt, shared=shared).generate(ids, lens)
    gen = hip.ShortlistGenerator(blob, Vs, Vt, shared=shared, check=True)
    assert (gen.frequent, gen.best) == (frequent, best)
    for _ in range(2):  # the handle is reusable
        got = gen.generate(ids, lens)
        assert np.array_equal(got, want)
    gen.close()


@pytest.mark.gpu
def test_gpu_generate_is_safe_on_one_shared_handle(hip, oracle):
    """ShortlistGenerator::generate is const and every Async worker calls it on ONE shared
    generator (Model.cc:117-120, Frontend.cc:212-226): six threads with batches of very
    different sizes (the staging buffers get re-reserved) on one handle, every result must
    be that thread's own batch's shortlist."""
    import threading
    Vs = Vt = 4000
    blob = synth.make_lexical_shortlist(Vs, Vt, 50, 20, seed=3)
    osl = oracle.OracleShortlist(blob, Vs, Vt)
    jobs = []
    for i, (B, S) in enumerate([(1, 3), (64, 32), (7, 128), (256, 32), (2, 2), (33, 17)]):
        ids, lens = synth.make_batch(Vs, B, S, seed=100 + i, ragged=True)
        jobs.append((ids, lens, osl.generate(ids, lens)))
    assert len({j[2].tobytes() for j in jobs}) == len(jobs)  # all different
    gen = hip.ShortlistGenerator(blob, Vs, Vt)
    bad = []

    def work(i):
        ids, lens, want = jobs[i]
        for rep in range(20):
            got = gen.generate(ids, lens)
            if not np.array_equal(got, want):
                bad.append((i, rep))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    gen.close()
    assert not bad, bad


@pytest.mark.gpu
def test_gpu_rejects_bad_blobs(hip):
    blob = bytearray(synth.make_lexical_shortlist(64, 64, 8, 4, seed=9))
    bad = bytearray(blob); bad[0] ^= 1
    with pytest.raises(hip.SlimtHipError, match="magic"):
        hip.ShortlistGenerator(bytes(bad), 64, 64)
    with pytest.raises(hip.SlimtHipError, match="file size"):
        hip.ShortlistGenerator(bytes(blob[:-4]), 64, 64)
    bad = bytearray(blob); bad[24] ^= 1  # header.best
    with pytest.raises(hip.SlimtHipError, match="checksum"):
        hip.ShortlistGenerator(bytes(bad), 64, 64, check=True)
    with pytest.raises(hip.SlimtHipError, match="out of bounds"):  # id >= target vocabulary
        hip.ShortlistGenerator(bytes(blob), 64, 8)
    gen = hip.ShortlistGenerator(bytes(blob), 64, 64)
    with pytest.raises(hip.SlimtHipError, match="out of range"):
        gen.generate(np.full((1, 4), 64, np.uint32), np.array([4], np.uint32))
    gen.close()


@pytest.mark.gpu
def test_gpu_generated_shortlist_feeds_translate(hip, oracle):
    """Model::forward's order (Model.cc:117-120,195-203): generate the batch's
    shortlist, then translate with it -- both on the device, against the oracle
    doing the same on the CPU."""
    m = synth.make_model("micro", eos_bias=3.0)
    blob = synth.make_lexical_shortlist(m.V, m.V, frequent=16, best=6, seed=11)
    ids, lens = synth.make_batch(m.V, 6, 9, seed=5, ragged=True)
    gen = hip.ShortlistGenerator(blob, m.V, m.V)
    sl = gen.generate(ids, lens)
    want_sl = oracle.OracleShortlist(blob, m.V, m.V).generate(ids, lens)
    assert np.array_equal(sl, want_sl) and sl.size % 8 == 0
    gm = hip.Model(m)
    ctx = hip.Context(gm, 6, 9)
    out, ln, _ = ctx.translate(ids, lens, sl)
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, _, _ = oracle.OracleModel(m).translate(ids, lens, want_sl, 1.5, 0)
    oracle.set_mode(oracle.FAITHFUL)
    assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out)
    ctx.close(); gm.close(); gen.close()


class _Hip:
    """Minimal device-buffer helper over libamdhip64 (tests only)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        # the runtime this process has mapped already (the product library's): asking the loader for "libamdhip64.so" by
        # name would map ROCm's copy NEXT to a PyTorch-bundled one
        from slimt_amd import capi
        capi.lib()
        mapped = capi.mapped_hip_runtimes()
        self.rt = C.CDLL(mapped[0] if mapped else "libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.c_void_p, C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.bufs = []

    def to_dev(self, a):
        a = np.ascontiguousarray(a)
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(a.nbytes, 16)) == 0
        assert self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        self.bufs.append(p)
        return p.value

    def from_dev(self, ptr, shape, dtype):
        out = np.empty(shape, dtype)
        assert self.rt.hipMemcpy(out.ctypes.data, self.C.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def free(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []


@pytest.mark.gpu
@pytest.mark.parametrize("preset,B,S,mode", [("tiny11", 20, 12, 0), ("tiny11", 5, 40, 0), ("mini", 6, 9, 0),
                                              ("micro", 6, 9, 1)])
def test_gpu_translate_with_generated_shortlist_on_device(hip, oracle, synth_models, preset, B, S, mode):
    """slimt_hip_translate_device_generated: shortlist generation and translation on
    one stream, the shortlist's size never visiting the host (persistent kernels;
    mode 1: one read-back). Against the oracle doing both steps on the CPU."""
    m = synth_models(preset, 6.0 if preset == "tiny11" else 1.0)
    blob = synth.make_lexical_shortlist(m.V, m.V, frequent=24, best=4, seed=B + S)
    ids, lens = synth.make_batch(m.V, B, S, seed=S, ragged=True)
    want_sl = oracle.OracleShortlist(blob, m.V, m.V).generate(ids, lens)
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, _, _ = oracle.OracleModel(m).translate(ids, lens, want_sl, 1.5, 0)
    oracle.set_mode(oracle.FAITHFUL)
    gm = hip.Model(m)
    ctx = hip.Context(gm, B, S)
    ctx.set_decode_mode(mode)
    gen = hip.ShortlistGenerator(blob, m.V, m.V)
    dev = _Hip()
    T = max(1, int(np.float32(1.5) * np.float32(S)))
    d_ids, d_len = dev.to_dev(ids), dev.to_dev(lens)
    d_out, d_olen = dev.to_dev(np.full((B, T), 7, np.uint32)), dev.to_dev(np.zeros(B, np.uint32))
    for _ in range(2):
        ctx.translate_device_generated(gen, d_ids, d_len, B, S, 1.5, 0, d_out, d_olen)
        ctx.synchronize()
        ln = dev.from_dev(d_olen, (B,), np.uint32)
        out = dev.from_dev(d_out, (B, T), np.uint32)
        assert np.array_equal(ln, w_ln)
        for b in range(B):
            assert np.array_equal(out[b, : ln[b]], w_out[b, : w_ln[b]]), b
    dev.free(); gen.close(); ctx.close(); gm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("preset,rows", [("tiny11", 64), ("tiny11", 32), ("base", 0)])
def test_gpu_shortlist_never_published_fails_the_batch(hip, oracle, synth_models, preset, rows):
    """a workgroup that waits for the shortlist generated inside its encoder launch used to give up after its
    bounded poll and pack the output layer from ids and a count nobody had published -- translations from a stale or
    partial shortlist, silently. Now a waiter that gives up packs nothing and raises the context's error word: the call
    FAILS. Reached here with a publication the waiters cannot see (slimt_hip_debug_break_shortlist_handoff) and a short
    poll limit, in each of the three encoders; afterwards the same context translates correctly again."""
    m = synth_models(preset, 6.0)
    blob = synth.make_lexical_shortlist(m.V, m.V, frequent=64, best=8, seed=21)
    B, S = 40, 12
    ids, lens = synth.make_batch(m.V, B, S, seed=33, ragged=True)
    gen = hip.ShortlistGenerator(blob, m.V, m.V)
    want_sl = oracle.OracleShortlist(blob, m.V, m.V).generate(ids, lens)
    oracle.set_mode(oracle.PORTABLE)
    w_out, w_ln, _, _ = oracle.OracleModel(m).translate(ids, lens, want_sl, 1.5, 0)
    oracle.set_mode(oracle.FAITHFUL)
    gm = hip.Model(m)
    ctx = hip.Context(gm, B, S)
    try:
        if rows:
            ctx.set_encode_rows(rows)
        out, ln, _ = ctx.translate_generated(gen, ids, lens)
        assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out)
        ctx.debug_break_shortlist_handoff(True, 1 << 10)
        with pytest.raises(hip.SlimtHipError, match="never published"):
            ctx.translate_generated(gen, ids, lens)
        ctx.debug_break_shortlist_handoff(False)
        for _ in range(2):  # the error does not stick, and the context is usable
            out, ln, _ = ctx.translate_generated(gen, ids, lens)
            assert np.array_equal(ln, w_ln) and np.array_equal(out, w_out)
    finally:
        ctx.close(); gm.close(); gen.close()
