"""Cluster logits (decode mode 6; the default for output layers of more than 16k columns -- BASELINE config 4, the full
32k vocabulary): four 16-sentence decoder workgroups share the output layer, each computing a quarter of its columns for
all 64 sentences and handing its best (logit, column) per sentence to the sentence's owner, which takes the first maximum
over the four candidates -- the reference's scan from class 0 with strict > (/root/reference/slimt/Transformer.cc:287-298),
whatever the split. Tokens, lengths and alignment rows must stay the checker's (oracle/, PORTABLE order), for whole and
short clusters (1..3 members: batches whose tile count is not a multiple of four), for sentences that end at staggered
steps (the members of a cluster leave the loop together), for column counts that do not divide by the members, and next
to the other tilings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(hip, oracle, synth_models):
    m = synth_models("tiny11", 6.0)
    gm, om = hip.Model(m), oracle.OracleModel(m)
    yield m, gm, om
    gm.close()


@pytest.mark.parametrize("B,S,n_sl", [(3, 9, None), (20, 32, None), (49, 17, None), (70, 32, None), (130, 7, None),
                                       (64, 32, 20000), (33, 21, 16392), (81, 32, 640), (257, 8, 4096), (16, 1, None)])
def test_cluster_logits_match_the_checker(hip, oracle, tiny, B, S, n_sl):
    from slimt_amd import synth
    m, gm, om = tiny
    ids, lens = synth.make_batch(m.V, B, S, seed=6600 + B + S, ragged=True)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl)
    oracle.set_mode(oracle.PORTABLE)
    want = om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    try:
        for mode in (6, 2, 0):
            ctx.set_decode_mode(mode)
            for policy in (1, 2):
                gm.set_kv_cache_policy(policy)
                got = ctx.translate(ids, lens, sl, want_align=True)
                assert all(np.array_equal(a, b) for a, b in zip(got, want)), (mode, policy)
        # twice in a row on one context: the arrival counters start from zero again
        ctx.set_decode_mode(6)
        got = ctx.translate(ids, lens, sl, want_align=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    finally:
        gm.set_kv_cache_policy(0)
        ctx.close()


def test_cluster_logits_without_the_admission_fall_back(hip, oracle, tiny):
    """The members of a cluster wait for each other, which is only safe while the decoder admission bounds the
    workgroups in flight: with the admission off (budget 0) mode 6 runs the plain 16-sentence tiling."""
    from slimt_amd import synth
    m, gm, om = tiny
    B, S = 40, 12
    ids, lens = synth.make_batch(m.V, B, S, seed=12, ragged=True)
    oracle.set_mode(oracle.PORTABLE)
    want = om.translate(ids, lens, None, 1.5, 0, want_align=True)[:3]
    oracle.set_mode(oracle.FAITHFUL)
    ctx = hip.Context(gm, B, S)
    try:
        gm.set_decoder_budget(0)
        ctx.set_decode_mode(6)
        got = ctx.translate(ids, lens, None, want_align=True)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    finally:
        gm.set_decoder_budget(224)  # (the default: 7/8 of the device's 256 CUs)
        ctx.close()


def test_concurrent_contexts_with_cluster_logits(hip, oracle, tiny):
    """Six contexts translating full-vocabulary batches at once under a small decoder budget: clusters of different
    launches wait inside the same admission, nothing deadlocks and every result is the checker's."""
    import threading
    from slimt_amd import synth
    m, gm, om = tiny
    shapes = [(70, 32), (33, 9), (130, 16), (20, 32), (64, 21), (49, 5)]
    jobs = []
    for i, (B, S) in enumerate(shapes):
        ids, lens = synth.make_batch(m.V, B, S, seed=9900 + i, ragged=True)
        oracle.set_mode(oracle.PORTABLE)
        want = om.translate(ids, lens, None, 1.5, 0, want_align=True)[:3]
        jobs.append((ids, lens, want))
    oracle.set_mode(oracle.FAITHFUL)
    gm.set_decoder_budget(24)
    ctxs = [hip.Context(gm, B, S) for B, S in shapes]
    errors = []

    def run(i):
        try:
            ids, lens, want = jobs[i]
            ctxs[i].set_decode_mode(6)
            for _ in range(4):
                got = ctxs[i].translate(ids, lens, None, want_align=True)
                if not all(np.array_equal(a, b) for a, b in zip(got, want)):
                    errors.append(i)
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    try:
        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(shapes))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in threads)
        assert not errors, errors
    finally:
        gm.set_decoder_budget(224)
        for c in ctxs:
            c.close()
