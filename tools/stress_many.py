"""GPU box diagnostic: MERGED launches (slimt_hip_translate_many_async[_generated]) under uneven load -- W contexts each
merge k = 1..8 batches of different sizes and padded lengths per call, back to back; a third of the calls share ONE fixed
shortlist (dense sub-batches: one output layer, sentences of several batches in a tile), a third run the full vocabulary,
a third generate every batch's own lexical shortlist inside the one encoder launch (aligned sub-batches); every word of
every batch -- tokens, lengths, alignment rows -- is compared with the CPU checker's result for that batch alone.
usage: python tools/stress_many.py [contexts=6] [calls per context=9] [preset=tiny11]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
from oracle import oracle as O
from slimt_amd import capi, synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 6
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 9
preset = sys.argv[3] if len(sys.argv) > 3 else "tiny11"
m = synth.make_model(preset, eos_bias=6.0)
gm = capi.Model(m)
om = O.OracleModel(m)
blob = synth.make_lexical_shortlist(m.V, m.V, 100, 1, seed=11, empty_fraction=0.4, min_count=1)
gen = capi.ShortlistGenerator(blob, m.V, m.V)
osl = O.OracleShortlist(blob, m.V, m.V)
fixed = synth.make_shortlist(m.V, 1024)
shapes = [(64, 32), (31, 32), (40, 27), (17, 29), (64, 30), (9, 32), (50, 26), (33, 31), (5, 28), (64, 32)]
if preset == "base":
    shapes = [(19, 32), (12, 27), (33, 30), (7, 32), (24, 29)]
# the checker's result of every batch under the three vocabularies (PORTABLE order)
O.set_mode(O.PORTABLE)
batches, want = [], []
for i, (B, S) in enumerate(shapes):
    ids, lens = synth.make_batch(m.V, B, S, seed=9100 + i, ragged=True)
    own = osl.generate(ids, lens)
    batches.append((ids, lens))
    want.append([om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3] for sl in (fixed, None, own)])
O.set_mode(O.FAITHFUL)
rows = capi.translate_many_rows([64] * 8)
ctxs = [capi.Context(gm, rows, 32) for _ in range(W)]
bad, lock, done = [], threading.Lock(), [0]


def work(w):
    r = np.random.Generator(np.random.PCG64(77 + w))
    for c in range(calls):
        k = int(r.integers(1, 9)) if preset != "base" else int(r.integers(1, 5))
        pick = [int(x) for x in r.integers(0, len(batches), size=k)]
        mode = (c + w) % 3  # 0: one fixed shortlist, 1: full vocabulary, 2: generated per batch
        pins, bufs = [], []
        for j in pick:
            ids, lens = batches[j]
            B, S = ids.shape
            T = max(int(np.float32(1.5) * np.float32(S)), 1)
            ps = [capi._Pinned() for _ in range(5)]
            b = (ps[0].array(np.uint32, (B, S)), ps[1].array(np.uint32, (B,)), ps[2].array(np.uint32, (B, T)),
                 ps[3].array(np.uint32, (B,)), ps[4].array(np.float32, (B, T, S)))
            b[0][...] = ids
            b[1][...] = lens
            b[2][...] = 0x5a5a5a5a
            b[3][...] = 0x5a5a5a5a
            b[4][...] = np.float32(7.25)
            pins.append(ps)
            bufs.append(b)
        ctxs[w].translate_many_async(bufs, fixed if mode == 0 else None, generator=gen if mode == 2 else None)
        ctxs[w].synchronize()
        for j, b in zip(pick, bufs):
            if not all(np.array_equal(a, e) for a, e in zip((b[2], b[3], b[4]), want[j][mode])):
                with lock:
                    bad.append((w, c, mode, k, j))
        with lock:
            done[0] += k
        for ps in pins:
            for p in ps:
                p.free()


ts = [threading.Thread(target=work, args=(w,)) for w in range(W)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"{preset}: {len(bad)} mismatching batches in {done[0]} batches of {W * calls} merged calls on {W} concurrent contexts "
      f"(fixed shortlist / full vocabulary / generated per batch in turn)")
for b in bad[:10]:
    print("   (context, call, mode, batches in the call, batch)", b)
sys.exit(1 if bad else 0)
