"""GPU box diagnostic: the shortlist generated INSIDE the encoder launch (encode_tall.hip: one workgroup publishes, the
others acquire) under uneven load -- W contexts translate batches of different sizes back to back, each batch with its
own lexical shortlist, half of them while a neighbour keeps every CU busy with a larger batch; every result is compared
with the CPU checker (its generator + its translation, PORTABLE order), every word of it.
usage: python tools/stress_generated.py [contexts=6] [repeats=6] [preset=tiny11]   (base: the D = 512 encoder)"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
from oracle import oracle as O
from slimt_amd import capi, synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
preset = sys.argv[3] if len(sys.argv) > 3 else "tiny11"
m = synth.make_model(preset, eos_bias=6.0)
gm = capi.Model(m)
om = O.OracleModel(m)
blob = synth.make_lexical_shortlist(m.V, m.V, 100, 1, seed=11, empty_fraction=0.4, min_count=1)
gen = capi.ShortlistGenerator(blob, m.V, m.V)
osl = O.OracleShortlist(blob, m.V, m.V)
# tiny11: the 64-row encoder (>= 32 tiles of it) and, for the small batches, the 32-row one; base: the D = 512 encoder
shapes = [(64, 32), (96, 32), (256, 32), (128, 16), (70, 31), (200, 20), (20, 32), (30, 12)]
if preset == "base":
    shapes = [(64, 32), (40, 17), (96, 32), (21, 9)]
jobs = []
O.set_mode(O.PORTABLE)
for i, (B, S) in enumerate(shapes):
    ids, lens = synth.make_batch(m.V, B, S, seed=7700 + i, ragged=True)
    sl = osl.generate(ids, lens)
    jobs.append((ids, lens, om.translate(ids, lens, sl, 1.5, 0, want_align=True)[:3], sl.size))
O.set_mode(O.FAITHFUL)
ctxs = [capi.Context(gm, 256, 32) for _ in range(W)]
bad, lock = [], threading.Lock()


def work(w):
    for rep in range(reps):
        for k in range(len(jobs)):
            ids, lens, want, _ = jobs[(k + w) % len(jobs)]
            got = ctxs[w].translate_pinned(ids, lens, None, want_align=True, generator=gen)
            if not all(np.array_equal(a, b) for a, b in zip(got, want)):
                with lock:
                    bad.append((w, rep, (k + w) % len(jobs)))


ts = [threading.Thread(target=work, args=(w,)) for w in range(W)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"{preset}: shortlist sizes {[j[3] for j in jobs]}; {len(bad)} mismatches in {W * reps * len(jobs)} generated translates on {W} concurrent contexts")
for b in bad[:10]:
    print("  ", b)
sys.exit(1 if bad else 0)
