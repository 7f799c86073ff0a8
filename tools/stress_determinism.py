"""GPU box diagnostic: repeat concurrent translates and characterise any
mismatch against the serial result (which rows, from which decode step)."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import capi, synth

preset = sys.argv[1] if len(sys.argv) > 1 else "tiny11"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W = int(sys.argv[4]) if len(sys.argv) > 4 else 6
m = synth.make_model(preset, eos_bias=6.0)
gm = capi.Model(m)
B = int(sys.argv[5]) if len(sys.argv) > 5 else 48
S = int(sys.argv[6]) if len(sys.argv) > 6 else 24
sl = synth.make_shortlist(m.V, 2048)
jobs = [synth.make_batch(m.V, B, S, seed=1000 + i, ragged=True) for i in range(W)]
ctxs = [capi.Context(gm, B, S) for _ in range(W)]
for c in ctxs:
    c.set_decode_mode(mode)
serial = [ctxs[0].translate(ids, lens, sl, want_align=True) for ids, lens in jobs]
# also check the encoder output alone for determinism
enc_ref = [ctxs[0].encode(ids, lens)[0] for ids, lens in jobs]
bad = []
lock = threading.Lock()

def work(i):
    for it in range(iters):
        out, ln, al = ctxs[i].translate(jobs[i][0], jobs[i][1], sl, want_align=True)
        if not (np.array_equal(out, serial[i][0]) and np.array_equal(ln, serial[i][1])
                and np.array_equal(al, serial[i][2])):
            rows = np.nonzero((out != serial[i][0]).any(axis=1) | (ln != serial[i][1]))[0]
            first = [int(np.argmax(out[r] != serial[i][0][r])) for r in rows]
            with lock:
                bad.append((i, it, rows.tolist(), first))
        enc = ctxs[i].encode(jobs[i][0], jobs[i][1])[0]
        if not np.array_equal(enc, enc_ref[i]):
            d = np.nonzero((enc != enc_ref[i]).any(axis=(1, 2)))[0]
            with lock:
                bad.append((i, it, "ENC rows", d.tolist()))

ts = [threading.Thread(target=work, args=(i,)) for i in range(W)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"{preset} mode {mode}: {len(bad)} mismatches in {W * iters} concurrent translate+encode pairs")
for b in bad[:12]:
    print("  ctx", b[0], "iter", b[1], b[2], b[3])
