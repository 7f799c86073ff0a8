"""GPU box, diagnosis build only (-DSLIMT_EXP_QKV_STAMPS=<round>): every wave's clock at three points of one Q/K/V projection
round of the 64-row encoder (workgroup 0, layer 2): when it arrives at the round (round 1: in front of its barrier), when it has
quantised its rows (round 0; round 1: behind the barrier), when the barrier in front of the MFMAs lets it go.
Relative to the layer's start (wave 0's phase stamp 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import capi, synth
B, S = 256, 32
m = synth.make_model("tiny11", eos_bias=-100.0)
gm = capi.Model(m); ctx = capi.Context(gm, B, S)
ctx.set_encode_rows(64)
ids, lens = synth.make_batch(m.V, B, S)
ctx.encode(ids, lens)
for rep in range(3):
    ctx.debug_decode_stamps(2)
    ctx.encode(ids, lens)
    st = ctx.debug_decode_stamps(-1).astype(np.int64)
    t0 = st[48]
    print(f"--- layer 2, repeat {rep}: phase stamps (us from layer start): " + " ".join(f"{(x - t0) / 100:.2f}" for x in st[48:59]))
    for w in range(16):
        a, b, c = [(st[3 * w + i] - t0) / 100 for i in range(3)]
        print(f"  wave {w:2d} ({'Q+V' if w < 8 else 'K  '}): arrives {a:6.2f}  quantised / through the first barrier {b:6.2f} (+{b - a:.2f})  starts its MFMAs {c:6.2f} (+{c - b:.2f})")
