import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from slimt_amd import capi, synth
B, S, n_sl = 256, 32, 4096
m = synth.make_model("tiny11", eos_bias=-100.0)
gm = capi.Model(m); ctx = capi.Context(gm, B, S)
ids, lens = synth.make_batch(m.V, B, S); sl = synth.make_shortlist(m.V, n_sl)
ctx.translate(ids, lens, sl)
for step in (5, 20):
    ctx.debug_decode_stamps(step)
    ctx.translate(ids, lens, sl)
    st = ctx.debug_decode_stamps(-1).astype(np.int64)
    print("step", step, "first logits pass", (st[45]-st[20])/100, "us; second", (st[41]-st[45])/100, "us")
