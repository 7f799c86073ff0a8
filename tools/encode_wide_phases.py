"""GPU box: per-phase time of the D = 512 persistent encoder (workgroup 0, one layer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, n_sl = 32, 4096
m = synth.make_model("base", eos_bias=-100.0)
gm = capi.Model(m); ctx = capi.Context(gm, B, S)
ids, lens = synth.make_batch(m.V, B, S); sl = synth.make_shortlist(m.V, n_sl)
ctx.translate(ids, lens, sl)
names = ["qkv round 0", "attention round 0", "qkv round 1", "attention round 1", "o_gemm", "ln+quant", "ffn1",
         "ffn2 loop", "ffn2 exchange", "ln"]
for layer in (0, 2):
    ctx.debug_decode_stamps(layer)
    ctx.translate(ids, lens, sl)
    st = ctx.debug_decode_stamps(-1).astype(np.int64)[48:59]
    print(f"--- wide encoder layer {layer} (workgroup 0): total {(st[10]-st[0])/100:.1f} us")
    for i in range(10):
        print(f"  {names[i]:22s} {(st[i+1]-st[i])/100:7.2f} us")
