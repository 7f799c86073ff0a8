"""GPU box: per-phase time of the persistent encoders that stage heads in rounds (workgroup 0, one
layer): the D = 512 kernel (preset base, 32 rows) or the 64-row D = 256 kernel (preset tiny11).
usage: encode_wide_phases.py [batch] [preset]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, n_sl = 32, 4096
preset = sys.argv[2] if len(sys.argv) > 2 else "base"
m = synth.make_model(preset, eos_bias=-100.0)
gm = capi.Model(m); ctx = capi.Context(gm, B, S)
if preset != "base":
    ctx.set_encode_rows(64)
ids, lens = synth.make_batch(m.V, B, S); sl = synth.make_shortlist(m.V, n_sl)
ctx.translate(ids, lens, sl)
names = ["qkv round 0", "attention round 0", "qkv round 1", "attention round 1", "o_gemm", "ln+quant", "ffn1",
         "ffn2 loop", "ffn2 exchange", "ln"]
for layer in (0, 2):
    ctx.debug_decode_stamps(layer)
    ctx.translate(ids, lens, sl)
    st = ctx.debug_decode_stamps(-1).astype(np.int64)[48:59]
    print(f"--- {preset} encoder layer {layer} (workgroup 0): total {(st[10]-st[0])/100:.1f} us")
    for i in range(10):
        print(f"  {names[i]:22s} {(st[i+1]-st[i])/100:7.2f} us")
