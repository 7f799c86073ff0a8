"""GPU box: outputs of the forced tilings against the 16-sentence tiling, per sentence (debugging aid: python tools/tilings_check.py [B] [S] [ragged|full])."""
import sys, numpy as np
sys.path.insert(0, ".")
from slimt_amd import capi as hip, synth
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 32
ragged = (sys.argv[3] == "ragged") if len(sys.argv) > 3 else True
m = synth.make_model("tiny11", seed=3, eos_bias=8.0)
gm = hip.Model(m)
ids, lens = synth.make_batch(m.V, B, S, seed=5, ragged=ragged)
sl = synth.make_shortlist(m.V, 4096)
ctx = hip.Context(gm, B, S)
res = {}
for mode in (2, 4, 5):
    ctx.set_decode_mode(mode)
    out, ln, al = ctx.translate(ids, lens, sl, want_align=True)
    res[mode] = (out.copy(), ln.copy(), al.copy())
for mode in (4, 5):
    o, l, a = res[mode]; o2, l2, a2 = res[2]
    bad = [b for b in range(B) if not np.array_equal(o[b], o2[b])]
    print(f"mode {mode}: {len(bad)} of {B} sentences differ; lens of the first: {[int(lens[b]) for b in bad[:10]]}; rows {bad[:10]}")
    for b in bad[:4]:
        t = int(np.argmax(o[b] != o2[b])); print(f"   sentence {b} len {lens[b]} first differs at step {t}")
    print(f"   alignment equal: {np.array_equal(a, a2)}; max diff {np.abs(a - a2).max()}")
