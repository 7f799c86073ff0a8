"""GPU box: per-phase time of the persistent decoder (workgroup 0, one step). argv: B, S, shortlist size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n_sl = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
preset = os.environ.get("SLIMT_PRESET", "tiny11")  # base: its six decoder layers' stamps collide with the logits' from layer 4 on -- layers 0 and 1 and the step total are shown
m = synth.make_model(preset, eos_bias=-100.0)
gm = capi.Model(m); ctx = capi.Context(gm, B, S)
if os.environ.get("SLIMT_KV_FORMAT"):  # 0 = packed, 20 bits where the accumulators fit (default); 2 = packed 24-bit; 1 = f32
    gm.set_kv_cache_format(int(os.environ["SLIMT_KV_FORMAT"]))
ctx.set_decode_mode(int(os.environ.get("SLIMT_DECODE_MODE", "0")))  # 3 = 32 sentences per workgroup
ids, lens = synth.make_batch(m.V, B, S); sl = synth.make_shortlist(m.V, n_sl) if n_sl else None  # 0 = the full vocabulary
ctx.translate(ids, lens, sl)
names = ["step_start"]
for l in range(2):
    names += [f"L{l}.ssru_quant", f"L{l}.ssru_gemm", f"L{l}.ln_h", f"L{l}.q_gemm", f"L{l}.attention",
              f"L{l}.o_gemm", f"L{l}.ln_o", f"L{l}.ffn1", f"L{l}.ffn2", f"L{l}.ln_x"]
idx = list(range(21)) + [43, 44, 45, 41, 42]
names += ["  logits: barrier", "  logits: stream", "  logits: wave arg-max", "  logits: barrier", "sample/record/embed"]
for step in (5, 20):
    ctx.debug_decode_stamps(step)
    ctx.translate(ids, lens, sl)
    st = ctx.debug_decode_stamps(-1).astype(np.int64)
    t = st[idx]
    mhz = (st[61] - st[60]) / max(1, (t[-1] - t[0])) * 100.0
    print(f"--- B={B} step {step}: total {(t[-1]-t[0])/100:.1f} us   (shader clock ~{mhz:.0f} MHz)")
    for i in range(1, len(idx)):
        print(f"  {names[i]:22s} {(t[i]-t[i-1])/100:7.2f} us")
    if os.environ.get("SLIMT_DECODE_MODE") == "6":  # cluster logits: 43 rows ready | 56 published | 57 all arrived | 58 gathered | 44 my columns done | 45 | 59 candidates out | 41 all arrived
        c = st[[43, 56, 57, 58, 44, 45, 59, 41]]
        for nm, a0, a1 in zip(["publish rows", "wait for the cluster", "gather rows", "my column tiles", "lane-group reduce", "wave reduce + candidates out", "wait for the cluster"], c[:-1], c[1:]):
            print(f"    cluster: {nm:28s} {(a1-a0)/100:7.2f} us")

# (encoder phases: tools/encode_wide_phases.py [batch] [tiny11 | base])
