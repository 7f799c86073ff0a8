"""GPU box: per-phase time of one decoder workgroup (tile 0 of context 0, one step) WHILE
the other workers run the bench load -- which phases do the neighbours stretch?
usage: python tools/decode_phases_loaded.py [workers=20] [batch=256] [src_len=32] [shortlist=4096]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import torch
from slimt_amd import capi, synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S = int(sys.argv[3]) if len(sys.argv) > 3 else 32
n_sl = int(sys.argv[4]) if len(sys.argv) > 4 else 4096  # 0 = the full vocabulary
dev = torch.device("cuda", 0)
m = synth.make_model("tiny11", seed=1234, eos_bias=-100.0)
gm = capi.Model(m)
if os.environ.get("SLIMT_KV_FORMAT"):  # 0 = packed, 20 bits where the accumulators fit (default); 2 = packed 24-bit; 1 = f32
    gm.set_kv_cache_format(int(os.environ["SLIMT_KV_FORMAT"]))
if os.environ.get("SLIMT_DECODER_BUDGET"):
    gm.set_decoder_budget(int(os.environ["SLIMT_DECODER_BUDGET"]))
if os.environ.get("SLIMT_XCD_AFFINITY"):
    gm.set_xcd_affinity(int(os.environ["SLIMT_XCD_AFFINITY"]))
ctxs = [capi.Context(gm, B, S) for _ in range(W)]
for c in ctxs:
    c.set_decode_mode(int(os.environ.get("SLIMT_DECODE_MODE", "0")))  # 3 = 32 sentences per workgroup
T = int(np.float32(1.5) * np.float32(S))
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
ids, lens = (to_dev(x) for x in synth.make_batch(m.V, B, S))
d_sl = to_dev(synth.make_shortlist(m.V, n_sl)) if n_sl else None
outs = [torch.zeros((B, T), dtype=torch.int32, device=dev) for _ in range(W)]
olen = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(W)]


def step(i):
    w = i % W
    ctxs[w].translate_device(ids.data_ptr(), lens.data_ptr(), B, S, d_sl.data_ptr() if n_sl else 0, n_sl, 1.5, 0,
                             outs[w].data_ptr(), olen[w].data_ptr(), 0, steps_hint=T)


names = ["step_start"]
for l in range(2):
    names += [f"L{l}.ssru_quant", f"L{l}.ssru_gemm", f"L{l}.ln_h", f"L{l}.q_gemm", f"L{l}.attention",
              f"L{l}.o_gemm", f"L{l}.ln_o", f"L{l}.ffn1", f"L{l}.ffn2", f"L{l}.ln_x"]
idx = list(range(21)) + [41, 42]
names += ["logits+argmax", "sample/record/embed"]
for i in range(4 * W):
    step(i)
torch.cuda.synchronize()
acc = np.zeros(len(idx) - 1)
n = 0
for rep in range(6):
    ctxs[0].debug_decode_stamps(10 + 5 * rep)  # arms context 0's next launch (synchronises its stream)
    for i in range(4 * W):
        step(i)
    torch.cuda.synchronize()
    st = ctxs[0].debug_decode_stamps(-1).astype(np.int64)
    t = st[idx]
    if t[-1] > t[0]:
        acc += np.diff(t) / 100.0
        n += 1
print(f"--- {W} workers, B={B}: decoder step under load, mean of {n} samples: total {acc.sum() / n:.1f} us")
for i in range(len(acc)):
    print(f"  {names[i + 1]:22s} {acc[i] / n:7.2f} us")
