import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from slimt_amd import capi, synth
rt = C.CDLL("libamdhip64.so")
def free_mem():
    f, t = C.c_size_t(), C.c_size_t()
    rt.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value
m = synth.make_model("tiny11", eos_bias=6.0)
ids, lens = synth.make_batch(m.V, 32, 20, ragged=True); sl = synth.make_shortlist(m.V, 1024)
blob = synth.make_lexical_shortlist(m.V, m.V, 100, 10, seed=1)
capi.device_count()
base = None
for it in range(40):
    gm = capi.Model(m); ctx = capi.Context(gm, 32, 40); gen = capi.ShortlistGenerator(blob, m.V, m.V)
    ctx.translate(ids, lens, sl); gen.generate(ids, lens)
    ids2, lens2 = synth.make_batch(m.V, 8, 40, seed=it, ragged=True); ctx.translate(ids2, lens2, None)
    gen.close(); ctx.close(); gm.close()
    fm = free_mem()
    if it == 4: base = fm
print("free after warm-up iterations:", base, "at the end:", fm, "delta MB:", (base - fm) / 1e6)
