"""GPU box: does XCD-affine placement put a batch's decoder workgroups where it says?
One worker (launches do not overlap), occupancy trace on: the 16 begin events of each
decoder launch and the XCDs they ran on.
usage: python tools/xcd_check.py [affinity=1] [launches=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import torch
from slimt_amd import capi, synth

aff = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, S, n_sl = 256, 32, 4096
dev = torch.device("cuda", 0)
m = synth.make_model("tiny11", seed=1234, eos_bias=-100.0)
gm = capi.Model(m)
gm.set_xcd_affinity(aff)
ctx = capi.Context(gm, B, S)
T = int(np.float32(1.5) * np.float32(S))
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
ids, lens = (to_dev(x) for x in synth.make_batch(m.V, B, S))
d_sl = to_dev(synth.make_shortlist(m.V, n_sl))
out = torch.zeros((B, T), dtype=torch.int32, device=dev)
olen = torch.zeros((B,), dtype=torch.int32, device=dev)
cap = 4096
buf = torch.zeros(1 + 3 * cap, dtype=torch.int64, device=dev)
capi._chk(capi.lib().slimt_hip_debug_occupancy_trace(buf.data_ptr(), cap))
for i in range(n):
    ctx.translate_device(ids.data_ptr(), lens.data_ptr(), B, S, d_sl.data_ptr(), n_sl, 1.5, 0,
                         out.data_ptr(), olen.data_ptr(), 0, steps_hint=T)
    torch.cuda.synchronize()
capi._chk(capi.lib().slimt_hip_debug_occupancy_trace(0, 0))
h = buf.cpu().numpy()
cnt = int(h[0])
ev = h[1:1 + 3 * min(cnt, cap)].reshape(-1, 3)
dec = [(int(e[2]), int(e[1]) >> 32, (int(e[0]) >> 16)) for e in ev if (int(e[0]) & 0xff) == 1 and ((int(e[0]) >> 8) & 0xff) == 0]
dec.sort()
print(f"affinity {aff}: {len(dec)} decoder workgroup starts in {n} launches")
for k in range(0, len(dec), 16):
    grp = dec[k:k + 16]
    xs = sorted(x for _, x, _ in grp)
    print(f"  launch {k // 16}: XCDs {xs}  blockIdx%8 {sorted(b % 8 for _, _, b in grp)}")
assert int(olen.sum().item()) == B * T
