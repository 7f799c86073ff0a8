"""Diagnostic (GPU box): print where v_mfma_i32_16x16x64_i8 puts things, as seen
through slimt_hip_affine_acc_i32 on a single 16x64x16 tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import capi

M, K, N = 16, 64, 16
W = np.zeros((N, K), dtype=np.int8)
# W[n][k] encodes (n, k): value = (n*64+k) % 127 + 1 in 1..127 -> decode by brute force
code = ((np.arange(N)[:, None] * 64 + np.arange(K)[None, :]) % 127 + 1).astype(np.int8)
colsum = code.astype(np.int64).sum(axis=1)
bad = 0
for (i0, k0) in [(0, 0), (1, 0), (0, 1), (0, 16), (0, 17), (5, 33), (15, 63), (4, 20)]:
    x = np.zeros((M, K), dtype=np.float32)
    x[i0, k0] = 1.0
    acc = capi.affine_acc_i32(x, code, 1.0).astype(np.int64) - 127 * colsum[None, :]
    rows = np.nonzero(np.any(acc != 0, axis=1))[0]
    exp_row = code[:, k0].astype(np.int64)
    ok = rows.tolist() == [i0] and np.array_equal(acc[i0], exp_row)
    bad += (not ok)
    print(f"one-hot A[{i0},{k0}] -> nonzero rows {rows.tolist()} ok={ok}")
    if not ok and len(rows):
        r = rows[0]
        # which k did each column pick up?
        ks = []
        for n in range(N):
            cand = np.nonzero(code[n].astype(np.int64) == acc[r, n])[0]
            ks.append(cand.tolist()[:2])
        print("   row", r, "per-column matching k:", ks)
print("MFMA layout", "OK" if bad == 0 else f"MISMATCH in {bad} probes")
