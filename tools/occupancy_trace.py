"""Where do the CUs spend their time under the bench load?

Runs the flagship workload (tiny11, B=256, S=32, shortlist 4096, W workers) with the
diagnostic occupancy trace on (slimt_hip_debug_occupancy_trace): thread 0 of every
workgroup of the persistent encoder / decoder logs a begin and an end event with its
physical CU (HW_ID, XCC_ID) and the 100 MHz wall clock. Both kernels take a whole CU
(LDS), so per CU the events alternate begin / end. Prints, over the middle of the run:
busy fraction of the CUs (all, per kernel, per XCD), workgroup durations, idle gaps.

usage: python tools/occupancy_trace.py [workers=16] [steps=192] [batch=256]
"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    S, n_sl = 32, 4096
    import torch
    from slimt_amd import capi, synth

    dev = torch.device("cuda", 0)
    model = synth.make_model("tiny11", seed=1234, eos_bias=-100.0)
    sl = synth.make_shortlist(model.V, n_sl)
    gm = capi.Model(model, device=0)
    ctxs = [capi.Context(gm, B, S) for _ in range(W)]
    for c in ctxs:
        c.set_decode_mode(int(os.environ.get("SLIMT_DECODE_MODE", "0")))  # 3 = 32 sentences per decoder workgroup
    T = int(np.float32(1.5) * np.float32(S))

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)

    batches = [tuple(to_dev(x) for x in synth.make_batch(model.V, B, S, seed=4321 + i)) for i in range(4)]
    d_sl = to_dev(sl)
    outs = [torch.zeros((B, T), dtype=torch.int32, device=dev) for _ in range(W)]
    lens = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(W)]

    def step(i):
        w = i % W
        ids, ln = batches[i % 4]
        ctxs[w].translate_device(ids.data_ptr(), ln.data_ptr(), B, S, d_sl.data_ptr(), n_sl, 1.5, 0,
                                 outs[w].data_ptr(), lens[w].data_ptr(), 0, steps_hint=T)

    for i in range(W):
        step(i)
    torch.cuda.synchronize()
    wgs_per_step = -(-B // 4) + -(-B * S // 32)
    cap = 2 * wgs_per_step * steps + 1024
    buf = torch.zeros(1 + 3 * cap, dtype=torch.int64, device=dev)
    capi._chk(capi.lib().slimt_hip_debug_occupancy_trace(buf.data_ptr(), cap))
    import time
    mode = os.environ.get("OCC_HOST", "single")  # single: one enqueue loop; threads: a host thread per worker
    t_host0 = time.perf_counter()
    if mode == "threads":
        import threading
        depth = int(os.environ.get("OCC_DEPTH", "1"))  # batches a worker keeps queued before it waits

        def worker(w):
            n = 0
            for i in range(w, steps, W):
                step(i)
                n += 1
                if n % depth == 0:
                    ctxs[w].synchronize()
            ctxs[w].synchronize()

        ts = [threading.Thread(target=worker, args=(w,)) for w in range(W)]
        [x.start() for x in ts]
        [x.join() for x in ts]
        t_enq = time.perf_counter() - t_host0
    else:
        for i in range(steps):
            step(i)
        t_enq = time.perf_counter() - t_host0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_host0
    toks = int(lens[0].sum().item()) * steps
    print(f"host mode {mode}: enqueue loop {1e3 * t_enq:.1f} ms, all done {1e3 * t_all:.1f} ms -> "
          f"{toks / t_all / 1e6:.2f} M tok/s")
    capi._chk(capi.lib().slimt_hip_debug_occupancy_trace(None, 0))
    h = buf.cpu().numpy().view(np.uint64)
    n = int(min(h[0], cap))
    ev = h[1:1 + 3 * n].reshape(n, 3)
    kernel = (ev[:, 0] & 0xFF).astype(np.int64)
    is_end = ((ev[:, 0] >> 8) & 1).astype(np.int64)
    hw = ev[:, 1] & 0xFFFFFFFF
    xcc = ((ev[:, 1] >> 32) & 0xF).astype(np.int64)
    cu = ((hw >> 8) & 0xF).astype(np.int64)
    sh = ((hw >> 12) & 0x1).astype(np.int64)
    se = ((hw >> 13) & 0x7).astype(np.int64)
    t = ev[:, 2].astype(np.int64)
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    t0, t1 = t.min(), t.max()
    lo, hi = t0 + (t1 - t0) // 5, t1 - (t1 - t0) // 5  # middle 60 %
    print(f"events {n} (capacity {cap}); distinct CUs {len(np.unique(key))}; XCDs {sorted(np.unique(xcc).tolist())}; "
          f"span {1e-5 * (t1 - t0):.2f} ms; window {1e-5 * (hi - lo):.2f} ms")
    busy = {1: 0, 2: 0}
    busy_xcd = {}
    durs = {1: [], 2: []}
    gaps = []
    gap_kind = {}   # (previous kernel, next kernel) -> [count, total ticks] of gaps > 20 us
    intervals = []  # (begin, end, kernel) of every workgroup
    unmatched = 0
    for k in np.unique(key):
        m = key == k
        order = np.argsort(t[m], kind="stable")
        tk, ek, kk = t[m][order], is_end[m][order], kernel[m][order]
        x = int(xcc[m][0])
        last_end = None
        last_kernel = 0
        i = 0
        while i + 1 < len(tk):
            if ek[i] == 0 and ek[i + 1] == 1 and kk[i] == kk[i + 1]:
                b, e = tk[i], tk[i + 1]
                durs[int(kk[i])].append(e - b)
                ov = max(0, min(e, hi) - max(b, lo))
                busy[int(kk[i])] += ov
                busy_xcd[x] = busy_xcd.get(x, 0) + ov
                if last_end is not None and b >= lo and b <= hi:
                    gaps.append(b - last_end)
                    if b - last_end > 2000:
                        gk = gap_kind.setdefault((last_kernel, int(kk[i])), [0, 0])
                        gk[0] += 1
                        gk[1] += b - last_end
                last_end = e
                last_kernel = int(kk[i])
                intervals.append((b, e, int(kk[i])))
                i += 2
            else:
                unmatched += 1
                i += 1
    ncu = len(np.unique(key))
    total = (hi - lo) * ncu
    print(f"busy fraction of {ncu} CUs in the window: {100.0 * (busy[1] + busy[2]) / total:.1f} %  "
          f"(decoder {100.0 * busy[1] / total:.1f} %, encoder {100.0 * busy[2] / total:.1f} %); "
          f"unmatched events {unmatched}")
    per = {x: 100.0 * v / ((hi - lo) * np.sum([1 for k in np.unique(key) if (k >> 8) == x])) for x, v in busy_xcd.items()}
    print("busy per XCD: " + "  ".join(f"{x}:{per[x]:.1f}%" for x in sorted(per)))
    for k, name in ((1, "decoder"), (2, "encoder")):
        d = 1e-2 * np.array(durs[k], dtype=np.float64)  # us
        if len(d):
            print(f"{name} workgroups {len(d)}: mean {d.mean():.0f} us, p10 {np.percentile(d, 10):.0f}, "
                  f"p50 {np.percentile(d, 50):.0f}, p90 {np.percentile(d, 90):.0f}")
    g = 1e-2 * np.array(gaps, dtype=np.float64)
    if len(g):
        print(f"idle gaps between consecutive workgroups on a CU: {len(g)}; mean {g.mean():.1f} us, p50 "
              f"{np.percentile(g, 50):.1f}, p90 {np.percentile(g, 90):.1f}, max {g.max():.0f}; "
              f"sum {1e-3 * g.sum():.1f} ms = {100.0 * 1e2 * g.sum() / total:.1f} % of the window's CU time")
    names = {1: "dec", 2: "enc"}
    print("gaps > 20 us by (previous -> next workgroup on the CU): " + "; ".join(
        f"{names[a]}->{names[b]}: {c} gaps, {1e-5 * tot:.0f} ms" for (a, b), (c, tot) in sorted(gap_kind.items())))
    # busy CUs over time (50 us bins) inside the window
    bins = np.arange(lo, hi, 5000)
    iv = np.array(intervals, dtype=np.int64)
    occ = np.zeros(len(bins), dtype=np.float64)
    occ_dec = np.zeros(len(bins), dtype=np.float64)
    for b, e, k in iv:
        i0 = max(0, (b - lo) // 5000)
        i1 = min(len(bins) - 1, (e - lo) // 5000)
        if e < lo or b > hi:
            continue
        occ[i0:i1 + 1] += 1
        if k == 1:
            occ_dec[i0:i1 + 1] += 1
    print(f"CUs with a workgroup per 50 us bin: mean {occ.mean():.0f}, p5 {np.percentile(occ, 5):.0f}, p25 "
          f"{np.percentile(occ, 25):.0f}, p50 {np.percentile(occ, 50):.0f}, p95 {np.percentile(occ, 95):.0f}; "
          f"decoder CUs mean {occ_dec.mean():.0f}, p5 {np.percentile(occ_dec, 5):.0f}, p95 {np.percentile(occ_dec, 95):.0f}")
    step = max(1, len(bins) // 60)
    print("timeline (busy CUs / decoder CUs every %d us): " % (50 * step) +
          " ".join(f"{int(occ[i])}/{int(occ_dec[i])}" for i in range(0, len(bins), step)))
    # encoder launches: workgroups that begin within 1 ms of each other on >= 200 CUs... (dispatch spread)
    enc_b = np.sort(iv[iv[:, 2] == 2][:, 0])
    if len(enc_b) > 1:
        d = np.diff(enc_b)
        print(f"encoder workgroup starts: {len(enc_b)}; mean spacing {1e-2 * d.mean():.2f} us "
              f"(= {1e5 / max(1.0, d.mean()):.0f} starts/ms)")
    for c in ctxs:
        c.close()
    gm.close()


if __name__ == "__main__":
    main()
