"""GPU box: W threads, one context each, calling the blocking slimt_hip_translate (pageable numpy
buffers in and out) in a loop -- slimt's own Async workers calling Model::forward (Frontend.cc:212-226).
usage: sync_workers_bench.py [workers] [batches per worker] [batch] [src_len] [pinned 0/1]"""
import json, os, sys, threading, time
if not os.environ.get("SLIMT_TOOL_NO_QUEUE_DEFAULT"):  # (set: rely on the default the library sets when it is loaded)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import capi, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
S = int(sys.argv[4]) if len(sys.argv) > 4 else 32
pinned = (sys.argv[5] if len(sys.argv) > 5 else "0") == "1"
m = synth.make_model("tiny11", eos_bias=-100.0)
gm = capi.Model(m)
sl = synth.make_shortlist(m.V, 4096)
ctxs = [capi.Context(gm, B, S) for _ in range(W)]
jobs = [synth.make_batch(m.V, B, S, seed=100 + w) for w in range(W)]
def run(w, n):
    f = ctxs[w].translate_pinned if pinned else ctxs[w].translate
    for _ in range(n):
        f(jobs[w][0], jobs[w][1], sl)
for phase, n in (("warm", 2), ("timed", N)):
    ts = [threading.Thread(target=run, args=(w, n)) for w in range(W)]
    t0 = time.time()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.time() - t0
T = int(1.5 * S)
print(json.dumps({"workload": f"{W} threads x {N} blocking translates, B={B}, S={S}, {'pinned staging (copy-free kernels)' if pinned else 'pageable buffers (copies)'}",
                  "target_tokens_per_s": W * N * B * T / dt, "seconds": dt}))
