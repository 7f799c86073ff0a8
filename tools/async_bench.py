"""GPU box: end-to-end rate of the C++ Async pipeline (host/Batcher.{hh,cc}):
token-budget batches from ragged sentences, `workers` worker threads each with
its own context, host buffers in and out (PCIe included, alignments returned)."""
import json, os, re, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import build as B, synth

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_sent = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
max_words = 8192  # 256 sentences of 32 tokens
m = synth.make_model("tiny11", eos_bias=-100.0)  # nobody emits EOS: floor(1.5 * S) tokens per sentence
r = np.random.Generator(np.random.PCG64(5))
exe = B.build_host()
with tempfile.TemporaryDirectory() as d:
    mb, cb, ob = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin"))
    open(mb, "wb").write(synth.write_bin(m))
    lens = r.integers(8, 33, size=n_sent)
    with open(cb, "wb") as f:
        f.write(struct.pack("<7If", m.enc_layers, m.dec_layers, m.H, max_words, 128, workers, 1, 1.5))
        f.write(struct.pack("<I", n_sent))
        for n in lens:
            s = np.concatenate([r.integers(2, m.V, size=n - 1), [0]]).astype(np.uint32)
            f.write(struct.pack("<I", int(n)) + s.tobytes())
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32")
    res = subprocess.run([exe, "--async", mb, cb, ob], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr
    ms = float(re.search(r"translated in ([0-9.]+) ms", res.stderr).group(1))
    raw = open(ob, "rb").read()
toks, off = 0, 0
for _ in range(n_sent):
    S, n = struct.unpack_from("<2I", raw, off)
    off += 8 + 4 * n
    toks += n
print(json.dumps({"workload": f"Async, tiny11 full vocabulary, {n_sent} ragged sentences (8..32 tokens), "
                              f"max_words={max_words}, workers={workers}, host buffers + alignments",
                  "sentences_per_s": n_sent / ms * 1e3, "target_tokens_per_s": toks / ms * 1e3,
                  "ms": ms, "target_tokens": toks}))
