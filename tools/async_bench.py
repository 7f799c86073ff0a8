"""GPU box: end-to-end rate of the C++ Service (host/Service.{hh,cc}):
token-budget batches from ragged sentences, `workers` double-buffered worker
threads (two contexts + pinned staging each), host buffers in and out (PCIe
included). argv: workers, sentences, shortlist size (0 = full vocabulary),
alignments (1/0)."""
import json, os, re, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import build as B, synth

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_sent = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
n_sl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
align = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
max_words = 8192 + 32  # 256 sentences of 32 tokens: (B + 1) * S <= max_words
m = synth.make_model("tiny11", eos_bias=-100.0)  # nobody emits EOS: floor(1.5 * S) tokens per sentence
r = np.random.Generator(np.random.PCG64(5))
exe = B.build_host()
with tempfile.TemporaryDirectory() as d:
    mb, cb, ob = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin"))
    open(mb, "wb").write(synth.write_bin(m))
    lo = int(os.environ.get("SLIMT_SERVICE_MIN_LEN", "8"))  # sentence lengths uniform in [lo, 32]
    lens = r.integers(lo, 33, size=n_sent)
    with open(cb, "wb") as f:
        f.write(struct.pack("<7If", m.enc_layers, m.dec_layers, m.H, max_words, 128, workers, 1, 1.5))
        f.write(struct.pack("<I", n_sent))
        for n in lens:
            s = np.concatenate([r.integers(2, m.V, size=n - 1), [0]]).astype(np.uint32)
            f.write(struct.pack("<I", int(n)) + s.tobytes())
        if n_sl:
            sl = synth.make_shortlist(m.V, n_sl)
            f.write(struct.pack("<I", sl.size) + sl.tobytes())
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", SLIMT_SERVICE_REPEAT="1",
               SLIMT_SERVICE_NO_ALIGN="0" if align else "1")
    res = subprocess.run([exe, "--async", mb, cb, ob], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr
    ms_cold = float(re.search(r"async: .* translated in ([0-9.]+) ms", res.stderr).group(1))
    ms = float(re.search(r"async-warm: .* translated in ([0-9.]+) ms", res.stderr).group(1))
    raw = open(ob, "rb").read()
toks, off = 0, 0
for _ in range(n_sent):
    S, n = struct.unpack_from("<2I", raw, off)
    off += 8 + 4 * n
    toks += n
print(json.dumps({"workload": f"Service, tiny11 {'shortlist ' + str(n_sl) if n_sl else 'full vocabulary'}, {n_sent} ragged "
                              f"sentences (8..32 tokens), max_words={max_words}, workers={workers} x 2 contexts, "
                              f"pinned host buffers{' + alignments' if align else ''}",
                  "sentences_per_s": n_sent / ms * 1e3, "target_tokens_per_s": toks / ms * 1e3,
                  "ms": ms, "ms_first_pass_with_worker_startup": ms_cold, "target_tokens": toks}))
