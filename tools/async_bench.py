"""GPU box: end-to-end rate of the C++ Service (host/Service.{hh,cc}):
token-budget batches from ragged sentences, `workers` double-buffered worker
threads (two contexts + pinned staging each), host buffers in and out (PCIe
included). argv: workers, sentences, shortlist size (0 = full vocabulary; "lex" = a lexical shortlist generated
per batch on the device, ServiceConfig::lexical_shortlist), alignments (1 / 0 / "flat" = ServiceConfig::flat_alignments)."""
import json, os, re, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slimt_amd import build as B, synth

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_sent = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
lexical = len(sys.argv) > 3 and sys.argv[3] == "lex"
n_sl = 0 if lexical or len(sys.argv) <= 3 else int(sys.argv[3])
align = (sys.argv[4] if len(sys.argv) > 4 else "1") in ("1", "flat")
flat = len(sys.argv) > 4 and sys.argv[4] == "flat"  # one [T][len] block per sentence instead of T small vectors
# 256 sentences of 32 tokens: (B + 1) * S <= max_words; SLIMT_SERVICE_MAX_WORDS=1024 = the reference's default (Frontend.hh:21-39),
# with SLIMT_SERVICE_MERGE=k (default 8; 1 = off) consecutive batches of one padded length per launch pair (ServiceConfig::merge_batches)
max_words = int(os.environ.get("SLIMT_SERVICE_MAX_WORDS", str(8192 + 32)))
m = synth.make_model("tiny11", eos_bias=-100.0)  # nobody emits EOS: floor(1.5 * S) tokens per sentence
r = np.random.Generator(np.random.PCG64(5))
exe = B.build_host()
with tempfile.TemporaryDirectory() as d:
    mb, cb, ob = (os.path.join(d, n) for n in ("model.bin", "case.bin", "out.bin"))
    open(mb, "wb").write(synth.write_bin(m))
    # sentence lengths uniform in [lo, hi]: 8..32 by default; SLIMT_SERVICE_MAX_LEN=64 = SURVEY 8(d)'s SECONDARY workload
    # (lengths ~ U{8..64}; the Service's queue cuts them into batches by the reference's rule, Batcher.cc:95-120: sentences by
    # ascending length while (rows + 1) * longest <= max_words, every row padded to the batch's longest)
    lo, hi = int(os.environ.get("SLIMT_SERVICE_MIN_LEN", "8")), int(os.environ.get("SLIMT_SERVICE_MAX_LEN", "32"))
    lens = r.integers(lo, hi + 1, size=n_sent)
    per_req = int(os.environ.get("SLIMT_SERVICE_REQUEST", "512"))  # sentences per translate() call
    with open(cb, "wb") as f:
        f.write(struct.pack("<7If", m.enc_layers, m.dec_layers, m.H, max_words, 128, workers, -(-n_sent // per_req), 1.5))
        for i, n in enumerate(lens):
            if i % per_req == 0:
                f.write(struct.pack("<I", min(per_req, n_sent - i)))
            s = np.concatenate([r.integers(2, m.V, size=n - 1), [0]]).astype(np.uint32)
            f.write(struct.pack("<I", int(n)) + s.tobytes())
        if n_sl:
            sl = synth.make_shortlist(m.V, n_sl)
            f.write(struct.pack("<I", sl.size) + sl.tobytes())
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", SLIMT_SERVICE_REPEAT="1",
               SLIMT_SERVICE_NO_ALIGN="0" if align else "1", SLIMT_SERVICE_STATS="1", SLIMT_SERVICE_DISCARD="1",
               SLIMT_SERVICE_FLAT_ALIGN="1" if flat else "0")
    if lexical:  # about 4000 ids for a batch of 256 x 32 tokens, different for every batch
        lb = os.path.join(d, "lex.bin")
        open(lb, "wb").write(synth.make_lexical_shortlist(m.V, m.V, 100, 1, seed=11, empty_fraction=0.4, min_count=1))
        env["SLIMT_SERVICE_LEXICAL"] = lb
    if os.environ.get("SLIMT_SERVICE_REPLICAS"):
        env["SLIMT_SERVICE_REPLICAS"] = os.environ["SLIMT_SERVICE_REPLICAS"]
    res = subprocess.run([exe, "--async", mb, cb, ob], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr
    stats = re.search(r"service-stats: (.*)", res.stderr)
    merged = re.findall(r"service-stats: merged launches: (.*)", res.stderr)
    host_timing = re.findall(r"host-timing: (.*)", res.stderr)
    ms_cold = float(re.search(r"async: .* translated in ([0-9.]+) ms", res.stderr).group(1))
    ms = float(re.search(r"async-warm: .* translated in ([0-9.]+) ms", res.stderr).group(1))
    toks = int(re.search(r"async-warm-tokens: (\d+)", res.stderr).group(1))  # the warm, timed pass: clients x rounds
print(json.dumps({"workload": f"Service, tiny11 {'lexical shortlist generated per batch on the device' if lexical else 'shortlist ' + str(n_sl) if n_sl else 'full vocabulary'}, {n_sent} ragged "
                              f"sentences ({lo}..{hi} tokens), max_words={max_words}, merge={os.environ.get('SLIMT_SERVICE_MERGE', '8 (default)')}, workers={workers} x 2 contexts, "
                              f"pinned host buffers{' + alignments' if align else ''}{' (one block per sentence)' if flat else ''}",
                  "target_tokens_per_s": toks / ms * 1e3,
                  "ms": ms, "ms_first_pass_with_worker_startup": ms_cold, "target_tokens": toks,
                  "service_stats": stats.group(1) if stats else None, "merged_launches": merged[-1] if merged else None,
                  "host_timing": host_timing[-1] if host_timing else None,
                  "client": (re.findall(r"client 0: (.*)", res.stderr) or [None])[-1]}))
