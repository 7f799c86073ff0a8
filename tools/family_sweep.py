#!/usr/bin/env python3
"""does the headline survive a different model? For every synthetic family (slimt_amd.synth.FAMILIES:
weight spread 32 / 48 / 64, a heavy-tailed draw, activation multiplier ranges 2-6 / 4-12 / 8-24, LayerNorm scale spread
0.05 / 0.3) on the headline shape (tiny11, B = 256, S = 32, shortlist 4096, 20 workers):
  * parity: one ragged batch with staggered EOS translated by the device == the CPU checker (PORTABLE order), tokens,
    lengths and alignment rows, bit for bit -- with whatever cache forms the family's accumulators take;
  * which forms they take (slimt_hip_debug_kv_formats of the last batch) and whether a watch switched a form off;
  * target tokens/s (bench.py --family F --forward-steps 0, nobody emits EOS).
Prints one table (stdout) -- committed as profiles/r06_model_families.txt."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
from slimt_amd import capi, synth
from oracle import oracle as O

families = sys.argv[1:] or list(synth.FAMILIES)
print(f"{'family':18s} {'parity':8s} {'int16':>6s} {'int20':>6s} {'int24':>6s} {'tight off':>9s} {'24-bit switch':>13s} {'M tok/s':>8s} {'sustained':>9s}")
for fam in families:
    m = synth.make_model("tiny11", seed=1234, eos_bias=6.0, family=fam)
    gm = capi.Model(m)
    B, S = 96, 32
    sl = synth.make_shortlist(m.V, 2048)
    # a calibration batch first (>= 1024 rows), then the batch that is checked -- through the forms the family gets
    ids0, lens0 = synth.make_batch(m.V, 64, S, seed=7)
    ctx = capi.Context(gm, 96, S)
    ctx.translate(ids0, lens0, sl)
    ids, lens = synth.make_batch(m.V, B, S, seed=11, ragged=True)
    out, ln, al = ctx.translate(ids, lens, sl, want_align=True)
    O.set_mode(O.PORTABLE)
    w_out, w_ln, w_al, _ = O.OracleModel(m).translate(ids, lens, sl, want_align=True)
    O.set_mode(O.FAITHFUL)
    ok = np.array_equal(out, w_out) and np.array_equal(ln, w_ln) and np.array_equal(al, w_al)
    ctx.close()
    gm.close()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--family", fam, "--steps", "20", "--warmup", "5",
                          "--forward-steps", "0", "--no-cpu-baseline", "--sustained-steps", "20"], capture_output=True, text=True)
    if res.returncode != 0:
        print(f"{fam:18s} {'ok' if ok else 'FAILED':8s} bench failed: {res.stderr[-200:]}")
        continue
    d = json.loads(res.stdout.strip().splitlines()[-1])
    kv = (d["roofline"].get("hbm_view") or {}).get("kv_cache_forms_last_batch") or {}
    watch = d.get("kv_watch") or {}
    print(f"{fam:18s} {'ok' if ok else 'FAILED':8s} {100 * kv.get('int16', 0):6.1f} {100 * kv.get('int20', 0):6.1f} {100 * kv.get('int24', 0):6.1f} "
          f"{str(watch.get('tight_layers_off')):>9s} {str(watch.get('switched_to_24_bit')):>13s} {d['value'] / 1e6:8.2f} {d['sustained']['value'] / 1e6:9.2f}", flush=True)
