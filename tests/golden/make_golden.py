#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (PORTABLE float order).

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot
be built here, so these fixtures pin the ORACLE'S OWN behaviour (a
regression anchor shared by the CPU tests and the GPU tests), not the
reference's. Inputs are regenerated from seeds by slimt_amd.synth; only the
expected outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402
from slimt_amd import synth  # noqa: E402

# name -> (preset, eos_bias, B, S, shortlist, ragged, seed)
CASES = {
    "micro_sl": ("micro", 3.0, 8, 8, 128, True, 11),
    "mini_full": ("mini", 1.0, 6, 10, None, True, 12),
    "tiny11_sl": ("tiny11", 6.0, 8, 12, 1024, True, 13),
}


def run_case(preset, eos_bias, B, S, n_sl, ragged, seed):
    m = synth.make_model(preset, seed=1234, eos_bias=eos_bias)
    om = O.OracleModel(m)
    ids, lens = synth.make_batch(m.V, B, S, seed=seed, ragged=ragged)
    sl = None if n_sl is None else synth.make_shortlist(m.V, n_sl)
    O.set_mode(O.PORTABLE)
    mask = O.make_mask(lens, S)
    emb = om.embed(ids)
    enc = om.encode(emb, mask)
    states = np.zeros((m.dec_layers, B, m.D), dtype=np.float32)
    logits0, attn0 = om.decode_step(enc, mask, states, None, sl)
    out, ln, al, steps = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
    # one op-level known answer from the model's own first FFN weight
    p = m["encoder_l1_ffn_W1"]
    r = np.random.Generator(np.random.PCG64(seed))
    x = r.normal(0, 2.0, size=(5, p.rows)).astype(np.float32)
    aq = float(m["encoder_l1_ffn_W1_QuantMultA"].data[0, 0])
    acc = O.affine_acc(x, p.data, aq)
    y = O.affine(x, p.data, m["encoder_l1_ffn_b1"].data, aq, p.mult)
    O.set_mode(O.FAITHFUL)
    return dict(ids=ids, lens=lens, enc_checksum=np.float64(enc.astype(np.float64).sum()),
                enc_row0=enc[0, 0].copy(), logits0_row0=logits0[0].copy(), attn0_row0=attn0[0].copy(),
                out_ids=out, out_len=ln, align=al, steps=np.int64(steps),
                affine_x=x, affine_acc=acc, affine_y=y)


if __name__ == "__main__":
    for name, cfg in CASES.items():
        res = run_case(*cfg)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
        print(name, "lengths", res["out_len"].tolist(), "steps", int(res["steps"]))
