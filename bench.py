#!/usr/bin/env python3
"""Headline benchmark: target tokens/s, en-de tiny11 int8 greedy decode at
batch 256 (BASELINE.json), on N MI355X of one node.

A "step" = one pass of the hot path (Model::forward: embed + 6-layer encoder +
greedy decode loop) over one batch of 256 synthetic sentences ON EVERY WORKER of
the rank: `workers` independent translate contexts (slimt::Async workers,
Frontend.cc:212-226) each get one batch per step, so a step is
workers x 256 sentences and the timed region of the driver's `--steps 20` is 400
batches (ramp and tail < 2 %). Inputs are resident in HBM before the timed region.

N > 1: one process per GPU, weights replicated, each rank translates its own
batches, NO data-path collective and no RCCL at all (sentences are independent):
ranks line up and reduce (MAX time, SUM tokens) over a gloo (TCP) group.
`python bench.py --gpus N` spawns the N ranks itself (before anything touches
the GPU); under `torch.distributed.run` the ranks come from the environment.

`--total-sentences 4096` = BASELINE config 5 (strong scaling): a fixed set of
sentences is cut into batches of `--batch`, the batches are dealt to the ranks
(slimt_amd.sharding.plan_shards) and a step translates the whole set once.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP events on
the stream the kernel runs on, over the timed region) and `cpu_baseline` (the
CPU port of the reference path, timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# One HIP stream per translate worker; ROCm multiplexes streams onto 4 hardware
# queues by default, which caps the number of batches really in flight. Must be
# set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5000.0  # dense int8 MFMA, 2x the ~2.5 PF bf16 (MI355X_MICROARCH.md, Matrix cores)
PEAK_HBM_GBS = 8000.0    # HBM3E spec (MI355X_MICROARCH.md, Chip-level parameters)
# what a register-resident loop of v_mfma_i32_16x16x64_i8 sustains on the chip (profiles/archive/r02_mfma_i8_rate_probe.jsonl,
# tools/probe_mfma.py: 4.24 POP/s): SURVEY 8(d) asks for the fraction against the nominal AND the measured peak
MEASURED_INT8_TOPS = 4240.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="tiny11")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--src-len", type=int, default=32)
    ap.add_argument("--shortlist", type=int, default=4096, help="0 = full vocabulary")
    ap.add_argument("--total-sentences", type=int, default=0,
                    help="strong scaling (BASELINE config 5): a fixed set of sentences, cut into batches of "
                         "--batch and dealt to the ranks; a step translates the whole set once")
    ap.add_argument("--profile-kernel", default="auto",
                    help="kernel family bracketed by HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sentences", type=int, default=128, help="sentences per CPU worker")
    ap.add_argument("--workers", type=int, default=20,
                    help="translate contexts (HIP streams) per GPU, like slimt::Async workers "
                         "(Frontend.cc:212-226): independent batches in flight on one device")
    ap.add_argument("--family", default="default",
                    help="synthetic model family (slimt_amd.synth.FAMILIES): weight spread / tails, activation multiplier range, "
                         "LayerNorm scale spread -- what the K/V cache forms' hit rates depend on")
    ap.add_argument("--eos-bias", type=float, default=-100.0,
                    help="added to the EOS logit's bias: -100 (default) = nobody emits EOS, every sentence runs T steps (the "
                         "headline); about 6 = sentences end at different steps (Model.cc:127-137 counts what they emit: "
                         "the line's tokens are then the recorded ones, and `decoder_tile_live_fraction` says how full a "
                         "16-sentence decoder tile is on average while it runs)")
    ap.add_argument("--rounds", type=int, default=4,
                    help="translate calls per worker and step (weak scaling): a step is `rounds` passes of the hot path over "
                         "one batch on every worker, so that the driver's 20 timed steps are >= 0.5 s of GPU time (one pass "
                         "is 6.6 ms; VERDICT r05: a 0.13 s timed region is the size of the box-to-box noise)")
    ap.add_argument("--merge", type=int, default=1,
                    help="batches of --batch sentences per translate call (slimt_hip_translate_many_device: ONE encoder and ONE "
                         "decoder launch for all of them, each batch with its own arrays); 1 = slimt_hip_translate_device")
    ap.add_argument("--ragged", action="store_true",
                    help="sentence lengths uniform in [S/4, S] instead of all S (not the headline config)")
    ap.add_argument("--decoder-budget", type=int, default=-1,
                    help="decoder workgroups admitted at a time (-1 = library default, 0 = no limit)")
    ap.add_argument("--decode-mode", type=int, default=0, help="0 = fused persistent decoder, 1 = step-wise")
    ap.add_argument("--encode-rows", type=int, default=0,
                    help="rows per workgroup of the emb-256 encoder: 0 = chosen per call (default), 32, 64")
    ap.add_argument("--kv-format", type=int, default=0,
                    help="decoder K/V cache: 0 = packed 24-bit accumulators where the shape has that form (default), 1 = f32 "
                         "(slimt_hip_model_set_kv_cache_format; same results, tuning)")
    ap.add_argument("--kv-narrow-limit", type=int, default=0,
                    help="diagnostic: accumulators must lie in [-limit, limit) for the 20-bit K/V form (1 = every sentence falls back)")
    ap.add_argument("--kv-tight-limit", type=int, default=-1,
                    help="diagnostic: signed accumulators must lie in [-limit, limit) for the 16-bit K/V form (-1 = library default 2^15, 0 = never tried)")
    ap.add_argument("--kv-policy", type=int, default=0,
                    help="decoder K/V cache loads: 0 = chosen per launch (default), 1 = temporal, 2 = non-temporal")
    ap.add_argument("--xcd-affinity", type=int, default=-1,
                    help="home XCDs of a batch's decoder workgroups: -1 = library default, 0 = off, 1 / 2 / 4")
    ap.add_argument("--adaptive-rows", type=int, default=-1,
                    help="decoder sentences per workgroup by occupancy (8 or 4 while CUs would idle): -1 = library "
                         "default (on), 0 = always 16, 1 = on")
    ap.add_argument("--sustained-steps", type=int, default=40,
                    help="after the timed region, one longer untimed-by-the-driver region of this many steps "
                         "(reported as `sustained`; 0 = skip)")
    ap.add_argument("--forward-steps", type=int, default=20,
                    help="after the timed region: Model::forward as the reference defines it (host buffers in, "
                         "tokens + lengths + alignment rows back in host memory; with one fixed shortlist and with "
                         "a lexical shortlist generated per batch), this many steps each (reported as "
                         "`model_forward*`; 0 = skip), and `single_stream` (one worker alone)")
    ap.add_argument("--all-kernels", action="store_true",
                    help="after the timed region, time every kernel family (untimed pass)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start N fresh rank processes (this
    process has not touched the GPU and never will), wait, return the worst exit code.
    Rank 0's stdout (the JSON line) is this process's stdout."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SLIMT_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def algorithmic_macs_per_sentence(D, F, Le, Ld, S, T, N):
    """int8 MACs, cross-attention K/V computed once."""
    return S * (Le * (4 * D * D + 2 * D * F) + Ld * 2 * D * D) + T * (Ld * (4 * D * D + 2 * D * F) + D * N)


def evidence_manifest():
    """profiles/CURRENT: the ONE evidence set this line cites, written by tools/round_evidence.sh after its
    counter passes ({"tag": ..., "tiny11": {"FETCH_SIZE": file, "WRITE_SIZE": file, "sq_pmc": file}, "base": {...}},
    file names relative to profiles/). No manifest, or a file it names missing: no counter figures (null) --
    nothing is guessed from file names."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "CURRENT")))
    except (OSError, ValueError):
        return {}


def _evidence(preset, key):
    name = (evidence_manifest().get(preset) or {}).get(key)
    if not name:
        return None, None
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))), name
    except (OSError, ValueError):
        return None, None


def pmc_traffic(kernel, preset="tiny11"):
    """HBM-side bytes per launch of `kernel` from the PMC passes profiles/CURRENT names
    (tools/profile_round.sh: separate rocprofv3 --pmc runs of this same workload). FETCH_SIZE is doubled
    (gfx950 correction, MI355X_MICROARCH.md, HBM section). None if absent."""
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rec, name = _evidence(preset, c)
        if rec is None:
            return None, None
        rec = rec.get(kernel) or ((rec.get("encode_tall") or rec.get("encode_wide")) if kernel == "encode_fused" else None)
        if not rec:
            return None, None
        out[c] = (rec["avg_KB_per_launch"] * 1024.0, name)
    return 2.0 * out["FETCH_SIZE"][0] + out["WRITE_SIZE"][0], [out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]]


def sq_counters(kernel, preset="tiny11"):
    """SQ / TCC counter digest of `kernel` from the full-occupancy pass profiles/CURRENT names
    (tools/pmc_sq.sh: separate rocprofv3 --pmc runs, one launch with every CU holding a workgroup). None if absent."""
    doc, name = _evidence(preset, "sq_pmc")
    if doc is None:
        return None, None
    ks = doc.get("kernels", {})
    rec = ks.get(kernel) or ((ks.get("encode_tall") or ks.get("encode_wide")) if kernel == "encode_fused" else None)
    if not rec:
        return None, None
    keep = ("mfma_busy_frac", "valu_busy_frac", "wave_parked_frac", "wave_issue_stall_frac", "wave_issuing_frac",
            "l2_hit_frac", "lds_bank_conflict_frac")
    return {k: rec[k] for k in keep if k in rec}, name


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # the CPU share of the container (cgroup v2 quota), not the host's core count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cores


def cpu_translate(model, S, n_sl, n_sent, workers, reference_cost):
    """`workers` single-threaded oracle workers, each translating its own batch of
    `n_sent` sentences (slimt's Async, Frontend.cc:207-227). Returns tokens, seconds, steps."""
    import threading
    from oracle import oracle as O
    from slimt_amd import synth
    O.set_mode(O.FAITHFUL)
    sl = synth.make_shortlist(model.V, n_sl) if n_sl else None
    oms = [O.OracleModel(model, reference_cost=reference_cost, threads=1) for _ in range(workers)]
    jobs = [synth.make_batch(model.V, n_sent, S, seed=4321 + w) for w in range(workers)]
    toks = [0] * workers
    steps = [0] * workers

    def work(w):
        out, ln, _, st = oms[w].translate(jobs[w][0], jobs[w][1], sl, 1.5, 0)
        toks[w], steps[w] = int(ln.sum()), int(st)

    ts = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    return sum(toks), time.perf_counter() - t0, max(steps)


def cpu_baseline(model, S, T, n_sl, n_sent):
    """The CPU port of the reference's op sequence (per-step K/V recompute and
    per-call PrepareBias included), FAITHFUL float order, on the host cores --
    run the way slimt runs on a CPU: one single-threaded worker per core, each
    translating its own batch (Async, Frontend.cc:207-227). `n_sent` sentences
    per worker. `variants` adds the single-thread rate and the rate with the
    cross-attention K/V cached per batch (what this repo's GPU path does;
    the reference recomputes them every step, Modules.cc:248)."""
    cores = cpu_share()
    # half of the share: above that the quota throttles erratically (measured on a
    # 16-core share: 8 workers 8.9 k tok/s, 10..16 workers 2.9..6.5 k), and the HIP
    # runtime's own threads are still alive in this process
    workers = max(1, int(os.environ.get("SLIMT_CPU_WORKERS", str(max(1, cores // 2)))))
    try:  # keep the port's per-op buffers on the heap: 64 KiB..MiB mmap / munmap pairs per op
        import ctypes  # would make it a page-fault benchmark
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)  # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 30)  # M_TRIM_THRESHOLD
    except OSError:
        pass
    total, dt, steps = cpu_translate(model, S, n_sl, n_sent, workers, True)
    out = {
        "value": total / dt, "unit": "tokens/s", "cores": workers, "kind": "port",
        "cpu_model": cpu_model_string(), "cpu_share_cores": cores,
        "sample": f"{workers} single-threaded workers x {n_sent} sentences x S={S}, {steps} decode "
                  f"steps, {total} tokens in {dt:.2f}s; C port of slimt's intgemm op sequence "
                  "(AVX512-VNNI vpdpbusd when available, per-step K/V recompute + PrepareBias as in "
                  "the reference), one batch per worker like slimt's Async",
    }
    variants = {}
    half = max(16, n_sent // 2)
    tk, d1, _ = cpu_translate(model, S, n_sl, half, 1, True)
    variants["as_reference_1_thread"] = {"value": tk / d1, "cores": 1,
                                         "sample": f"1 worker x {half} sentences, {tk} tokens in {d1:.2f}s"}
    tk, d2, _ = cpu_translate(model, S, n_sl, n_sent, workers, False)
    variants["cached_kv"] = {"value": tk / d2, "cores": workers,
                             "sample": f"{workers} workers x {n_sent} sentences, cross-attention K/V and "
                                       f"prepared bias computed once per batch, {tk} tokens in {d2:.2f}s"}
    tk, d3, _ = cpu_translate(model, S, n_sl, half, 1, False)
    variants["cached_kv_1_thread"] = {"value": tk / d3, "cores": 1,
                                      "sample": f"1 worker x {half} sentences, {tk} tokens in {d3:.2f}s"}
    out["variants"] = variants
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one. Nothing has touched the GPU in this process.
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dry = os.environ.get("SLIMT_BENCH_DRY") == "1"
    # SLIMT_BENCH_REHEARSAL=1: every rank on GPU 0 -- exercises the N-rank path
    # (rendezvous, barriers, max/sum reduction, rank-0 JSON) on a one-GPU box.
    rehearsal = os.environ.get("SLIMT_BENCH_REHEARSAL") == "1"
    # One process per GPU: each rank keeps to its own host cores (a disjoint, equal share of the cores this job may
    # use, by LOCAL_RANK), set before anything touches the GPU or starts a thread. SLIMT_BENCH_PIN=0: leave it alone.
    cpus = None
    if world > 1 and os.environ.get("SLIMT_BENCH_PIN", "1") != "0":
        from slimt_amd.sharding import pin_rank
        cpus = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if rehearsal:
        local_rank = 0

    import numpy as np
    import torch
    from slimt_amd import synth
    from slimt_amd.sharding import count_ranks, device_identity, plan_shards, reduce_timing

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo, not RCCL: the ranks exchange one barrier and two scalars, on the host.
        # gloo announces its connections on stdout; stdout is the JSON line's.
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
            dist_mod.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        dist = dist_mod

    B, S = args.batch, args.src_len
    T = int(np.float32(1.5) * np.float32(S))
    W = max(1, args.workers)
    MG = max(1, min(8, args.merge))  # batches per translate call
    RD = 1 if args.total_sentences > 0 else max(1, args.rounds)  # translate calls per worker and step
    n_sl = args.shortlist
    D, F, H, Le, Ld, V = synth.PRESETS[args.preset]
    N_out = n_sl if n_sl else V
    strong = args.total_sentences > 0
    if strong:
        my_batches = plan_shards(args.total_sentences, B, world)[rank]  # [(start, count)]
        batches_per_step = len(my_batches)
    else:
        my_batches = [(0, B)] * (W * MG * RD)
        batches_per_step = W * MG * RD

    if dry:
        def step(i):
            time.sleep(0.001)

        def sync():
            pass

        tokens_per_step = sum(c for _, c in my_batches) * T
        prof = {"launches": 0, "total_ms": 0.0, "int8_macs": 0.0, "weight_bytes": 0.0}
        enc_fused = dec_fused = False
        prof_name = "none"
        ctxs = []
        gm = None
    else:
        from slimt_amd import capi
        if not torch.cuda.is_available() or capi.device_count() <= 0:
            raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        model = synth.make_model(args.preset, seed=1234, eos_bias=args.eos_bias, family=args.family)  # -100: nobody emits EOS
        sl = synth.make_shortlist(model.V, n_sl) if n_sl else None
        gm = capi.Model(model, device=local_rank)
        if args.decoder_budget >= 0:
            gm.set_decoder_budget(args.decoder_budget)
        if args.kv_policy:
            gm.set_kv_cache_policy(args.kv_policy)
        if args.kv_format:
            gm.set_kv_cache_format(args.kv_format)
        if args.kv_narrow_limit:
            gm.debug_kv_narrow_limit(args.kv_narrow_limit)
        if args.kv_tight_limit >= 0:
            gm.debug_kv_tight_limit(args.kv_tight_limit)
        if args.xcd_affinity >= 0:
            gm.set_xcd_affinity(args.xcd_affinity)
        if args.adaptive_rows >= 0:
            gm.set_adaptive_decoder_rows(bool(args.adaptive_rows))
        rows_ctx = capi.translate_many_rows([B] * MG) if MG > 1 else B  # sentences a context's launches hold
        ctxs = [capi.Context(gm, rows_ctx, S) for _ in range(W)]
        for c in ctxs:
            c.set_decode_mode(args.decode_mode)
            if args.encode_rows:
                c.set_encode_rows(args.encode_rows)
        dev = torch.device("cuda", local_rank)

        def to_dev(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)

        # distinct batches, resident in HBM before the timed region
        if strong:
            total = args.total_sentences
            ids_all, lens_all = synth.make_batch(model.V, total, S, seed=4321, ragged=args.ragged)
            batches = [(to_dev(ids_all[s: s + c]), to_dev(lens_all[s: s + c]), c) for s, c in my_batches]
        else:
            batches = []
            for i in range(4 * MG):
                ids, lens = synth.make_batch(model.V, B, S, seed=4321 + 97 * rank + i, ragged=args.ragged)
                batches.append((to_dev(ids), to_dev(lens), B))
        d_sl = to_dev(sl) if sl is not None else None
        n_slots = max(W, 1) * MG
        d_outs = [torch.zeros((B, T), dtype=torch.int32, device=dev) for _ in range(n_slots)]
        d_lens_out = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(n_slots)]

        def step_merged(i):
            # MG batches per worker and call, each with its own input and output arrays, one launch pair for all
            for w in range(W):
                call = []
                for q in range(MG):
                    d_ids, d_lens, nb = batches[((i * W + w) * MG + q) % len(batches)]
                    call.append((d_ids.data_ptr(), d_lens.data_ptr(), nb, d_sl.data_ptr() if d_sl is not None else 0, n_sl,
                                 d_outs[w * MG + q].data_ptr(), d_lens_out[w * MG + q].data_ptr(), 0))
                ctxs[w].translate_many_device(call, S, 1.5, 0, steps_hint=T)

        def step(i):
            # `rounds` x one batch on every worker (weak) / every batch of this rank's shard, dealt to
            # the workers round-robin (strong); the calls are asynchronous (fixed step count)
            if MG > 1 and not strong:
                for r in range(RD):
                    step_merged(i * RD + r)
                return
            for j in range(batches_per_step):
                w = j % W
                d_ids, d_lens, nb = batches[j] if strong else batches[(i * W + j) % len(batches)]
                ctxs[w].translate_device(d_ids.data_ptr(), d_lens.data_ptr(), nb, S,
                                         d_sl.data_ptr() if d_sl is not None else 0, n_sl, 1.5, 0,
                                         d_outs[w].data_ptr(), d_lens_out[w].data_ptr(), 0, steps_hint=T)

        def sync():
            torch.cuda.synchronize()

        for i in range(max(1, args.warmup)):
            step(i)
        sync()
        tokens_per_step = sum(c for _, c in my_batches) * T  # nobody emits EOS: T tokens per sentence
        live_fraction = None
        if args.eos_bias > -50.0:
            # sentences end when they emit EOS (Model.cc:127-137): count what every distinct batch records, once, and
            # how full the decoder's 16-sentence tiles run (a tile lives until its longest sentence ends)
            per_batch, live_num, live_den = [], 0, 0
            for d_ids, d_lens, nb in batches:
                ctxs[0].translate_device(d_ids.data_ptr(), d_lens.data_ptr(), nb, S, d_sl.data_ptr() if d_sl is not None else 0,
                                         n_sl, 1.5, 0, d_outs[0].data_ptr(), d_lens_out[0].data_ptr(), 0, steps_hint=T)
                sync()
                ol = d_lens_out[0][:nb].cpu().numpy().astype(np.int64)
                per_batch.append(int(ol.sum()))
                for t0 in range(0, nb, 16):
                    tile = ol[t0:t0 + 16]
                    live_num += int(tile.sum())
                    live_den += 16 * int(tile.max())
            live_fraction = live_num / max(1, live_den)
            if strong:
                tokens_per_step = sum(per_batch)
            else:  # the step cycles through the distinct batches evenly (W * rounds * merge calls over len(batches) batches)
                calls = batches_per_step
                tokens_per_step = sum(per_batch[(0 * W + j) % len(batches)] for j in range(calls)) if MG == 1 else \
                    sum(per_batch[q % len(batches)] for q in range(calls))
        elif not strong:
            check = int(d_lens_out[0].sum().item())
            if check != B * T:
                raise SystemExit(f"bench: expected {B * T} tokens from one batch, got {check}")

        kmap = {"gemm_enc": capi.K_GEMM_ENC, "gemm_dec": capi.K_GEMM_DEC, "logits": capi.K_LOGITS,
                "attn_enc": capi.K_ATTN_ENC, "attn_dec": capi.K_ATTN_DEC, "ssru": capi.K_SSRU,
                "decode_fused": capi.K_DECODE_FUSED, "encode_fused": capi.K_ENCODE_FUSED}
        enc_fused, dec_fused = ctxs[0].plan(S)
        auto_kernel = "decode_fused" if dec_fused else "gemm_dec"
        prof_name = auto_kernel if args.profile_kernel == "auto" else args.profile_kernel
        for c in ctxs:
            c.profile_enable(kmap.get(prof_name, capi.K_NONE))

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    kv_forms = kv_watch = None
    if not dry:
        # which form the last batch's K/V caches took, per sentence and decoder layer (slimt_hip_debug_kv_formats)
        seen = ctxs[0].debug_kv_formats(Ld, rows_ctx)
        if seen is not None:
            kv_forms = {"int16": float((seen == 2).mean()), "int20": float((seen == 0).mean()), "int24": float((seen == 1).mean())}
        # the cache forms' watches (engine.cpp): did a layer stop trying the 16-bit form, did the model switch to 24 bits?
        t_off, t_missed, t_sub = gm.debug_kv_tight_watch()
        sw24, w24, sub24 = gm.debug_kv_watch()
        kv_watch = {"tight_layers_off": t_off, "tight_missed": t_missed[:Ld], "tight_submitted": t_sub[:Ld],
                    "switched_to_24_bit": sw24, "needed_24_bit": w24, "sentence_layers": sub24,
                    "recalibrations": None}
        try:  # (an older library loaded through SLIMT_HIP_LIB for an A/B lacks the entry point)
            kv_watch["recalibrations"] = gm.debug_kv_recalibrations()
        except capi.SlimtHipError:
            pass
        prof = {"launches": 0, "total_ms": 0.0, "int8_macs": 0.0, "weight_bytes": 0.0}
        for c in ctxs:
            r = c.profile_read()
            for k in prof:
                prof[k] += r[k]
            c.profile_enable(capi.K_NONE)

    cpu = torch.device("cpu")
    dt_max, total_tokens_per_step = reduce_timing(dist, cpu, dt, tokens_per_step)
    n_ranks_seen = count_ranks(dist, cpu)  # every rank that took part adds one: N on an N-GPU run, whatever n_gpus says
    if world > 1:  # which device and which cores this rank ran on (stderr: stdout is the JSON line's)
        who = {"rank": rank, "local_rank": local_rank, "world": world, "cpus": cpus,
               "tokens_per_step": tokens_per_step, "ms_per_step": 1e3 * dt / args.steps}
        who.update({"device": None, "pci_bus_id": None, "numa_node": None} if dry else device_identity(local_rank))
        print("bench-rank " + json.dumps(who), file=sys.stderr, flush=True)

    tokens_per_pass_all = total_tokens_per_step // RD  # one translate call on every worker of every GPU (the forward regions' step)
    sustained = None
    if args.sustained_steps > 0 and not dry:
        barrier()
        t1 = time.perf_counter()
        for i in range(args.sustained_steps):
            step(i)
        barrier()
        dts, _ = reduce_timing(dist, cpu, time.perf_counter() - t1, tokens_per_step)
        sustained = {"steps": args.sustained_steps, "batches_per_gpu": args.sustained_steps * batches_per_step,
                     "value": total_tokens_per_step * args.sustained_steps / dts, "seconds": dts}

    forward = None
    if args.eos_bias > -50.0:
        args.forward_steps = 0  # (the host-buffer regions check T tokens per sentence)
    if args.forward_steps > 0 and not dry and not strong and MG > 1:
        # Model::forward on host buffers, merged: MG pinned batches per worker through slimt_hip_translate_many_async
        from slimt_amd import capi as _capi
        K = args.forward_steps
        fb, pins = [], []
        for w in range(W):
            group = []
            for q in range(MG):
                ids_h, lens_h = synth.make_batch(model.V, B, S, seed=8000 + 97 * rank + w * MG + q, ragged=args.ragged)
                ps = [_capi._Pinned() for _ in range(5)]
                bufs = (ps[0].array(np.uint32, (B, S)), ps[1].array(np.uint32, (B,)), ps[2].array(np.uint32, (B, T)),
                        ps[3].array(np.uint32, (B,)), ps[4].array(np.float32, (B, T, S)))
                bufs[0][...] = ids_h
                bufs[1][...] = lens_h
                pins.append(ps)
                group.append(bufs)
            fb.append(group)

        def fstep():
            for w in range(W):
                ctxs[w].translate_many_async(fb[w], sl)
        for _ in range(2):
            fstep()
        barrier()
        t = time.perf_counter()
        for _ in range(K):
            fstep()
        barrier()
        d, _ = reduce_timing(dist, cpu, time.perf_counter() - t, tokens_per_step // RD)
        got = int(sum(int(b[3].sum()) for g in fb for b in g))
        if got != W * MG * B * T:
            raise SystemExit(f"bench: merged model_forward produced {got} tokens, expected {W * MG * B * T}")
        forward = {"model_forward": {"value": tokens_per_pass_all * K / d, "ms_per_step": 1e3 * d / K, "steps": K, "alignments": True,
                                     "io": f"{MG} pinned batches per call (slimt_hip_translate_many_async): ids + lengths read from, "
                                           "tokens + lengths + alignment rows written to pinned host memory by the two launches"}}
        for ps in pins:
            for p_ in ps:
                p_.free()
    elif args.forward_steps > 0 and not dry and not strong:
        # Model::forward as the reference's workers call it (Model.cc:111-204): ids and lengths in
        # (pinned) HOST memory, tokens, lengths and the alignment rows of every step (Model.cc:84-108)
        # back in host memory -- slimt_hip_translate_async[_generated] on the same 20 x 256 workload.
        from slimt_amd import capi as _capi
        K = args.forward_steps
        fb = []
        for w in range(W):
            ids_h, lens_h = synth.make_batch(model.V, B, S, seed=8000 + 97 * rank + w, ragged=args.ragged)
            bufs = ctxs[w].pinned_buffers(B, S, 1.5, True)
            bufs[0][...] = ids_h
            bufs[1][...] = lens_h
            fb.append(bufs)
        lex = synth.make_lexical_shortlist(model.V, model.V, 100, 1, seed=11, empty_fraction=0.4, min_count=1)
        gen = _capi.ShortlistGenerator(lex, model.V, model.V, device=local_rank)

        def fregion(with_align, generator):
            def fstep():
                for w in range(W):
                    bufs = fb[w] if with_align else fb[w][:4] + (None,)
                    ctxs[w].translate_async(bufs, shortlist=None if generator else sl, generator=generator)
            for _ in range(2):
                fstep()
            barrier()
            t = time.perf_counter()
            for _ in range(K):
                fstep()
            barrier()
            d, _ = reduce_timing(dist, cpu, time.perf_counter() - t, tokens_per_step // RD)
            got = int(sum(int(fb[w][3].sum()) for w in range(W)))
            if got != W * B * T:
                raise SystemExit(f"bench: model_forward produced {got} tokens, expected {W * B * T}")
            return {"value": tokens_per_pass_all * K / d, "ms_per_step": 1e3 * d / K, "steps": K}

        io = ("ids + lengths read from pinned host memory, tokens + lengths + alignment rows [B,T,S] written to "
              "pinned host memory by the persistent kernels (no copy queued)")
        forward = {
            "model_forward": dict(fregion(True, None), io=io, alignments=True,
                                  shortlist=f"one host list of {n_sl} ids (uploaded when it changes)" if n_sl else "full vocabulary"),
            "model_forward_no_alignments": dict(fregion(False, None), alignments=False),
        }
        if n_sl:
            n_gen = int(gen.generate(np.asarray(fb[0][0]), np.asarray(fb[0][1])).size)
            forward["model_forward_per_batch_shortlist"] = dict(
                fregion(True, gen), io=io, alignments=True,
                shortlist=f"ShortlistGenerator::generate per batch on the device (Model.cc:117-120; synthetic "
                          f"binary lexical shortlist, {n_gen} ids for worker 0's batch), no host shortlist")
        gen.close()
        # one stream alone: batches of one worker back to back
        K1 = max(4, K // 2)
        d_ids, d_lens, nb = batches[0]
        def one():
            ctxs[0].translate_device(d_ids.data_ptr(), d_lens.data_ptr(), nb, S, d_sl.data_ptr() if d_sl is not None else 0,
                                     n_sl, 1.5, 0, d_outs[0].data_ptr(), d_lens_out[0].data_ptr(), 0, steps_hint=T)
        one()
        barrier()
        t = time.perf_counter()
        for _ in range(K1):
            one()
        barrier()
        d1 = time.perf_counter() - t
        forward["single_stream"] = {"value": nb * T * K1 / d1, "ms_per_batch": 1e3 * d1 / K1, "batches": K1,
                                    "note": "one worker alone (one stream, batches back to back), device-resident I/O, this rank"}

    per_kernel = None
    if args.all_kernels and rank == 0 and not dry:
        per_kernel = {}
        for name, kid in kmap.items():
            for c in ctxs:
                c.profile_enable(kid)
            step(0)
            sync()
            launches, ms = 0, 0.0
            for c in ctxs:
                r = c.profile_read()
                launches += r["launches"]
                ms += r["total_ms"]
                c.profile_enable(capi.K_NONE)
            per_kernel[name] = {"launches_per_step": launches, "ms_per_step_sum_over_streams": ms,
                                "avg_us": 1e3 * ms / max(1, launches)}

    if rank == 0:
        value = total_tokens_per_step * args.steps / dt_max
        macs_sentence = algorithmic_macs_per_sentence(D, F, Le, Ld, S, T, N_out)
        sentences_per_step = args.total_sentences if strong else world * sum(c for _, c in my_batches)
        whole_job_tops = 2.0 * macs_sentence * sentences_per_step * args.steps / dt_max / 1e12
        avg_ms = prof["total_ms"] / max(1, prof["launches"])
        ops = 2.0 * prof["int8_macs"] / max(1, prof["launches"])       # algorithmic int8 OPs / launch
        wbytes = prof["weight_bytes"] / max(1, prof["launches"])       # weight bytes streamed / launch
        achieved = ops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # the committed counters are of the default workload of each preset: no figure for others
        profiled = (args.preset in ("tiny11", "base") and B == 256 and S == 32 and n_sl == 4096 and not args.ragged and MG == 1 and
                    args.family == "default" and args.eos_bias <= -50.0)
        traffic, traffic_src = pmc_traffic(prof_name, args.preset) if profiled else (None, None)
        cus = -(-(B * MG) // 16) if prof_name == "decode_fused" else 256
        in_flight = prof["total_ms"] / (1e3 * dt) if dt > 0 else 0.0  # launches of this kernel running at once (this rank)
        # the path is a dense int8 contraction, so the bound is the int8 MFMA
        # roofline. `achieved` = algorithmic int8 OPs of one launch of the dominant kernel /
        # its mean duration (HIP events on its own stream); one launch occupies `cus_per_launch`
        # of the 256 CUs and `launches_in_flight` of them overlap (with the other workers'
        # encoders), so the chip-level figures are given beside it.
        roofline = {
            "kernel": prof_name, "bound": "mfma", "achieved": achieved, "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
            "frac": achieved / PEAK_INT8_TOPS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_ops_per_launch": ops, "launches": prof["launches"], "avg_launch_us": 1e3 * avg_ms,
            "cus_per_launch": cus, "launches_in_flight": in_flight,
            # `frac` prices ONE launch (16 CUs for a batch of 256) against the whole chip; the launches of this
            # kernel that run at once, together, against the same peak:
            "frac_chip": achieved * in_flight / PEAK_INT8_TOPS,
            "measured_peak": MEASURED_INT8_TOPS, "frac_measured_peak": achieved / MEASURED_INT8_TOPS,
            "frac_chip_measured_peak": achieved * in_flight / MEASURED_INT8_TOPS,
            "evidence_tag": evidence_manifest().get("tag") if profiled else None,
            "frac_of_occupied_cus": achieved / (PEAK_INT8_TOPS * cus / 256.0),
            "counters_full_occupancy": sq_counters(prof_name, args.preset)[0] if profiled else None,
            "counters_source": sq_counters(prof_name, args.preset)[1] if profiled else None,
            "whole_job": {"achieved": whole_job_tops / world, "frac": whole_job_tops / world / PEAK_INT8_TOPS,
                          "frac_measured_peak": whole_job_tops / world / MEASURED_INT8_TOPS,
                          "note": "all kernels, per GPU: algorithmic int8 OPs of every translated sentence / "
                                  "wall time of the timed region"},
        }
        if roofline["counters_full_occupancy"]:
            roofline["mfma_busy_frac"] = roofline["counters_full_occupancy"].get("mfma_busy_frac")
        if prof_name == "decode_fused":
            # HBM view of the same kernel. SURVEY 8(d) counts the cross-attention K/V cache as
            # written once and re-read on-chip: `algorithmic_bytes_per_launch` = weights + target
            # embedding rows + K/V once. This implementation re-reads the K/V cache from the
            # memory side every step (9..12 batches decode at a time: their caches exceed the
            # 256 MB Infinity Cache): `implementation_bytes_per_launch`, and its ratio.
            # The cache the decoder re-reads is the packed 24-bit form where the kernels have it
            # (tiny11, S <= 32, not S = 1, 2, 5: slimt_hip_model_set_kv_cache_format), f32 elsewhere;
            # SURVEY's algorithmic K/V stays the f32 tensor it names.
            kv_once = float(B * MG) * Ld * 2 * S * D * 4
            kv24 = (dec_fused and enc_fused and args.kv_format in (0, 2) and ((S + 3) // 4 * 4) * 3 <= S * 4 and
                    ((D == 256 and D // H == 32 and S <= 128) or (D == 512 and D // H == 64 and S <= 32)))
            # ... 20 bits per value where the kernels have the narrow form and the accumulators fit it (every sentence of
            # the synthetic models does: tests/test_gpu_kv_narrow.py)
            kv20 = (kv24 and args.kv_format == 0 and ((D == 256 and S <= 64) or (D == 512 and S <= 32)) and
                    ((S + 7) // 8) * 5120 <= ((S + 3) // 4) * 3072)
            kv_impl = kv_once * (0.625 if kv20 else 0.75 if kv24 else 1.0)
            # ... 16 bits where the tight form has a reader (this launch's last batch says which sentences took which form)
            if kv20 and kv_forms:
                kv_impl = kv_once * (0.5 * kv_forms["int16"] + 0.625 * kv_forms["int20"] + 0.75 * kv_forms["int24"])
            w_once = float(Ld * (4 * D * D + 2 * D * F) + D * N_out)
            io_bytes = float(B * MG) * T * (D + 4)
            alg_bytes = w_once + io_bytes + kv_once
            kv_reread = kv_impl * T
            impl_bytes = kv_reread + w_once * T + io_bytes
            gbs = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roofline["hbm_view"] = {
                "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                "algorithmic_bytes_per_launch": alg_bytes,
                "implementation_bytes_per_launch": impl_bytes,
                "implementation_over_algorithmic": impl_bytes / alg_bytes,
                "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                "kv_cache_forms_last_batch": kv_forms,
                "kv_cache_format": ("int16 accumulators less per-column centres (2 bytes per value; 20- and 24-bit fallbacks per sentence and layer)"
                                    if kv20 and kv_forms and kv_forms["int16"] > 0.5
                                    else "int20 accumulators (2.5 bytes per value; 24-bit fallback per sentence and layer)" if kv20
                                    else "int24 accumulators (3 bytes per value)" if kv24 else "f32"),
                "bytes_model": {"kv_cache_once": kv_once, "kv_cache_reread_every_step": kv_reread,
                                "weights_once": w_once, "weights_once_per_step": w_once * T,
                                "embedding_rows_and_ids": io_bytes},
                "implementation_GBs_per_launch": impl_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                "implementation_GBs_chip": impl_bytes / (avg_ms * 1e-3) / 1e9 * in_flight if avg_ms > 0 else 0.0,
                # the PMC figure (bytes that missed the 4 MB L2s, Infinity-Cache hits included) of one launch, as a rate over
                # the launches in flight together: what the fabric side carries for this kernel
                "fabric_GBs_chip": (traffic / (avg_ms * 1e-3) / 1e9 * in_flight) if traffic and avg_ms > 0 else None,
                "l2_stream_bytes_per_launch": wbytes + kv_reread,  # weights per workgroup and step + K/V
                "l2_stream_GBs_per_cu": (wbytes + kv_reread) / (avg_ms * 1e-3) / 1e9 / cus if avg_ms > 0 else 0.0,
            }
        out = {
            "metric": "target tokens/sec, en-de tiny11 int8 greedy, batch=256",
            "value": value, "unit": "tokens/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps,
            "warmup": max(1, args.warmup), "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "int8", "data": "dry-run: no device work (SLIMT_BENCH_DRY)" if dry else "synthetic",
            "config": {
                "workload": (f"en-de {args.preset} int8 greedy decode, "
                             + (f"{MG} batches of {B} sentences per translate call (one launch pair, own arrays each), " if MG > 1 else
                                f"batch={B} sentences per translate call, ") +
                             f"S={S} source tokens, T={T} decode steps, "
                             f"{'shortlist ' + str(n_sl) if n_sl else 'full 32k vocabulary'}; "
                             + (f"one step = the same {args.total_sentences} sentences cut into batches of {B} "
                                f"and dealt to {world} GPU(s)" if strong else
                                f"one step = {RD} translate call(s) on each of the {W} workers of every GPU "
                                f"({W * MG * RD * B} sentences per GPU and step)")),
                "preset": args.preset, "batch": B, "merged_batches_per_call": MG, "src_len": S, "decode_steps": T,
                "shortlist": n_sl, "parallelism": f"dp{world} (replicated weights, no collective, no RCCL)",
                "workers_per_gpu": W, "rounds_per_step": RD, "batches_per_step_per_gpu": batches_per_step,
                "sentences_per_step_all_gpus": sentences_per_step,
                "decode": "fused-persistent" if dec_fused else "step-wise",
                "encode": "fused-persistent" if enc_fused else "layer-by-layer",
                "tokens_per_step_all_gpus": total_tokens_per_step,
                "int8_ops_per_token": 2.0 * macs_sentence / T,
                "whole_job_int8_tops": whole_job_tops,
            },
            "roofline": roofline,
        }
        if not dry and args.family != "default":
            out["config"]["family"] = {"name": args.family, **synth.FAMILIES[args.family]}
        if not dry and args.eos_bias > -50.0:
            out["config"]["eos_bias"] = args.eos_bias
            out["decoder_tile_live_fraction"] = live_fraction
            out["config"]["workload"] += (f"; EOS bias {args.eos_bias}: sentences end at different steps, tokens counted as recorded "
                                          f"(Model.cc:127-137), mean {tokens_per_step / max(1, sum(c for _, c in my_batches)):.1f} per sentence")
        if kv_watch is not None:
            out["kv_watch"] = kv_watch
        if sustained is not None:
            out["sustained"] = sustained
        if forward is not None:
            out.update(forward)
        if per_kernel is not None:
            out["per_kernel"] = per_kernel
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(model, S, T, n_sl, args.cpu_sentences)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()
    if gm is not None:
        gm.close()


if __name__ == "__main__":
    main()
