#!/usr/bin/env python3
"""Headline benchmark: target tokens/s, en-de tiny11 int8 greedy decode at
batch 256 (BASELINE.json), on N MI355X of one node.

A "step" = one pass of the hot path (Model::forward: embed + 6-layer encoder +
greedy decode loop) over one batch of 256 synthetic sentences, inputs already
resident in HBM. N > 1: one process per GPU (torchrun), weights replicated,
each rank translates its own batches, no data-path collective (sentences are
independent); torch.distributed is used only for the barrier and the
max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP events on
the stream the kernel runs on, over the timed region) and `cpu_baseline` (the
CPU port of the reference path, timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import re
import sys
import time

# One HIP stream per translate worker; ROCm multiplexes streams onto 4 hardware
# queues by default, which caps the number of batches really in flight. Must be
# set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5000.0  # dense int8 MFMA, 2x the ~2.5 PF bf16 (MI355X_MICROARCH.md, Matrix cores)
PEAK_HBM_GBS = 8000.0    # HBM3E spec (MI355X_MICROARCH.md, Chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=768)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--preset", default="tiny11")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--src-len", type=int, default=32)
    ap.add_argument("--shortlist", type=int, default=4096, help="0 = full vocabulary")
    ap.add_argument("--profile-kernel", default="auto",
                    help="kernel family bracketed by HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sentences", type=int, default=128, help="sentences per CPU worker")
    ap.add_argument("--workers", type=int, default=20,
                    help="translate contexts (HIP streams) per GPU, like slimt::Async workers "
                         "(Frontend.cc:212-226): independent batches in flight on one device")
    ap.add_argument("--ragged", action="store_true",
                    help="sentence lengths uniform in [S/4, S] instead of all S (not the headline config)")
    ap.add_argument("--decoder-budget", type=int, default=-1,
                    help="decoder workgroups admitted at a time (-1 = library default: 3/4 of the CUs, 0 = no limit)")
    ap.add_argument("--decode-mode", type=int, default=0, help="0 = fused persistent decoder, 1 = step-wise")
    ap.add_argument("--all-kernels", action="store_true",
                    help="after the timed region, time every kernel family (untimed pass)")
    return ap.parse_args()


def algorithmic_macs_per_sentence(D, F, Le, Ld, S, T, N):
    """int8 MACs, cross-attention K/V computed once."""
    return S * (Le * (4 * D * D + 2 * D * F) + Ld * 2 * D * D) + T * (Ld * (4 * D * D + 2 * D * F) + D * N)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC pass
    (profiles/*_pmc_{FETCH,WRITE}_SIZE.json, written by tools/profile_round.sh:
    separate rocprofv3 --pmc runs of this same workload). FETCH_SIZE is doubled
    (gfx950 correction, MI355X_MICROARCH.md, HBM section). None if absent."""
    import glob
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{c}.json")),
                       key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])  # r01_v9 < r01_v13
        if not files:
            return None, None
        rec = json.load(open(files[-1])).get(kernel)
        if not rec:
            return None, None
        out[c] = (rec["avg_KB_per_launch"] * 1024.0, os.path.basename(files[-1]))
    return 2.0 * out["FETCH_SIZE"][0] + out["WRITE_SIZE"][0], [out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]]


def cpu_baseline(model, S, T, n_sl, n_sent):
    """The CPU port of the reference's op sequence (per-step K/V recompute and
    per-call PrepareBias included), FAITHFUL float order, on the host cores --
    run the way slimt runs on a CPU: one single-threaded worker per core, each
    translating its own batch (Async, Frontend.cc:207-227). `n_sent` sentences
    per worker."""
    import threading
    from oracle import oracle as O
    from slimt_amd import synth
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # the CPU share of the container (cgroup v2 quota), not the host's core count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
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
    O.set_mode(O.FAITHFUL)
    sl = synth.make_shortlist(model.V, n_sl) if n_sl else None
    oms = [O.OracleModel(model, reference_cost=True, threads=1) for _ in range(workers)]
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
    dt = time.perf_counter() - t0
    total = sum(toks)
    return {
        "value": total / dt, "unit": "tokens/s", "cores": workers, "kind": "port",
        "sample": f"{workers} single-threaded workers x {n_sent} sentences x S={S}, {max(steps)} decode "
                  f"steps, {total} tokens in {dt:.2f}s; C port of slimt's intgemm op sequence "
                  "(AVX512-VNNI vpdpbusd when available, per-step K/V recompute + PrepareBias as in "
                  "the reference), one batch per worker like slimt's Async",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    from slimt_amd import capi, synth

    if not torch.cuda.is_available() or capi.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # SLIMT_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo -- exercises the
    # torchrun path (rendezvous, barriers, max/sum reduction, rank-0 JSON) on a
    # one-GPU box. Its numbers mean nothing; the real N-GPU run uses RCCL.
    rehearsal = os.environ.get("SLIMT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist_mod.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    B, S = args.batch, args.src_len
    T = int(np.float32(1.5) * np.float32(S))
    model = synth.make_model(args.preset, seed=1234, eos_bias=-100.0)  # nobody emits EOS
    n_sl = args.shortlist
    sl = synth.make_shortlist(model.V, n_sl) if n_sl else None
    N_out = n_sl if n_sl else model.V
    gm = capi.Model(model, device=local_rank)
    if args.decoder_budget >= 0:
        gm.set_decoder_budget(args.decoder_budget)
    W = max(1, args.workers)
    ctxs = [capi.Context(gm, B, S) for _ in range(W)]
    for c in ctxs:
        c.set_decode_mode(args.decode_mode)
    ctx = ctxs[0]
    dev = torch.device("cuda", local_rank)

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)

    # a few distinct batches, resident in HBM before the timed region
    n_batches = 4
    batches = []
    for i in range(n_batches):
        ids, lens = synth.make_batch(model.V, B, S, seed=4321 + 97 * rank + i, ragged=args.ragged)
        batches.append((to_dev(ids), to_dev(lens)))
    d_sl = to_dev(sl) if sl is not None else None
    d_outs = [torch.zeros((B, T), dtype=torch.int32, device=dev) for _ in range(W)]
    d_lens_out = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(W)]
    d_len = d_lens_out[0]

    def step(i):
        # step i runs on worker i % W; calls are asynchronous (fixed step count)
        w = i % W
        d_ids, d_lens = batches[i % n_batches]
        ctxs[w].translate_device(d_ids.data_ptr(), d_lens.data_ptr(), B, S,
                                 d_sl.data_ptr() if d_sl is not None else 0, n_sl, 1.5, 0,
                                 d_outs[w].data_ptr(), d_lens_out[w].data_ptr(), 0, steps_hint=T)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, W)):
        step(i)
    torch.cuda.synchronize()
    tokens_per_step = int(d_len.sum().item())

    kmap = {"gemm_enc": capi.K_GEMM_ENC, "gemm_dec": capi.K_GEMM_DEC, "logits": capi.K_LOGITS,
            "attn_enc": capi.K_ATTN_ENC, "attn_dec": capi.K_ATTN_DEC, "ssru": capi.K_SSRU,
            "decode_fused": capi.K_DECODE_FUSED, "encode_fused": capi.K_ENCODE_FUSED}
    enc_fused, dec_fused = ctx.plan(S)
    auto_kernel = "decode_fused" if dec_fused else "gemm_dec"
    prof_name = auto_kernel if args.profile_kernel == "auto" else args.profile_kernel
    for c in ctxs:
        c.profile_enable(kmap.get(prof_name, capi.K_NONE))

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    prof = {"launches": 0, "total_ms": 0.0, "int8_macs": 0.0, "weight_bytes": 0.0}
    for c in ctxs:
        r = c.profile_read()
        for k in prof:
            prof[k] += r[k]
        c.profile_enable(capi.K_NONE)

    from slimt_amd.sharding import reduce_timing
    dt_max, total_tokens_per_step = reduce_timing(dist, torch.device("cpu") if rehearsal else dev, dt,
                                                  tokens_per_step)

    per_kernel = None
    if args.all_kernels and rank == 0:
        per_kernel = {}
        for name, kid in kmap.items():
            ctx.profile_enable(kid)
            for i in range(2):
                step(i)
            r = ctx.profile_read()
            per_kernel[name] = {"launches_per_step": r["launches"] / 2,
                                "ms_per_step": r["total_ms"] / 2,
                                "avg_us": 1e3 * r["total_ms"] / max(1, r["launches"])}
        ctx.profile_enable(capi.K_NONE)

    if rank == 0:
        value = total_tokens_per_step * args.steps / dt_max
        avg_ms = prof["total_ms"] / max(1, prof["launches"])
        ops = 2.0 * prof["int8_macs"] / max(1, prof["launches"])       # algorithmic int8 OPs / launch
        wbytes = prof["weight_bytes"] / max(1, prof["launches"])       # algorithmic weight bytes / launch
        achieved = ops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        macs_sentence = algorithmic_macs_per_sentence(model.D, model.F, model.enc_layers,
                                                      model.dec_layers, S, T, N_out)
        traffic, traffic_src = pmc_traffic(prof_name)
        cus = -(-B // 16) if prof_name == "decode_fused" else 256
        in_flight = prof["total_ms"] / (1e3 * dt) if dt > 0 else 0.0  # launches of this kernel running at once (this rank)
        mfma = {
            "achieved": achieved, "peak": PEAK_INT8_TOPS, "unit": "TOP/s", "frac": achieved / PEAK_INT8_TOPS,
            "frac_of_occupied_cus": achieved / (PEAK_INT8_TOPS * cus / 256.0),
            "chip_frac_all_kernels": 2.0 * macs_sentence * B * args.steps / dt_max / 1e12 / PEAK_INT8_TOPS,  # per GPU
            "algorithmic_ops_per_launch": ops,
        }
        if prof_name == "decode_fused":
            # The persistent decoder is a streaming kernel (16 rows per workgroup: <10 % MFMA duty).
            # Algorithmic memory-side bytes of one launch: the cross-attention
            # K/V cache is re-read every step and, with >= 8 batches in flight, exceeds the 256 MB
            # Infinity Cache (PMC: profiles/r01_v12_fullocc_pmc.txt); weights once per step and batch
            # (they are shared through L2 by the batch's workgroups); target embeddings; ids out.
            D, F, Ld = model.D, model.F, model.dec_layers
            kv_bytes = float(B) * T * Ld * 2 * S * D * 4
            w_bytes = float(T) * (Ld * (4 * D * D + 2 * D * F) + D * N_out)
            io_bytes = float(B) * T * (D + 4)
            alg_bytes = kv_bytes + w_bytes + io_bytes
            gbs = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roofline = {
                "kernel": prof_name, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_model": {"kv_cache_reread_per_step": kv_bytes, "weights_once_per_step": w_bytes,
                                "embedding_rows_and_ids": io_bytes},
                "launches": prof["launches"], "avg_launch_us": 1e3 * avg_ms,
                "cus_per_launch": cus, "launches_in_flight": in_flight,
                "chip_achieved": gbs * in_flight, "chip_frac": gbs * in_flight / PEAK_HBM_GBS,
                "note": "achieved/frac are per launch of ONE kernel instance, which occupies ceil(B/16) of the "
                        "256 CUs; `launches_in_flight` of them overlap (plus the other workers' encoders), "
                        "chip_* = per-launch rate x launches in flight",
                "l2_stream_bytes_per_launch": wbytes + kv_bytes,  # weights per workgroup and step + K/V
                "l2_stream_GBs_per_cu": (wbytes + kv_bytes) / (avg_ms * 1e-3) / 1e9 / cus if avg_ms > 0 else 0.0,
                "mfma": mfma,
            }
        else:
            roofline = dict(mfma, kernel=prof_name, bound="mfma", traffic=traffic, traffic_source=traffic_src,
                            cus_per_launch=cus, launches=prof["launches"], avg_launch_us=1e3 * avg_ms,
                            launches_in_flight=in_flight, algorithmic_weight_bytes_per_launch=wbytes)
        out = {
            "metric": "target tokens/sec, en-de tiny11 int8 greedy, batch=256",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic",
            "config": {
                "workload": f"en-de {args.preset} int8 greedy decode, batch={B} sentences/GPU, "
                            f"S={S} source tokens, T={T} decode steps, "
                            f"{'shortlist ' + str(n_sl) if n_sl else 'full 32k vocabulary'}",
                "preset": args.preset, "batch_per_gpu": B, "src_len": S, "decode_steps": T,
                "shortlist": n_sl, "parallelism": f"dp{world} (replicated weights, no collective)",
                "workers_per_gpu": W, "decode": "fused-persistent" if dec_fused else "step-wise",
                "encode": "fused-persistent" if enc_fused else "layer-by-layer",
                "tokens_per_step_all_gpus": total_tokens_per_step,
                "int8_ops_per_token": 2.0 * macs_sentence / T,
                "whole_job_int8_tops": 2.0 * macs_sentence * B * world * args.steps / dt_max / 1e12,
            },
            "roofline": roofline,
        }
        if per_kernel is not None:
            out["per_kernel"] = per_kernel
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, S, T, n_sl, args.cpu_sentences)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()
    gm.close()


if __name__ == "__main__":
    main()
