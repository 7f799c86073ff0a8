"""N > 1 path of bench.py on CPU: world_size-2 gloo processes. Sentences are
independent, so the multi-GPU path is pure sharding + (MAX time, SUM tokens)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_shards_partitions_exactly():
    from slimt_amd.sharding import plan_shards
    for n, b, w in [(4096, 512, 8), (4096, 256, 8), (1000, 256, 3), (5, 256, 2), (0, 16, 4), (257, 256, 2)]:
        plan = plan_shards(n, b, w)
        assert len(plan) == w
        spans = sorted(s for r in plan for s in r)
        pos = 0
        for start, count in spans:
            assert start == pos and 0 < count <= b
            pos += count
        assert pos == n
        sizes = [len(r) for r in plan]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        plan_shards(10, 0, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from slimt_amd.sharding import plan_shards, reduce_timing
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        plan = plan_shards(1000, 256, world)
        mine = sum(c for _, c in plan[rank])
        tokens = mine * 48
        seconds = 1.0 + rank  # rank 1 is the slow one
        dist.barrier()
        dt, total = reduce_timing(dist, torch.device("cpu"), seconds, tokens)
        if rank == 0:
            print(json.dumps({{"dt": dt, "total": total, "mine": mine}}))
        dist.barrier()
        dist.destroy_process_group()
    """))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["dt"] == 2.0            # MAX over ranks
    assert res["total"] == 1000 * 48   # SUM over ranks: every sentence counted once
    assert res["mine"] == 512          # rank 0: batches 0 and 2 (256 + 256)


def _run_bench(extra, env_extra=None, timeout=300):
    env = dict(os.environ, SLIMT_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    _run_bench.stderr = p.stderr
    return lines


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it must RUN two ranks
    (one process per GPU, gloo rendezvous on 127.0.0.1) and print exactly one JSON
    line with n_gpus = 2. SLIMT_BENCH_DRY replaces the device work by no-ops: what
    is under test is the launcher, the barrier and the MAX / SUM reduction."""
    import json
    lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workers", "4"])
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    # a step = `rounds` (default 4) batches of 256 on each of the 4 workers of each of the 2 ranks
    assert d["config"]["rounds_per_step"] == 4
    assert d["config"]["sentences_per_step_all_gpus"] == 2 * 4 * 4 * 256
    assert d["config"]["tokens_per_step_all_gpus"] == 2 * 4 * 4 * 256 * 48
    assert "dry-run" in d["data"] and "roofline" in d and d["vs_baseline"] is None


def test_bench_fixed_4096_sentences_strong_scaling():
    """BASELINE config 5: the SAME 4096 sentences at every N (8 x 512 on 8 GPUs; here
    2 ranks x 4 batches of 512, and 16 x 256 dealt to 2 ranks)."""
    import json
    for batch, per_rank in ((512, 4), (256, 8)):
        lines = _run_bench(["--gpus", "2", "--steps", "2", "--total-sentences", "4096", "--batch", str(batch)])
        d = json.loads(lines[-1])
        assert d["n_gpus"] == 2 and d["scaling"] == "strong"
        assert d["config"]["sentences_per_step_all_gpus"] == 4096
        assert d["config"]["batches_per_step_per_gpu"] == per_rank


def test_bench_under_a_launcher_environment():
    """The driver's way: torch.distributed.run exports RANK / WORLD_SIZE / MASTER_*;
    bench.py must then NOT spawn but join as the rank it is told to be."""
    import json
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, SLIMT_BENCH_DRY="1", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                                       "--workers", "2", "--rounds", "1"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert outs[1][0].strip() == ""  # only rank 0 prints
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["sentences_per_step_all_gpus"] == 2 * 2 * 256


def test_cpu_sets_are_disjoint_equal_and_cover_ranks():
    from slimt_amd.sharding import cpu_set_for_rank
    avail = [3, 0, 1, 2, 8, 9, 10, 11, 16, 17]
    for world in (1, 2, 3, 4, 5, 8, 10):
        sets = [cpu_set_for_rank(r, world, avail, topology={}) for r in range(world)]
        assert all(len(s) == len(avail) // world for s in sets)
        flat = [c for s in sets for c in s]
        assert len(flat) == len(set(flat)) and set(flat) <= set(avail)
        assert all(s == sorted(s) for s in sets)
    # more ranks than cores: one core each, shared round-robin
    assert [cpu_set_for_rank(r, 4, [5, 7], topology={}) for r in range(4)] == [[5], [7], [5], [7]]
    with pytest.raises(ValueError):
        cpu_set_for_rank(2, 2, avail)
    with pytest.raises(ValueError):
        cpu_set_for_rank(0, 1, [])


def test_cpu_sets_follow_the_host_topology():
    """a two-socket SMT host numbers cores 0..7 = socket 0, 8..15 = socket 1, 16..31 = their SMT siblings. A
    plain cut of the sorted ids put ranks 2, 3 of 8 on socket 1 next to GPUs of socket 0 and made ranks 0 and 4 share
    physical cores. Ordered by (package, physical core): siblings stay together, sockets split in the middle; and with
    the GPUs' own neighbourhoods (local_cpulist) each rank takes cores next to ITS GPU."""
    from slimt_amd.sharding import cpu_set_for_rank
    topo = {}
    for c in range(32):
        topo[c] = ((c % 16) // 8, c % 8)  # cpu c and c + 16 are siblings; 0..7 / 16..23 socket 0
    avail = list(range(32))
    sets = [cpu_set_for_rank(r, 8, avail, topology=topo) for r in range(8)]
    assert sets[0] == [0, 1, 16, 17] and sets[3] == [6, 7, 22, 23] and sets[4] == [8, 9, 24, 25]
    for r, s_ in enumerate(sets):
        assert {topo[c][0] for c in s_} == {r // 4}                 # ranks 0..3 socket 0, 4..7 socket 1
        assert len({topo[c] for c in s_}) == 2 and len(s_) == 4     # two whole physical cores each
    flat = [c for s_ in sets for c in s_]
    assert sorted(flat) == avail
    # GPUs 0..3 hang off socket 1, 4..7 off socket 0 (an unusual slot order): the ranks follow their GPUs
    sock = [[c for c in range(32) if topo[c][0] == k] for k in (0, 1)]
    gpu_cpus = [sock[1]] * 4 + [sock[0]] * 4
    sets = [cpu_set_for_rank(r, 8, avail, topology=topo, gpu_cpus=gpu_cpus) for r in range(8)]
    assert all({topo[c][0] for c in sets[r]} == {1 if r < 4 else 0} for r in range(8))
    assert sorted(c for s_ in sets for c in s_) == avail
    # two ranks on that node use GPUs 0 and 1: both next to socket 1, which they split; fewer GPUs listed than ranks: ignored
    two = [cpu_set_for_rank(r, 2, avail, topology=topo, gpu_cpus=gpu_cpus) for r in range(2)]
    assert all({topo[c][0] for c in s_} == {1} for s_ in two) and sorted(two[0] + two[1]) == sorted(sock[1])
    assert cpu_set_for_rank(1, 2, avail, topology=topo, gpu_cpus=gpu_cpus[:1]) == cpu_set_for_rank(1, 2, avail, topology=topo)


def test_eight_dry_ranks_report_and_reduce():
    """The driver's N = 8 case rehearsed without devices: eight ranks rendezvous over gloo on
    127.0.0.1, pin disjoint core sets, and the one JSON line carries n_ranks_seen = 8 and 8 ranks' worth of tokens."""
    import json
    lines = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--workers", "2", "--rounds", "1"], timeout=600)
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["scaling"] == "weak"
    assert d["config"]["sentences_per_step_all_gpus"] == 8 * 2 * 256
    reports = [json.loads(l[len("bench-rank "):]) for l in _run_bench.stderr.splitlines() if l.startswith("bench-rank ")]
    assert sorted(r["rank"] for r in reports) == list(range(8))
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) >= 8:
        seen = [c for r in reports for c in r["cpus"]]
        assert len(seen) == len(set(seen)) and set(seen) <= set(allowed)
    # BASELINE config 5 on 8 ranks: 4096 sentences, 512 per rank, cut into 4 batches of 128 per rank
    d = json.loads(_run_bench(["--gpus", "8", "--steps", "2", "--total-sentences", "4096", "--batch", "128"], timeout=600)[-1])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["sentences_per_step_all_gpus"] == 4096
    assert d["config"]["batches_per_step_per_gpu"] == 4


def test_four_dry_ranks_pin_disjoint_cores_and_all_report():
    """Readiness for the 8-GPU node without one: four ranks (no device work) each pin themselves to a disjoint share of
    this container's cores before anything else, say on stderr which device / bus / cores they ran on, and the one JSON
    line carries n_ranks_seen = the SUM of ones over the process group -- so the driver's N-GPU run proves N ranks
    reported (/root/reference/slimt/Frontend.cc:207-227 is the worker model mapped one thread per GPU here)."""
    import json
    lines = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--workers", "2", "--rounds", "1"])
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["n_ranks_seen"] == 4
    assert d["config"]["sentences_per_step_all_gpus"] == 4 * 2 * 256
    reports = [json.loads(l[len("bench-rank "):]) for l in _run_bench.stderr.splitlines() if l.startswith("bench-rank ")]
    assert sorted(r["rank"] for r in reports) == [0, 1, 2, 3]
    allowed = sorted(os.sched_getaffinity(0))
    per = len(allowed) // 4
    seen = []
    for r in sorted(reports, key=lambda r: r["rank"]):
        assert r["world"] == 4 and r["local_rank"] == r["rank"]
        assert len(r["cpus"]) == max(per, 1) and set(r["cpus"]) <= set(allowed)
        assert r["tokens_per_step"] == 2 * 256 * 48
        seen += r["cpus"]
    if per >= 1:
        assert len(seen) == len(set(seen)), reports  # disjoint
    # a single process sees itself only, and pins nothing
    one = json.loads(_run_bench(["--steps", "2", "--workers", "2", "--rounds", "1"])[0])
    assert one["n_gpus"] == 1 and one["n_ranks_seen"] == 1
    assert "bench-rank" not in _run_bench.stderr
