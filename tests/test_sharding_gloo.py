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
