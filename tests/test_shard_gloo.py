"""N>1 host path on CPU: world_size-2 gloo, each rank codes its shard of the blocks (through the
lock-step emulation of the kernel body, since there is no GPU here) and rank 0 gathers the
bitstreams; the concatenation must equal the single-process result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_balanced():
    from cbc_amd import shard
    br = [4096] * 10 + [100]
    rs = shard.shard_ranges(br, 4)
    assert rs[0][0] == 0 and rs[-1][1] == len(br)
    for (a, b), (c, d) in zip(rs, rs[1:]):
        assert b == c
    loads = [sum(br[a:b]) for a, b in rs]
    assert max(loads) - min(loads) <= 4096
    assert shard.shard_ranges(br, 1) == [(0, len(br))]
    assert len(shard.shard_ranges([5], 8)) == 8


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import blockref
    from cbc_amd import host, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb = host.synth(21, 1_000_000, 6000, 100, block_reads=512)
    payloads, res = blockref.emu_encode(pb)          # every rank can compute everything; it keeps its shard
    b0, b1 = shard.shard_ranges(pb.blocks["n_reads"], world)[rank]
    mine = payloads[b0:b1]
    local = torch.from_numpy(np.frombuffer(b"".join(mine), dtype=np.uint8).copy()) if mine else torch.zeros(0, dtype=torch.uint8)
    sizes = torch.tensor([len(p) for p in mine], dtype=torch.int64)
    allp, alls = shard.gather_bitstreams(dist, local, sizes, torch.device("cpu"), dst=0)
    if rank == 0:
        q.put((allp.numpy().tobytes() == b"".join(payloads), alls.tolist() == [len(p) for p in payloads], pb.n_blocks))
    dist.destroy_process_group()


def test_gather_two_ranks(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_bytes, ok_sizes, nb = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_bytes and ok_sizes and nb >= 10
