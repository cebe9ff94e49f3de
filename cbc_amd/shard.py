"""Multi-GPU host path: blocks are independent streams, so they shard across ranks with no
collective on the data path; the one exchange step is the gather of the per-rank bitstreams (and
their per-block sizes) to rank 0, which concatenates them in block order.

Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) with device tensors on the GPU
box, "gloo" with CPU tensors in the CPU test suite.
"""
import numpy as np


def shard_ranges(block_reads, world):
    """Contiguous block ranges per rank, balanced by record count.

    block_reads: per-block record counts (array).  Returns [(b0, b1)] * world with b0 <= b1.
    """
    br = np.asarray(block_reads, dtype=np.int64)
    n = len(br)
    if world <= 1:
        return [(0, n)]
    cum = np.concatenate([[0], np.cumsum(br)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        k = min(max(k, cuts[-1]), n)
        cuts.append(k)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def gather_bitstreams(dist, local_packed, local_sizes, device, dst=0):
    """Gather every rank's compacted payload bytes and per-block sizes to rank `dst`.

    local_packed: 1-D uint8 tensor (this rank's payloads, block order); local_sizes: 1-D int64
    tensor (bytes per block).  Returns (payload uint8 tensor, sizes int64 tensor) on `dst`
    in global block order, (None, None) elsewhere.
    """
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    meta = torch.tensor([local_packed.numel(), local_sizes.numel()], dtype=torch.int64, device=device)
    metas = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta)
    nbytes = [int(m[0].item()) for m in metas]
    nblk = [int(m[1].item()) for m in metas]
    cap_b, cap_n = max(max(nbytes), 1), max(max(nblk), 1)
    pb = torch.zeros(cap_b, dtype=torch.uint8, device=device)
    pb[:local_packed.numel()] = local_packed
    ps = torch.zeros(cap_n, dtype=torch.int64, device=device)
    ps[:local_sizes.numel()] = local_sizes
    if rank == dst:
        gl_b = [torch.empty(cap_b, dtype=torch.uint8, device=device) for _ in range(world)]
        gl_s = [torch.empty(cap_n, dtype=torch.int64, device=device) for _ in range(world)]
    else:
        gl_b = gl_s = None
    dist.gather(pb, gl_b, dst=dst)
    dist.gather(ps, gl_s, dst=dst)
    if rank != dst:
        return None, None
    payload = torch.cat([gl_b[r][:nbytes[r]] for r in range(world)])
    sizes = torch.cat([gl_s[r][:nblk[r]] for r in range(world)])
    return payload, sizes


def encode_sharded(dist, block_reads, encode_range, device, dst=0):
    """The strong-scaling step: ONE dataset, every rank codes its contiguous block range, rank `dst`
    receives the bitstreams in global block order.

    block_reads: per-block record counts of the WHOLE dataset (identical on every rank);
    encode_range(b0, b1) -> (uint8 tensor of the range's payloads in block order, int64 tensor of their
    sizes), both on `device`.  Returns (range, payload, sizes): payload/sizes are None off `dst`.
    bench.py --scaling strong and tests/test_shard_gloo.py both go through here."""
    world, rank = dist.get_world_size(), dist.get_rank()
    b0, b1 = shard_ranges(block_reads, world)[rank]
    local, sizes = encode_range(b0, b1)
    payload, allsizes = gather_bitstreams(dist, local, sizes, device, dst=dst)
    return (b0, b1), payload, allsizes


def encode_sharded_by_contig(dist, pb, encode_blocks, device, dst=0):
    """cfg4's partitioning: whole contigs are dealt to the ranks, largest first to the least
    loaded (PackedBatch.assign_contigs -> cbc_assign_contigs, the rule the C host's `cbc --devices` uses), every
    rank codes the blocks of its contigs, and rank `dst` puts the gathered bitstreams back into global block
    order.  encode_blocks(list of block indices) -> (uint8 tensor of their payloads in that order, int64 tensor
    of sizes), both on `device`.  Returns (my block indices, payload, sizes): the last two in global block order
    on `dst`, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    part = pb.assign_contigs(world)
    mine = pb.blocks_of_part(part, rank)
    local, sizes = encode_blocks(mine)
    payload, allsizes = gather_bitstreams(dist, local, sizes, device, dst=dst)
    if rank != dst:
        return mine, None, None
    # rank order -> block order: every rank's list is known from the (deterministic) assignment
    order = [b for r in range(world) for b in pb.blocks_of_part(part, r)]
    sz = allsizes.tolist()
    starts = [0]
    for v in sz:
        starts.append(starts[-1] + int(v))
    where = {b: i for i, b in enumerate(order)}
    pieces = [payload[starts[where[b]]:starts[where[b] + 1]] for b in range(pb.n_blocks)]
    out_sizes = torch.tensor([int(sz[where[b]]) for b in range(pb.n_blocks)], dtype=torch.int64)
    return mine, (torch.cat(pieces) if pieces else payload[:0]), out_sizes
