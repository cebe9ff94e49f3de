#!/usr/bin/env python3
"""bench.py -- Mbases/s encoded (bit-exact) on synthetic 150 bp SAM, BASELINE.json's metric.

One process per GPU.  A "step" is one pass of the hot path over one resident batch: the encode
launch (one arithmetic stream per workgroup over every block: a model wavefront feeding a coder wavefront) + the device-side compaction of the
per-block bitstreams, and for N > 1 the gather of every rank's bitstreams to rank 0 over RCCL
(the path's one real exchange step).  Inputs are packed and resident in HBM before the timed
region starts.  Work per GPU is fixed as N grows (each rank codes its own shard of `--reads`
records), so scaling is "weak".

Prints ONE JSON line on rank 0 (see the task contract), with two extra objects:
  roofline     -- algorithmic bytes (2*L+18 per read) / HIP-event kernel time vs 8 TB/s
  cpu_baseline -- the oracle (CPU restatement, kind "port") timed single-threaded on this host on a
                  bounded sample of the same workload, SAM parsing included (N=1, rank 0 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CHR1_LEN = 248_956_422          # human chr1-sized contig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="records per GPU (cfg2: 10M)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--contig-len", type=int, default=CHR1_LEN)
    ap.add_argument("--block-reads", type=int, default=4096)
    ap.add_argument("--cpu-sample-reads", type=int, default=10_000_000,
                    help="records of the same workload timed on one host core by the oracle (10 M ~ 11 s of CPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="encode", choices=["encode", "decode"],
                    help="encode = BASELINE.json's metric (default); decode = the mirror kernel on the same workload "
                         "(the payloads are produced by one untimed encode launch and checked to decode to the packed bases)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real path); gloo = rehearsal of the N>1 host logic when "
                         "several ranks must share one GPU (payloads take a detour through host memory)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        if args.gpus != 1 or world != 1:
            sys.exit(2)

    import torch
    import torch.distributed as dist
    from cbc_amd import gpu, host

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the cbc hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    if args.backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks but %d GPUs; RCCL needs one GPU per rank (use --backend gloo to rehearse)"
                         % (world, ndev))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where collectives run
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    # ---- workload: this rank's shard (cfg2 shape), packed on the host, then made resident ----
    t0 = time.time()
    pb = host.synth(0xCBC00002 + rank, args.contig_len, args.reads, args.read_len, 0.003, 0.02, b"chr1",
                    block_reads=args.block_reads)
    t_gen = time.time() - t0
    enc = gpu.Encoder(dev_index)
    L = gpu.lib()
    blocks = pb.blocks.copy()
    scratch_bytes = int(L.cbc_gpu_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, pb.tok.ctypes.data))

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)

    d_recs, d_seq, d_tok = to_dev(pb.recs), to_dev(pb.seq), to_dev(pb.tok)
    d_names, d_blocks, d_ref = to_dev(pb.names), to_dev(blocks), to_dev(pb.ref)
    d_out = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(pb.n_blocks * 16, dtype=torch.uint8, device=dev)
    d_offs = torch.zeros(pb.n_blocks + 1, dtype=torch.int64, device=dev)
    packed_cap = max(1 << 20, 8 * pb.n_recs)
    d_packed = torch.empty(packed_cap, dtype=torch.uint8, device=dev)
    caps = host.LdsCaps(pb.cap_pos, pb.cap_var)
    db = gpu.DeviceBatch(d_recs.data_ptr(), d_seq.data_ptr(), d_tok.data_ptr(), d_names.data_ptr(),
                         d_blocks.data_ptr(), pb.n_blocks, d_ref.data_ptr(), d_ref.numel(),
                         d_out.data_ptr(), scratch_bytes, d_res.data_ptr(), d_seq.numel(), max(pb.n_tok, 1),
                         pb.n_recs, caps)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_bases, n_recs, n_blocks = pb.n_bases, pb.n_recs, pb.n_blocks
    lds_bytes = int(L.cbc_gpu_lds_bytes(ctypes.byref(caps)))

    gather_cap = None
    gather_list = None
    dec = None

    def encode_once():
        enc.encode_device(db, stream)
        enc.compact_device(d_out.data_ptr(), d_blocks.data_ptr(), d_res.data_ptr(), n_blocks, d_offs.data_ptr(),
                           d_packed.data_ptr(), packed_cap, stream)

    def step():
        if args.mode == "decode":
            enc.decode_device(dec["db"], stream)
            return
        encode_once()
        if world > 1:
            dist.gather(d_packed[:gather_cap].to(cdev), gather_list, dst=0)

    # first launch: check every block finished and size the gather
    encode_once()
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(host.RESULT_DTYPE)
    if (res["status"] != 0).any():
        bad = int(np.nonzero(res["status"])[0][0])
        raise SystemExit("block %d failed: status %d at record %d" % (bad, res[bad]["status"], res[bad]["fail_read"]))
    payload_bytes = int(d_offs[-1].item())
    n_symbols = int(res["n_symbols"].sum())
    first_payload = d_packed[:payload_bytes].clone()       # every later step must reproduce these bytes
    if world > 1:
        m = torch.tensor([payload_bytes], dtype=torch.int64, device=cdev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        gather_cap = min(packed_cap, (int(m.item()) + 4095) // 4096 * 4096)
        if rank == 0:
            gather_list = [torch.empty(gather_cap, dtype=torch.uint8, device=cdev) for _ in range(world)]

    if args.mode == "decode":
        # lay the decode launch out over the compacted payloads that are already resident
        stride = (args.read_len + 3) // 4 * 4
        offs = d_offs.cpu().numpy().astype(np.uint64)
        dblocks = np.zeros(n_blocks, dtype=host.DEC_BLOCK_DTYPE)
        dblocks["in_off"] = offs[:-1]
        dblocks["in_bytes"] = (offs[1:] - offs[:-1]).astype(np.uint32)
        dblocks["ref_off"] = blocks["ref_off"]
        dblocks["n_reads"] = blocks["n_reads"]
        rb = np.concatenate([[0], np.cumsum(blocks["n_reads"].astype(np.uint64))])[:-1]
        dblocks["rec_base"] = rb
        dblocks["seq_base"] = rb * stride
        dblocks["read_length"] = args.read_len
        dblocks["seq_stride"] = stride
        dec = {"blocks": to_dev(dblocks),
               "recs": torch.zeros(n_recs * 16, dtype=torch.uint8, device=dev),
               "seq": torch.zeros(n_recs * stride + 16, dtype=torch.uint8, device=dev),
               "res": torch.zeros(n_blocks * 16, dtype=torch.uint8, device=dev),
               "vs": torch.zeros(max(n_blocks * pb.cap_var, 1), dtype=torch.int32, device=dev)}
        dec["db"] = gpu.DecDeviceBatch(d_packed.data_ptr(), packed_cap, dec["blocks"].data_ptr(), n_blocks,
                                       d_ref.data_ptr(), d_ref.numel(), dec["recs"].data_ptr(), n_recs,
                                       dec["seq"].data_ptr(), dec["seq"].numel(), dec["res"].data_ptr(),
                                       dec["vs"].data_ptr(), dec["vs"].numel(), caps)
        enc.decode_device(dec["db"], stream)
        torch.cuda.synchronize()
        dres = dec["res"].cpu().numpy().view(host.RESULT_DTYPE)
        if (dres["status"] != 0).any():
            raise SystemExit("decode failed: %r" % (dres[dres["status"] != 0][:1],))
        got = dec["seq"][:n_recs * stride].view(n_recs, stride)[:, :args.read_len].cpu().numpy()
        if not (got == pb.seq[:n_recs * args.read_len].reshape(n_recs, args.read_len)).all():
            raise SystemExit("decode does not reproduce the packed bases")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(enc.last_kernel_ms())      # HIP events recorded on the launch stream
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_bases, n_recs], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bases, total_recs = int(tot[0].item()), int(tot[1].item())
    else:
        total_bases, total_recs = n_bases, n_recs

    if args.mode == "encode":
        torch.cuda.synchronize()
        if int(d_offs[-1].item()) != payload_bytes or not torch.equal(d_packed[:payload_bytes], first_payload):
            raise SystemExit("bench.py: the timed launches did not reproduce the first launch's bitstreams")

    ms_per_step = elapsed * 1e3 / args.steps
    value = total_bases * args.steps / elapsed / 1e6
    k_ms = float(np.mean(kernel_ms))
    alg_bytes = (2 * args.read_len + 18) * n_recs          # per launch, this rank
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    # HBM traffic per launch: PMC counters cannot be read from inside this process; the committed rocprofv3
    # passes of this same command (profiles/README.md) are reported when the workload is the default one.
    traffic, traffic_src = None, None
    if args.mode == "encode" and args.reads == 10_000_000 and args.read_len == 150 and args.block_reads == 4096:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
        if cands:
            try:
                traffic = json.load(open(cands[-1])).get("hbm_bytes_per_launch")
                traffic_src = os.path.relpath(cands[-1], ROOT)
            except Exception:
                traffic = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle                            # checker, timed as the reported baseline
        sample = min(args.cpu_sample_reads, args.reads)
        spb, sam, fa = host.synth(0xCBC00002, args.contig_len, sample, args.read_len, 0.003, 0.02, b"chr1",
                                  want_text=True, block_reads=args.block_reads)
        t1 = time.perf_counter()
        data, st = oracle.encode(sam, fa, return_stats=True)
        t_cpu = time.perf_counter() - t1
        cpu = {"value": round(st.n_bases / t_cpu / 1e6, 2), "unit": "Mbases/s", "cores": 1, "kind": "port",
               "sample": "%d reads x %d bp of the same synthetic workload%s, one whole-file stream, SAM text "
                         "parsing and FASTA load included, %.1f s of CPU" % (
                             sample, args.read_len, " (all of it)" if sample == args.reads else "", t_cpu),
               "host_cpus": os.cpu_count()}
        spb.close()

    if rank == 0:
        out = {
            "metric": "Mbases/s encoded (bit-exact) on synthetic 150 bp SAM" if args.mode == "encode"
                      else "Mbases/s decoded (round trip verified) on synthetic 150 bp SAM",
            "value": round(value, 2), "unit": "Mbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s: synthetic %d bp SAM, %d reads per GPU vs a chr1-sized (%d bp) uniform-ACGT "
                                   "contig, block-parallel %s" % (
                                       {10_000_000: "cfg2", 49_791_284: "cfg3 (30x)"}.get(args.reads, "custom"),
                                       args.read_len, args.reads, args.contig_len, args.mode),
                       "reads_per_gpu": n_recs, "blocks_per_gpu": n_blocks, "block_reads": args.block_reads,
                       "lds_bytes_per_block": lds_bytes, "payload_bytes_per_gpu": payload_bytes,
                       "bits_per_read": round(payload_bytes * 8.0 / n_recs, 3),
                       "symbols_per_read": round(n_symbols / n_recs, 3),
                       "parallelism": "blocks sharded over %d GPU(s), gather of bitstreams to rank 0 (%s)" % (world, args.backend),
                       "host_pack_seconds": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "cbc_%s_blocks_kernel%s" % (args.mode, "_w6" if args.mode == "encode" and n_blocks > 10 * torch.cuda.get_device_properties(dev).multi_processor_count else ""), "kernel_ms": round(k_ms, 3),
                         "algorithmic_bytes_per_launch": alg_bytes},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
