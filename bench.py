#!/usr/bin/env python3
"""bench.py -- Mbases/s encoded (bit-exact) on synthetic 150 bp SAM, BASELINE.json's metric.

One process per GPU.  A "step" is one pass of the hot path over one resident batch: the encode
launch (one arithmetic stream per workgroup over every block: a model wavefront feeding a coder
wavefront) + the device-side compaction of the per-block bitstreams, and for N > 1 the gather of
every rank's bitstreams to rank 0 over RCCL (the path's one real exchange step).  Inputs are packed
and resident in HBM before the timed region starts.

  python bench.py --gpus N ...        launched bare: the parent starts N rank processes of itself
                                      (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE anything
                                      touches the GPU, supervises them (the first rank that fails ends
                                      the others), forwards rank 0's JSON line and exits non-zero if any
                                      rank does
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   also works

--scaling weak (default): every rank codes its own cfg2-sized shard (`--reads` records per GPU, seed +
rank), work per GPU fixed as N grows.  --scaling strong: ONE dataset of `--reads` records is cut into
contiguous block ranges (cbc_amd.shard.shard_ranges); the container gathered on rank 0 is checked byte
for byte against a single-GPU encode of the whole dataset.

Prints ONE JSON line on rank 0 (see the task contract), with extra objects:
  roofline     -- algorithmic bytes (2*L+18 per read) / HIP-event kernel time vs 8 TB/s,
                  plus `issue`: the instruction-issue picture of the committed SQ counter pass (nulled
                  when that pass was taken from other kernel sources than the ones built here)
  cpu_baseline -- CPU legs timed single-threaded on this host on a bounded sample of the same workload
                  (N=1, rank 0 only): the oracle on SAM text (parse included) and the packed-input CPU
                  port (parse excluded).  The CPU port codes the very blocks the GPU coded and its bytes
                  are compared with the GPU's: `verified_vs_cpu_port` (the run fails on a mismatch)
  e2e          -- SURVEY 8d's other two timed regions, measured in the run outside the headline region
                  (N=1): the host-buffer entry point (PCIe included) and the `cbc` CLI on SAM text
  rccl         -- N > 1: backend, world size, per-rank device / kernel ms / payload bytes, and the
                  checksum every rank took of its payload on the device before the gather next to the
                  one rank 0 recomputed on what the collective delivered (the run fails on a mismatch)
  cfg4         -- N > 1: BASELINE config 4 (GRCh38-shaped, chromosome-sharded, strong scaling) run
                  after the headline region through the same exchange, so that one SCALE run reports it
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CHR1_LEN = 248_956_422          # human chr1-sized contig
# what each measured kernel is compiled from (the kernels are separate functions of one translation unit: a change to the
# long-read body does not touch the block kernels' code)
_BLOCK_SOURCES = ["cbc_gpu.hip", "cbc_encode_body.h", "cbc_decode_body.h", "cbc_plan.h", "cbc_wave_gpu.h"]
KERNEL_SOURCES = {"encode": _BLOCK_SOURCES, "decode": _BLOCK_SOURCES,
                  "long_encode": _BLOCK_SOURCES + ["cbc_long_body.h"], "long_decode": _BLOCK_SOURCES + ["cbc_long_body.h"]}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="records per GPU (weak) or in total (strong); cfg2: 10M")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--contig-len", type=int, default=CHR1_LEN)
    ap.add_argument("--block-reads", type=int, default=4096)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--scale", type=float, default=None,
                    help="cfg4: fraction of GRCh38 (contig lengths and read counts both scaled, coverage stays 30x); 1.0 = 618 M reads. "
                         "Default 0.05 for --workload cfg4, 0.1 for the cfg4 pass of a multi-GPU run")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg5", "cfg4"],
                    help="cfg2 = BASELINE.json's headline (150 bp reads, the reference's format, default); cfg5 = 1 M x 10 kb long "
                         "reads with a 5 %% indel + substitution mix through the long-read format extension (no reference parity "
                         "exists for it: the reference cannot code such reads); cfg4 = the 24 GRCh38 primary contigs at 30x, whole "
                         "contigs dealt to the ranks largest first (chromosome sharding), --scale of the full size")
    ap.add_argument("--cpu-sample-reads", type=int, default=10_000_000,
                    help="records of the same workload timed on one host core by the CPU legs (10 M ~ 11 s each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg4", action="store_true", help="N > 1: skip the cfg4 pass after the headline region")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the PCIe-inclusive and CLI legs after the headline region")
    ap.add_argument("--e2e-reads", type=int, default=4_000_000, help="records of the CLI leg (SAM text is 360 B per record)")
    ap.add_argument("--mode", default="encode", choices=["encode", "decode"],
                    help="encode = BASELINE.json's metric (default); decode = the mirror kernel on the same workload "
                         "(the payloads are produced by one untimed encode launch and checked to decode to the packed bases)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: build the one-rank process group anyway and go through every collective of the N > 1 path "
                         "(gather of the bitstreams, checksum exchange, cfg4 pass) -- how the RCCL call sites are rehearsed on a one-GPU box")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real path); gloo = rehearsal of the N>1 host logic when "
                         "several ranks must share one GPU (payloads take a detour through host memory)")
    a = ap.parse_args(argv)
    if a.workload == "cfg5":                                  # one 100 Mb contig, 1 M x 10 kb, 5 % edits
        if a.reads == 10_000_000:
            a.reads = 1_000_000
        if a.read_len == 150:
            a.read_len = 10_000
        if a.contig_len == CHR1_LEN:
            a.contig_len = 100_000_000
        if a.block_reads == 4096:
            a.block_reads = 64
        if a.cpu_sample_reads == 10_000_000:
            a.cpu_sample_reads = 100_000                      # 1 Gbase: ~10 s of CPU
    return a


def spawn_ranks(args):
    """`python bench.py --gpus N` launched bare: start N rank processes of this script and supervise them.  Nothing in
    this parent imports torch or touches HIP.  The first rank that exits non-zero ends the others (terminate, then kill):
    ranks left waiting in a collective would otherwise hold the GPUs until the collective's own timeout."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None:
        rcs = [p.poll() for p in procs]
        for r, rc in enumerate(rcs):
            if rc is not None and rc != 0:
                failed = (r, rc)
                break
        if all(rc is not None for rc in rcs):
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    if out0 and out0[0]:
        sys.stdout.write(out0[0])
        sys.stdout.flush()
    if failed is not None:
        print("bench.py: rank %d failed with rc %d; the other ranks were stopped" % failed, file=sys.stderr)
        return max(1, abs(failed[1]) & 0xff or 1)
    return 0


def kernel_source_sha(leg):
    """sha256 over the sources the leg's kernel (encode | decode | long_encode | long_decode) is built from: ties a
    committed counter pass to a kernel build."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[leg]:
        with open(os.path.join(ROOT, "cbc_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def issue_picture(root, which, kernel_ms, n_recs):
    """roofline.traffic / roofline.issue: what the committed counter passes of this same command say (HBM bytes per
    launch; the issue ports, the number that binds here).  PMC counters cannot be read from inside this process, so the
    passes are files under profiles/ named by profiles/CURRENT.json; a pass stamped with another kernel-source hash than
    the sources in this tree is stale and reported as null.  Counter values are per launch, summed over the chip."""
    try:
        cur = json.load(open(os.path.join(root, "profiles", "CURRENT.json")))
        path = os.path.join(root, "profiles", cur[which])
        pm = json.load(open(path))
    except Exception:
        return None, None, None
    src = os.path.relpath(path, root)
    if pm.get("kernel_source_sha") != kernel_source_sha(which[:-4]):
        return None, src + " (stale: taken from other kernel sources)", None
    sq = pm.get("SQ_per_launch") or {}
    issue = None
    if sq.get("SQ_INSTS_SALU") and sq.get("SQ_BUSY_CYCLES"):
        n_cu, simd_per_cu = 256, 4
        # shader cycles of one launch: the pass's own busy-cycle counter where it has one (per XCD-SE average), else
        # the kernel time at the 2.4 GHz engine clock
        clock_ghz = pm.get("engine_clock_ghz") or 2.4
        kcyc = (pm.get("kernel_ms") or kernel_ms) * clock_ghz * 1e6
        salu, valu = sq["SQ_INSTS_SALU"], sq["SQ_INSTS_VALU"]
        issue = {"source": src, "engine_clock_ghz": clock_ghz,
                 "salu_per_record": round(salu / n_recs, 1), "valu_per_record": round(valu / n_recs, 1),
                 "lds_per_record": round(sq.get("SQ_INSTS_LDS", 0) / n_recs, 2),
                 "waves_per_launch": sq.get("SQ_WAVES")}
        if kcyc:
            # What the issue ports sustain was measured on the box (tools/ub3.hip, profiles/r03_ubench_issue.log; whole
            # wavefront instructions per cycle of a 2.4 GHz clock from the launches' event times, any residency from 2
            # wavefronts per SIMD up): scalar 0.98 per CU (one scalar unit); vector 0.41 per SIMD for add / and / xor /
            # mov, 0.24 for shifts, multiplies, compares, DPP, lane reads and three-operand forms -- most of what these
            # kernels issue; scalar + vector pairs 1.7 per CU together.
            n_simd = n_cu * simd_per_cu
            issue["salu_per_cycle_per_cu"] = round(salu / (kcyc * n_cu), 3)
            issue["valu_per_cycle_per_simd"] = round(valu / (kcyc * n_simd), 3)
            issue["inst_per_cycle_per_cu"] = round((salu + valu) / (kcyc * n_cu), 3)
            issue["inst_per_cycle_per_simd"] = round((salu + valu) / (kcyc * n_simd), 3)
            issue["measured_peaks"] = {"salu_per_cycle_per_cu": 0.98, "valu_per_cycle_per_simd": [0.24, 0.41], "mixed_per_cycle_per_cu": 1.7,
                                       "source": "profiles/r03_ubench_issue.log"}
            issue["salu_busy"] = round(salu / (kcyc * n_cu) / 0.98, 3)
            issue["valu_busy"] = [round(valu / (kcyc * n_simd) / 0.41, 3), round(valu / (kcyc * n_simd) / 0.24, 3)]     # if all were full-rate / all half-rate
            issue["frac_of_mixed_issue_peak"] = round((salu + valu) / (kcyc * n_cu) / 1.7, 3)
            if sq.get("SQ_WAVE_CYCLES"):
                issue["mean_waves_per_simd"] = round(sq["SQ_WAVE_CYCLES"] * 4 / (kcyc * n_cu * simd_per_cu), 2)
                for key, name in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_issue_stall_frac"),
                                  ("SQ_ACTIVE_INST_ANY", "wave_issuing_frac")):
                    if sq.get(key):
                        issue[name] = round(sq[key] / sq["SQ_WAVE_CYCLES"], 3)
    return pm.get("hbm_bytes_per_launch"), src, issue


class Rank:
    """What every workload needs of this process: its place in the job, its device, the process group."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world), file=sys.stderr)
            sys.exit(2)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the cbc hot path has no CPU fallback")
        ndev = torch.cuda.device_count()
        self.dev_index = self.local_rank % max(ndev, 1)
        if args.backend == "nccl" and self.world > ndev:
            raise SystemExit("bench.py: %d ranks but %d GPUs; RCCL needs one GPU per rank (use --backend gloo to rehearse)"
                             % (self.world, ndev))
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        self.backend = args.backend
        self.cdev = self.dev if args.backend == "nccl" else torch.device("cpu")     # where collectives run
        self.dist = dist
        self.multi = self.world > 1 or args.force_dist       # the exchange path runs (one rank: a self-gather)
        if self.multi:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=self.dev)
            else:
                dist.init_process_group(backend="gloo")
        from cbc_amd import gpu
        self.enc = gpu.Encoder(self.dev_index)
        self.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._d_sum = torch.zeros(1, dtype=torch.int64, device=self.dev)

    def device_checksum(self, t, nbytes):
        """cbc_gpu_checksum_device over the first nbytes of a device tensor (as a signed 64-bit Python int)."""
        import torch
        self.enc.checksum_device(t.data_ptr(), int(nbytes), self._d_sum.data_ptr(), self.stream)
        torch.cuda.current_stream().synchronize()
        return int(self._d_sum.item())

    def checksum_of(self, t, nbytes):
        """The same checksum of a tensor wherever it lives (device: the kernel; host: libcbc_host's twin)."""
        from cbc_amd import host
        if t.is_cuda:
            return self.device_checksum(t, nbytes)
        c = host.checksum64(t[:int(nbytes)].numpy())
        return c - (1 << 64) if c >= (1 << 63) else c

    def verify_exchange(self, gather_list, payload_bytes, my_sum, extra):
        """After the timed region: every rank reports (bytes, checksum taken on its device BEFORE the gather, extras);
        rank 0 recomputes the checksum on what the last gather delivered.  Returns the per-rank table on rank 0 and
        raises on every rank when anything differs."""
        import torch
        dist = self.dist
        meta = torch.tensor([int(payload_bytes), int(my_sum)] + [int(x) for x in extra], dtype=torch.int64, device=self.cdev)
        metas = [torch.zeros_like(meta) for _ in range(self.world)]
        dist.all_gather(metas, meta)
        ok, table = True, []
        if self.rank == 0:
            for r in range(self.world):
                nb, want = int(metas[r][0].item()), int(metas[r][1].item())
                got = self.checksum_of(gather_list[r], nb)
                table.append({"rank": r, "payload_bytes": nb, "checksum_sent": "%016x" % (want & (2 ** 64 - 1)),
                              "checksum_received": "%016x" % (got & (2 ** 64 - 1)), "extra": [int(v) for v in metas[r][2:].tolist()]})
                ok = ok and got == want
        flag = torch.tensor([0 if ok else 1], dtype=torch.int64, device=self.cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            if self.rank == 0:
                print("bench.py: the bitstreams rank 0 received differ from what the ranks sent: %r" % (table,), file=sys.stderr)
            raise SystemExit(3)
        return table

    def device_table(self):
        """Per-rank device identity, gathered to every rank (small Python objects)."""
        import torch
        p = torch.cuda.get_device_properties(self.dev)
        me = {"rank": self.rank, "device_index": self.dev_index, "device": torch.cuda.get_device_name(self.dev),
              "arch": getattr(p, "gcnArchName", None), "pci_bus_id": getattr(p, "pci_bus_id", None),
              "compute_units": p.multi_processor_count}
        if not self.multi:
            return [me]
        out = [None] * self.world
        self.dist.all_gather_object(out, me)
        return out


def run_rank(args, R):
    import numpy as np
    import torch
    from cbc_amd import gpu, host, shard
    dist, rank, world, dev, cdev, enc, stream = R.dist, R.rank, R.world, R.dev, R.cdev, R.enc, R.stream
    multi = R.multi
    strong = args.scaling == "strong" and world > 1

    # ---- workload (cfg2 shape), packed on the host, then made resident ----
    t0 = time.time()
    long_fmt = args.workload == "cfg5"
    seed = (0xCBC00005 if long_fmt else 0xCBC00002) + (0 if strong else rank)
    if long_fmt:
        pb = host.synth_long(seed, args.contig_len, args.reads, args.read_len, 0.05, b"chrL", block_reads=args.block_reads)
    else:
        pb = host.synth(seed, args.contig_len, args.reads, args.read_len, 0.003, 0.02, b"chr1", block_reads=args.block_reads)
    t_gen = time.time() - t0
    L = gpu.lib()
    blocks = pb.blocks.copy()
    if long_fmt:
        scratch_bytes = int(L.cbc_gpu_long_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, 8))
    else:
        scratch_bytes = int(L.cbc_gpu_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, pb.tok.ctypes.data))
    b0, b1 = shard.shard_ranges(blocks["n_reads"], world)[rank] if strong else (0, pb.n_blocks)

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)

    d_recs, d_seq, d_tok = to_dev(pb.recs), to_dev(pb.seq), to_dev(pb.tok)
    d_names, d_blocks, d_ref = to_dev(pb.names), to_dev(blocks), to_dev(pb.ref)
    d_out = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
    d_res = torch.full((pb.n_blocks * 16,), 0xff, dtype=torch.uint8, device=dev)   # a block that never reports reads as failed
    d_offs = torch.zeros(pb.n_blocks + 1, dtype=torch.int64, device=dev)
    packed_cap = max(1 << 20, 8 * pb.n_recs, pb.n_bases // 8 if long_fmt else 0)
    d_packed = torch.empty(packed_cap, dtype=torch.uint8, device=dev)
    caps = host.LdsCaps(pb.cap_pos, pb.cap_var)

    def batch(lo, hi):
        return gpu.DeviceBatch(d_recs.data_ptr(), d_seq.data_ptr(), d_tok.data_ptr(), d_names.data_ptr(),
                               d_blocks.data_ptr() + 64 * lo, hi - lo, d_ref.data_ptr(), d_ref.numel(),
                               d_out.data_ptr(), scratch_bytes, d_res.data_ptr() + 16 * lo, d_seq.numel(), max(pb.n_tok, 1),
                               pb.n_recs, caps)

    db = batch(b0, b1)
    my_blocks = b1 - b0
    n_recs = int(blocks["n_reads"][b0:b1].sum())
    n_bases = int(pb.info["n_bases"][b0:b1].sum())
    lds_bytes = int((L.cbc_gpu_long_lds_bytes if long_fmt else L.cbc_gpu_lds_bytes)(ctypes.byref(caps)))

    gather_cap = None
    gather_list = None
    dec = None

    # HIP events around the dominant kernel of every timed step, on the stream it is launched on (torch's current stream),
    # read only after the timed region: no host synchronisation inside it
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def encode_once(d=None, lo=None, n=None, ev=None):
        d, lo, n = (db, b0, my_blocks) if d is None else (d, lo, n)
        if n == 0:
            return
        if ev is not None:
            ev[0].record()
        (enc.encode_long_device if long_fmt else enc.encode_device)(d, stream)
        if ev is not None:
            ev[1].record()
        enc.compact_device(d_out.data_ptr(), d_blocks.data_ptr() + 64 * lo, d_res.data_ptr() + 16 * lo, n,
                           d_offs.data_ptr(), d_packed.data_ptr(), packed_cap, stream)

    def step(ev=None):
        if args.mode == "decode":
            if ev is not None:
                ev[0].record()
            (enc.decode_long_device if long_fmt else enc.decode_device)(dec["db"], stream)
            if ev is not None:
                ev[1].record()
            return
        encode_once(ev=ev)
        if multi:
            dist.gather(d_packed[:gather_cap].to(cdev), gather_list, dst=0)

    # first launch: check every block finished and size the gather
    encode_once()
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(host.RESULT_DTYPE)[b0:b1]
    if (res["status"] != 0).any():
        bad = int(np.nonzero(res["status"])[0][0])
        raise SystemExit("block %d failed: status %d at record %d" % (b0 + bad, res[bad]["status"], res[bad]["fail_read"]))
    offs_first = d_offs.cpu().numpy().astype(np.uint64)[:my_blocks + 1]
    payload_bytes = int(offs_first[my_blocks]) if my_blocks else 0
    n_symbols = int(res["n_symbols"].sum())
    first_payload = d_packed[:payload_bytes].clone()       # every later step must reproduce these bytes
    first_sum = R.device_checksum(d_packed, payload_bytes) if multi else None
    strong_check = None
    if multi:
        m = torch.tensor([payload_bytes], dtype=torch.int64, device=cdev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        gather_cap = min(packed_cap, (int(m.item()) + 4095) // 4096 * 4096)
        if rank == 0:
            gather_list = [torch.empty(gather_cap, dtype=torch.uint8, device=cdev) for _ in range(world)]
    if strong:
        # the container rank 0 assembles from the ranks' ranges == what ONE GPU produces for the whole dataset
        def my_range(lo, hi):
            assert (lo, hi) == (b0, b1)
            return first_payload.to(cdev), torch.from_numpy(res["nbytes"].astype(np.int64)).to(cdev)
        _, allp, alls = shard.encode_sharded(dist, blocks["n_reads"], my_range, cdev, dst=0)
        if rank == 0:
            whole = batch(0, pb.n_blocks)
            encode_once(whole, 0, pb.n_blocks)
            torch.cuda.synchronize()
            full_bytes = int(d_offs[pb.n_blocks].item())
            full = d_packed[:full_bytes].to(cdev)
            strong_check = bool(allp.numel() == full_bytes and torch.equal(allp, full))
            if not strong_check:
                raise SystemExit("bench.py: the gathered container differs from the single-GPU container")
            encode_once()                                  # leave this rank's own range in d_packed again
            torch.cuda.synchronize()

    if args.mode == "decode":
        # lay the decode launch out over the compacted payloads that are already resident
        stride = (args.read_len + 3) // 4 * 4
        offs = offs_first
        dblocks = np.zeros(my_blocks, dtype=host.DEC_BLOCK_DTYPE)
        dblocks["in_off"] = offs[:-1]
        dblocks["in_bytes"] = (offs[1:] - offs[:-1]).astype(np.uint32)
        dblocks["ref_off"] = blocks["ref_off"][b0:b1]
        dblocks["n_reads"] = blocks["n_reads"][b0:b1]
        rb = np.concatenate([[0], np.cumsum(blocks["n_reads"][b0:b1].astype(np.uint64))])[:-1]
        dblocks["rec_base"] = rb
        if long_fmt:                                          # bases are written compactly: block base = bases before it
            nbv = pb.info["n_bases"][b0:b1].astype(np.uint64)
            dblocks["seq_base"] = np.concatenate([[0], np.cumsum((nbv + 7) & ~np.uint64(7))])[:-1]
            dblocks["reserved"][:, 0] = nbv.astype(np.uint32)
            stride = 0
            seq_out_bytes = int(((nbv + 7) & ~np.uint64(7)).sum()) + 16
        else:
            dblocks["seq_base"] = rb * stride
            seq_out_bytes = n_recs * stride + 16
        dblocks["read_length"] = min(args.read_len, 256)
        dblocks["seq_stride"] = stride
        dec = {"blocks": to_dev(dblocks),
               "recs": torch.zeros(n_recs * 16, dtype=torch.uint8, device=dev),
               "seq": torch.zeros(seq_out_bytes, dtype=torch.uint8, device=dev),
               "res": torch.full((my_blocks * 16,), 0xff, dtype=torch.uint8, device=dev),
               "vs": torch.zeros(max(my_blocks * pb.cap_var, 1), dtype=torch.int32, device=dev)}
        dec["db"] = gpu.DecDeviceBatch(d_packed.data_ptr(), packed_cap, dec["blocks"].data_ptr(), my_blocks,
                                       d_ref.data_ptr(), d_ref.numel(), dec["recs"].data_ptr(), n_recs,
                                       dec["seq"].data_ptr(), dec["seq"].numel(), dec["res"].data_ptr(),
                                       dec["vs"].data_ptr(), dec["vs"].numel(), caps)
        (enc.decode_long_device if long_fmt else enc.decode_device)(dec["db"], stream)
        torch.cuda.synchronize()
        dres = dec["res"].cpu().numpy().view(host.RESULT_DTYPE)
        if (dres["status"] != 0).any():
            raise SystemExit("decode failed: %r" % (dres[dres["status"] != 0][:1],))
        r0 = int(blocks["rec_base"][b0]) if my_blocks else 0
        if long_fmt:                                          # fixed-length synthetic reads: every block's bases are contiguous on both sides
            got = dec["seq"].cpu().numpy()
            for k in range(my_blocks):
                o, nbk, s0 = int(dblocks["seq_base"][k]), int(dblocks["reserved"][k][0]), int(blocks["seq_base"][b0 + k])
                if not (got[o:o + nbk] == pb.seq[s0:s0 + nbk]).all():
                    raise SystemExit("decode does not reproduce the packed bases (block %d)" % (b0 + k))
            del got
        else:
            got = dec["seq"][:n_recs * stride].view(n_recs, stride)[:, :args.read_len].cpu().numpy()
            if not (got == pb.seq[r0 * args.read_len:(r0 + n_recs) * args.read_len].reshape(n_recs, args.read_len)).all():
                raise SystemExit("decode does not reproduce the packed bases")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(args.steps):
        step(kev[i])
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t_start
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_bases, n_recs], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bases, total_recs = int(tot[0].item()), int(tot[1].item())
    else:
        total_bases, total_recs = n_bases, n_recs

    kernel_ms = [a.elapsed_time(b) for a, b in kev]          # read after the timed region (everything has completed)
    k_ms = float(np.mean(kernel_ms))

    if args.mode == "encode":
        torch.cuda.synchronize()
        now_bytes = int(d_offs[my_blocks].item()) if my_blocks else 0
        if now_bytes != payload_bytes or not torch.equal(d_packed[:payload_bytes], first_payload):
            raise SystemExit("bench.py: the timed launches did not reproduce the first launch's bitstreams")

    rccl = None
    if multi:
        # the exchange proves itself: checksum of this rank's payload taken on its device vs what rank 0 holds after the
        # LAST timed gather (decode mode makes no gather inside the steps: one is made here)
        if args.mode == "decode":
            dist.gather(d_packed[:gather_cap].to(cdev), gather_list, dst=0)
        table = R.verify_exchange(gather_list, payload_bytes, first_sum, [int(k_ms * 1000), int(t_local * 1e6 / args.steps), n_recs])
        devs = R.device_table()
        if rank == 0:
            for row, d in zip(table, devs):
                ex = row.pop("extra")
                row.update({"device_index": d["device_index"], "device": d["device"], "arch": d["arch"], "pci_bus_id": d["pci_bus_id"],
                            "kernel_ms": ex[0] / 1000.0, "step_ms_local": ex[1] / 1000.0, "reads": ex[2]})
            rccl = {"backend": args.backend, "is_rccl": args.backend == "nccl", "world_size": dist.get_world_size(),
                    "collective": "gather of each rank's compacted bitstreams (padded to %d bytes) to rank 0, once per step" % gather_cap,
                    "gathered_bytes_per_step": int(sum(r["payload_bytes"] for r in table)),
                    "padded_bytes_per_step": int(gather_cap) * world,
                    "checksums_match": True, "ranks": table}

    ms_per_step = elapsed * 1e3 / args.steps
    value = total_bases * args.steps / elapsed / 1e6
    n_tok_mine = int(blocks["n_tok"][b0:b1].sum())
    # per launch, this rank: read + reference bases + 16 B record + ~2 B out; long reads also carry their CIGAR tokens
    alg_bytes = (2 * args.read_len + 18) * n_recs + (4 * n_tok_mine if long_fmt else 0)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    traffic, traffic_src, issue = None, None, None
    default_shape = args.reads == (1_000_000 if long_fmt else 10_000_000) and args.read_len == (10_000 if long_fmt else 150) \
        and args.block_reads == (64 if long_fmt else 4096) and not strong
    if default_shape:
        traffic, traffic_src, issue = issue_picture(ROOT, ("long_" if long_fmt else "") + args.mode + "_pmc", k_ms, n_recs)

    cpu = None
    whole_file_bits = None
    verified = None
    if rank == 0 and world == 1 and not multi and not args.no_cpu_baseline:
        from oracle import oracle                            # the checker, also timed as the reported baseline
        # parse excluded + verification: the CPU port codes the VERY blocks the GPU coded (same packed batch), bounded by
        # --cpu-sample-reads; its bytes must equal the GPU's compacted payload over those blocks
        cum = np.cumsum(blocks["n_reads"].astype(np.int64))
        nb_s = max(1, int(np.searchsorted(cum, min(args.cpu_sample_reads, int(cum[-1])), side="right")))
        nb_s = min(nb_s, pb.n_blocks)
        t1 = time.perf_counter()
        cflat, coffs, cres = oracle.cpu_encode_blocks(pb, blocks=None if nb_s == pb.n_blocks else range(nb_s), return_flat=True,
                                                      long_reads=long_fmt)
        t_blk = time.perf_counter() - t1
        s_recs, s_bases = int(cum[nb_s - 1]), int(pb.info["n_bases"][:nb_s].sum())
        gbytes = int(offs_first[nb_s])
        gflat = first_payload[:gbytes].cpu().numpy()
        verified = bool(int(coffs[-1]) == gbytes and np.array_equal(coffs, offs_first[:nb_s + 1]) and np.array_equal(cflat, gflat)
                        and (cres["n_symbols"] == res["n_symbols"][:nb_s]).all())
        if not verified:
            bad = [b for b in range(nb_s) if int(coffs[b + 1] - coffs[b]) != int(offs_first[b + 1] - offs_first[b])
                   or not np.array_equal(cflat[int(coffs[b]):int(coffs[b + 1])], gflat[int(offs_first[b]):int(offs_first[b + 1])])]
            raise SystemExit("bench.py: the GPU's bitstreams differ from the CPU port's on the same blocks (first: %r)" % (bad[:4],))
        port = {"value": round(s_bases / t_blk / 1e6, 2), "unit": "Mbases/s", "cores": 1,
                "sample": "the first %d blocks (%d reads) the GPU coded, as packed blocks (no text parsing), block by block on one "
                          "core, %.1f s; bytes compared with the GPU's" % (nb_s, s_recs, t_blk),
                "payload_bytes": int(coffs[-1])}
        if long_fmt:
            cpu = dict(port, kind="port", host_cpus=os.cpu_count())
            cpu["sample"] += "; oracle/cbc_long.c (the format has no reference implementation)"
        else:
            sample = min(args.cpu_sample_reads, args.reads)
            spb, sam, fa = host.synth(0xCBC00002, args.contig_len, sample, args.read_len, 0.003, 0.02, b"chr1",
                                      want_text=True, block_reads=args.block_reads)
            t1 = time.perf_counter()
            data, st = oracle.encode(sam, fa, return_stats=True)
            t_cpu = time.perf_counter() - t1
            whole_file_bits = round(len(data) * 8.0 / max(st.n_records, 1), 3)
            cpu = {"value": round(st.n_bases / t_cpu / 1e6, 2), "unit": "Mbases/s", "cores": 1, "kind": "port",
                   "sample": "%d reads x %d bp of the same synthetic workload%s, one whole-file stream, SAM text "
                             "parsing and FASTA load included, %.1f s of CPU" % (
                                 sample, args.read_len, " (all of it)" if sample == args.reads else "", t_cpu),
                   "host_cpus": os.cpu_count(), "parse_excluded": port}
            del sam, fa, data
            spb.close()

    e2e = None
    if rank == 0 and world == 1 and not multi and not args.no_e2e and not long_fmt and args.mode == "encode":
        e2e = e2e_legs(args, enc, pb, first_payload, offs_first)

    cfg4 = None
    if multi and not args.no_cfg4 and not long_fmt and args.mode == "encode":
        del d_recs, d_seq, d_tok, d_out, d_packed, first_payload
        torch.cuda.empty_cache()
        cfg4 = cfg4_pass(args, R, args.scale if args.scale is not None else 0.1)

    if rank == 0:
        n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        out = {
            "metric": ("Mbases/s %s on synthetic 10 kb long-read SAM (format extension v3: no reference parity exists)" % (
                           "encoded (== CPU statement of the format)" if args.mode == "encode" else "decoded (round trip verified)")) if long_fmt else
                      "Mbases/s encoded (bit-exact) on synthetic 150 bp SAM" if args.mode == "encode"
                      else "Mbases/s decoded (round trip verified) on synthetic 150 bp SAM",
            "value": round(value, 2), "unit": "Mbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "verified_vs_cpu_port": verified,
            "config": {"workload": "%s: synthetic %d bp SAM, %d reads %s vs a %d bp uniform-ACGT "
                                   "contig, block-parallel %s" % (
                                       "cfg5 (long reads, 5 %% edits)" if long_fmt else {10_000_000: "cfg2", 49_791_284: "cfg3 (30x)"}.get(args.reads, "custom"),
                                       args.read_len, args.reads, "in total" if strong else "per GPU", args.contig_len, args.mode),
                       "reads_rank0": n_recs, "blocks_rank0": my_blocks, "block_reads": args.block_reads,
                       "lds_bytes_per_block": lds_bytes, "payload_bytes_rank0": payload_bytes,
                       "bits_per_read": round(payload_bytes * 8.0 / max(n_recs, 1), 3),
                       "whole_file_bits_per_read": whole_file_bits,
                       "symbols_per_read": round(n_symbols / max(n_recs, 1), 3),
                       "parallelism": "one GPU, no collective" if not multi else
                                      "blocks sharded over %d GPU(s) (one process each), gather of bitstreams to rank 0 (%s)" % (
                                          world, "RCCL" if args.backend == "nccl" else "gloo rehearsal"),
                       "gathered_equals_single_gpu": strong_check,
                       "host_pack_seconds": round(t_gen, 1),
                       "note": "the one output that is bit-identical to the reference's WHOLE file is the compat stream (cbc -c --compat): "
                               "one serial arithmetic stream, one wavefront, ~38 Mbases/s (0.17x of one CPU core); the figure here is "
                               "block mode, each block bit-identical to the reference run on that block alone, at 16.6 vs 12.9 bit/read"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("cbc_long_%s_kernel" % args.mode) if long_fmt else
                                   "cbc_%s_blocks_kernel%s" % (args.mode, "_w6" if args.mode == "encode" and my_blocks > 10 * n_cus else ""),
                         "kernel_ms": round(k_ms, 3), "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_source_sha": kernel_source_sha(("long_" if long_fmt else "") + args.mode), "issue": issue},
            "cpu_baseline": cpu,
        }
        if e2e is not None:
            out["e2e"] = e2e
        if rccl is not None:
            out["rccl"] = rccl
        if cfg4 is not None:
            out["cfg4"] = cfg4
        print(json.dumps(out), flush=True)


def e2e_legs(args, enc, pb, first_payload, offs_first):
    """SURVEY 8d's timed regions (ii) and (iii), measured in this run outside the headline region, N = 1 only.
    (ii) the host-buffer entry point of the C ABI on the very batch the headline coded: H2D of the packed blocks (bases as
    2-bit codes), encode, compaction, D2H of the bitstreams -- second call of two, so the context's persistent device
    buffers and pinned staging exist (the first call's time is reported too); output bytes compared with the headline's.
    (iii) the `cbc` CLI on SAM text of the same generator (--e2e-reads records): wall time of `cbc -c` incl. text parsing."""
    import numpy as np
    from cbc_amd import host
    out = {}
    try:
        enc.upload_reference(pb.ref)
        codes, runs = host.pack_2bit(pb.seq)
        want = first_payload.cpu().numpy()

        def call(two_bit):
            t1 = time.perf_counter()
            if two_bit:
                _, res, offs, flat = enc.encode_blocks_2bit(pb, codes, runs, want_payload_list=False)
            else:
                _, res, offs, flat = enc.encode_blocks(pb, want_payload_list=False)
            dt = time.perf_counter() - t1
            if int(offs[-1]) != int(offs_first[-1]) or not np.array_equal(flat, want):
                raise SystemExit("bench.py: the host-buffer entry point's bitstreams differ from the device path's")
            return dt, int(offs[-1])

        cold, nbytes = call(True)                              # first call: the context's device arenas are allocated, host memory pageable
        pageable = min(call(True)[0] for _ in range(2))
        t1 = time.perf_counter()
        regs = [pb.recs, codes, pb.tok, pb.seq]
        for a in regs:
            enc.host_register(a)
        t_reg = time.perf_counter() - t1
        pinned = min(call(True)[0] for _ in range(3))
        stages = enc.last_e2e()
        call(False)
        pinned_1b = min(call(False)[0] for _ in range(2))
        stages_1b = enc.last_e2e()
        for a in regs:
            enc.host_unregister(a)
        out.update({"device_gbases_s": round(pb.n_bases / pinned / 1e9, 2), "device_ms": round(pinned * 1e3, 2),
                    "bytes_h2d_per_read": round(stages["h2d_bytes"] / max(pb.n_recs, 1), 1),
                    "bytes_d2h_per_read": round(stages["d2h_bytes"] / max(pb.n_recs, 1), 2),
                    "chunks": stages["n_chunks"],
                    "device_gbases_s_pageable_host_memory": round(pb.n_bases / pageable / 1e9, 2),
                    "device_gbases_s_first_call": round(pb.n_bases / cold / 1e9, 2),
                    "device_gbases_s_1_byte_per_base": round(pb.n_bases / pinned_1b / 1e9, 2),
                    "bytes_h2d_per_read_1_byte_per_base": round(stages_1b["h2d_bytes"] / max(pb.n_recs, 1), 1),
                    "host_register_ms": round(t_reg * 1e3, 1),
                    "device_what": "cbc_gpu_encode_blocks_2bit on host buffers: chunked H2D (bases as 2-bit codes) overlapped with the encode "
                                   "launches + compaction + D2H; best of 3 calls with the caller's arrays page-locked (cbc_gpu_host_register, "
                                   "its one-off cost is host_register_ms) and the context's device buffers in place; bytes == the device path's"})
    except AttributeError as e:                                # an older library without the entry points
        out["device_error"] = str(e)
    exe = os.path.join(ROOT, "cbc_amd", "csrc", "cbc")
    if os.path.exists(exe) and args.e2e_reads > 0:
        import tempfile
        n = min(args.e2e_reads, args.reads)
        clen = max(int(args.contig_len * (n / args.reads)), 10 * args.read_len)
        spb, sam, fa = host.synth(0xCBC00002, clen, n, args.read_len, 0.003, 0.02, b"chr1", want_text=True, block_reads=args.block_reads)
        nb = spb.n_bases
        spb.close()
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            open(os.path.join(td, "in.sam"), "wb").write(sam)
            open(os.path.join(td, "ref.fa"), "wb").write(fa)
            sam_bytes = len(sam)
            del sam, fa
            best, stages = None, None
            for _ in range(2):
                t1 = time.perf_counter()
                r = subprocess.run([exe, "-c", os.path.join(td, "in.sam"), os.path.join(td, "out.cbc"), os.path.join(td, "ref.fa"), "--verbose"],
                                   capture_output=True, text=True)
                dt = time.perf_counter() - t1
                if r.returncode != 0:
                    out["cli_error"] = (r.stderr or r.stdout)[-300:]
                    break
                if best is None or dt < best:
                    best, stages = dt, [l.strip() for l in (r.stdout + r.stderr).splitlines() if "stage" in l][:12]
            if best is not None:
                out.update({"cli_gbases_s": round(nb / best / 1e9, 3), "cli_seconds": round(best, 3), "cli_reads": n,
                            "cli_sam_bytes": sam_bytes, "cli_stages": stages,
                            "cli_what": "`cbc -c in.sam out.cbc ref.fa` wall time (process start to exit, files in /dev/shm), best of 2"})
    return out


# GRCh38 primary assembly, chr1..22, X, Y (bases) -- SURVEY.md 8d, config 4
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415]
GRCH38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]


def assign_largest_first(loads, n_parts):
    """cbc_assign_contigs' rule on plain numbers: largest remaining item to the least loaded part, ties to the lower index."""
    part, load = [0] * len(loads), [0] * n_parts
    for c in sorted(range(len(loads)), key=lambda i: (-loads[i], i)):
        q = min(range(n_parts), key=lambda k: (load[k], k))
        part[c] = q
        load[q] += loads[c]
    return part


def cfg4_workload(scale, read_len):
    """Contig lengths and read counts of BASELINE config 4 at `scale` of GRCh38 (30x coverage at every scale)."""
    lens = [max(int(l * scale), 4 * read_len + 1000) for l in GRCH38]
    reads = [max(int(30 * l / read_len), 1) for l in lens]
    return lens, reads


def cfg4_pass(args, R, scale):
    """BASELINE config 4: whole GRCh38-shaped genome at 30x, chromosome-sharded (strong scaling: the genome is fixed, the
    ranks split it).  Every rank generates and codes the contigs it is dealt (no collective on the data path); the one
    exchange is the gather of the bitstreams to rank 0, checksummed on both sides.  Returns the result dict on rank 0."""
    import numpy as np
    import torch
    from cbc_amd import gpu, host
    dist, rank, world, dev, cdev, enc, stream = R.dist, R.rank, R.world, R.dev, R.cdev, R.enc, R.stream
    multi = R.multi
    lens, reads = cfg4_workload(scale, args.read_len)
    part = assign_largest_first(reads, world)
    mine = [c for c in range(24) if part[c] == rank]
    L = gpu.lib()
    t0 = time.time()
    res_list, n_bases, n_recs, n_blocks = [], 0, 0, 0

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)

    for c in mine:
        pb = host.synth(0xCBC00004 + c, lens[c], reads[c], args.read_len, 0.003, 0.02, GRCH38_NAMES[c].encode(), block_reads=args.block_reads)
        blocks = pb.blocks.copy()
        scratch = int(L.cbc_gpu_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, pb.tok.ctypes.data))
        C = {"t": [to_dev(x) for x in (pb.recs, pb.seq, pb.tok, pb.names, blocks, pb.ref)], "nb": pb.n_blocks,
             "out": torch.empty(scratch, dtype=torch.uint8, device=dev), "res": torch.full((pb.n_blocks * 16,), 0xff, dtype=torch.uint8, device=dev),
             "offs": torch.zeros(pb.n_blocks + 1, dtype=torch.int64, device=dev), "cap": max(1 << 20, 8 * pb.n_recs)}
        C["packed"] = torch.empty(C["cap"], dtype=torch.uint8, device=dev)
        t = C["t"]
        C["db"] = gpu.DeviceBatch(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), pb.n_blocks,
                                  t[5].data_ptr(), t[5].numel(), C["out"].data_ptr(), scratch, C["res"].data_ptr(), t[1].numel(),
                                  max(pb.n_tok, 1), pb.n_recs, host.LdsCaps(pb.cap_pos, pb.cap_var))
        res_list.append(C); n_bases += pb.n_bases; n_recs += pb.n_recs; n_blocks += pb.n_blocks
        pb.close()
    t_gen = time.time() - t0

    # one launch per contig; the launches of a step go round-robin over four HIP streams so that small contigs overlap
    side = [torch.cuda.Stream(device=dev) for _ in range(4)]
    hs = [ctypes.c_void_p(x.cuda_stream) for x in side]

    for C in res_list:                                        # first launch: every block must finish
        enc.encode_device(C["db"], stream)
        enc.compact_device(C["out"].data_ptr(), C["t"][4].data_ptr(), C["res"].data_ptr(), C["nb"], C["offs"].data_ptr(), C["packed"].data_ptr(), C["cap"], stream)
    torch.cuda.synchronize()
    payload = 0
    for C in res_list:
        r = C["res"].cpu().numpy().view(host.RESULT_DTYPE)
        if (r["status"] != 0).any():
            raise SystemExit("a block failed: %r" % (r[r["status"] != 0][:1],))
        C["bytes"] = int(C["offs"][C["nb"]].item())
        C["at"] = payload
        payload += C["bytes"]
    # this rank's bitstreams (all its contigs, contig order) as ONE buffer: what the exchange step sends
    gather_cap, gather_list = max(payload, 1), None
    if multi:
        m = torch.tensor([payload], dtype=torch.int64, device=cdev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        gather_cap = max(int(m.item()), 1)
        if rank == 0:
            gather_list = [torch.empty(gather_cap, dtype=torch.uint8, device=cdev) for _ in range(world)]
    flat = torch.zeros(gather_cap, dtype=torch.uint8, device=dev)

    def flatten():
        for C in res_list:
            flat[C["at"]:C["at"] + C["bytes"]].copy_(C["packed"][:C["bytes"]])

    flatten()
    first_flat = flat[:payload].clone()
    first_sum = R.device_checksum(flat, payload)

    def step():
        for i, C in enumerate(res_list):
            q = hs[i % len(hs)]
            enc.encode_device(C["db"], q)
            enc.compact_device(C["out"].data_ptr(), C["t"][4].data_ptr(), C["res"].data_ptr(), C["nb"], C["offs"].data_ptr(),
                               C["packed"].data_ptr(), C["cap"], q)
        for x in side:
            x.synchronize()
        flatten()                                             # the per-step concatenation a real exchange needs
        if multi:
            dist.gather(flat.to(cdev), gather_list, dst=0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t_start
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if not torch.equal(flat[:payload], first_flat):
        raise SystemExit("bench.py (cfg4): the timed launches did not reproduce the first launch's bitstreams")
    tot = torch.tensor([float(n_bases), float(n_recs), float(n_blocks), float(payload), elapsed], dtype=torch.float64, device=cdev)
    table = None
    if multi:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[4].item())
        table = R.verify_exchange(gather_list, payload, first_sum, [int(t_local * 1e6 / args.steps), n_recs, len(mine)])
    total_bases, total_recs, total_blocks, total_payload = (int(tot[i].item()) for i in range(4))
    if rank != 0:
        return None
    alg = (2 * args.read_len + 18) * total_recs * args.steps / elapsed / 1e9
    out = {
        "metric": "Mbases/s encoded (bit-exact) on synthetic 150 bp SAM", "value": round(total_bases * args.steps / elapsed / 1e6, 2),
        "unit": "Mbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "cfg4: 24 contigs of GRCh38's primary lengths x %.3g at 30x coverage, %d bp reads, chromosome-sharded over %d GPU(s) "
                               "(whole contigs dealt largest first; every rank generates and codes its own contigs)" % (scale, args.read_len, world),
                   "scale": scale, "reads_total": total_recs, "blocks_total": total_blocks, "payload_bytes_total": total_payload,
                   "bits_per_read": round(total_payload * 8.0 / max(total_recs, 1), 3), "contigs_rank0": [GRCH38_NAMES[c] for c in mine],
                   "contig_to_rank": {GRCH38_NAMES[c]: part[c] for c in range(24)},
                   "reads_rank0": n_recs, "host_pack_seconds_rank0": round(t_gen, 1),
                   "parallelism": "one GPU, no collective" if not multi else
                                  "contigs sharded over %d GPU(s) (one process each), gather of bitstreams to rank 0 (%s)" % (
                                      world, "RCCL" if args.backend == "nccl" else "gloo rehearsal")},
        "roofline": {"bound": "hbm", "achieved": round(alg, 3), "peak": HBM_PEAK_GBS * world,
                     "unit": "GB/s", "frac": round(alg / (HBM_PEAK_GBS * world), 6),
                     "traffic": None, "kernel": "cbc_encode_blocks_kernel(_w6) over every contig", "note": "whole-step wall time, all launches of the step"},
        "cpu_baseline": None}
    if table is not None:
        for row in table:
            ex = row.pop("extra")
            row.update({"step_ms_local": ex[0] / 1000.0, "reads": ex[1], "contigs": ex[2]})
        out["rccl"] = {"backend": args.backend, "is_rccl": args.backend == "nccl", "world_size": dist.get_world_size(),
                       "gathered_bytes_per_step": total_payload, "padded_bytes_per_step": gather_cap * world,
                       "checksums_match": True, "ranks": table}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    R = Rank(args)
    if args.workload == "cfg4":
        out = cfg4_pass(args, R, args.scale if args.scale is not None else 0.05)
        if R.rank == 0:
            print(json.dumps(out), flush=True)
    else:
        run_rank(args, R)
    if R.multi:
        R.dist.destroy_process_group()


if __name__ == "__main__":
    main()
