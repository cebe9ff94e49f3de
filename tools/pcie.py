"""PCIe-inclusive rates of the host-buffer entry points (never bench.py's `value`): bytes per read over PCIe and wall time of
cbc_gpu_encode_blocks / cbc_gpu_decode_blocks with bases at 1 byte each vs in 2-bit transport form.  python tools/pcie.py [reads]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from cbc_amd import host, gpu
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pb = host.synth(0xCBC00002, 248956422, N, 150, block_reads=4096)
enc = gpu.Encoder(0)
t = time.time(); enc.upload_reference(pb.ref); t_ref = time.time() - t
t = time.time(); rc, rr = host.pack_2bit(pb.ref); t_pack_ref = time.time() - t
t = time.time(); enc.upload_reference_2bit(rc, rr, len(pb.ref)); t_ref2 = time.time() - t
print("reference %d bases: upload %.3f s at 1 B/base; host 2-bit pack %.3f s + upload %.3f s (%.2f B/base over PCIe)" % (
    len(pb.ref), t_ref, t_pack_ref, t_ref2, (rc.nbytes + rr.nbytes) / len(pb.ref)))
for rep in range(2):
    t = time.time(); p1, r1, offs, flat = enc.encode_blocks(pb); t1 = time.time() - t
t = time.time(); sc, sr = host.pack_2bit(pb.seq); t_pack = time.time() - t
for rep in range(2):
    t = time.time(); p2, r2, _, _ = enc.encode_blocks_2bit(pb, sc, sr); t2 = time.time() - t
assert p1 == p2
h2d_1 = pb.recs.nbytes + pb.seq.nbytes + pb.tok.nbytes + pb.blocks.nbytes
h2d_2 = pb.recs.nbytes + sc.nbytes + sr.nbytes + pb.tok.nbytes + pb.blocks.nbytes
print("encode %d reads: 1 B/base %.3f s (%.1f Gbases/s, H2D %.0f B/read); 2-bit %.3f s (%.1f Gbases/s, H2D %.0f B/read; host pack of the bases %.3f s on %d CPUs, not included)" % (
    N, t1, pb.n_bases / t1 / 1e9, h2d_1 / N, t2, pb.n_bases / t2 / 1e9, h2d_2 / N, t_pack, os.cpu_count()))
blob = pb.container(flat, offs)
body = pb.ref[:248956422]
rows = np.concatenate([body[:len(body) // 60 * 60].reshape(-1, 60), np.full((len(body) // 60, 1), 10, dtype=np.uint8)], axis=1)
fa = b">chr1\n" + rows.tobytes() + body[len(body) // 60 * 60:].tobytes() + b"\n"
plan = host.UnpackPlan(blob, fa)
for rep in range(2):
    t = time.time(); recs, seq, dres = enc.decode_blocks(plan); t3 = time.time() - t
for rep in range(2):
    t = time.time(); recs2, bases2, dres2, pcie = enc.decode_blocks_2bit(plan); t4 = time.time() - t
print("decode: 1 B/base %.3f s (D2H %.0f B/read); 2-bit rows %.3f s incl. the host-side rebuild in numpy (D2H %.0f B/read)" % (
    t3, (recs.nbytes + plan.n_recs * plan.seq_stride) / N, t4, (recs2.nbytes + pcie) / N))
