"""Whole-file stream ("compat" mode) speed: one arithmetic stream = one wavefront.  python tools/stream_speed.py [reads]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from cbc_amd import host, gpu
from oracle import oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pb0, sam, fa = host.synth(0xCBC00002, int(N * 24.9), N, 150, want_text=True)
pb = host.pack_sam(sam, fa, whole_file=True)
enc = gpu.Encoder(0); enc.upload_reference(pb.ref)
t = time.time(); stream, sr = enc.encode_stream(pb); te = time.time() - t
kms = enc.last_kernel_ms()
t = time.time(); exp = oracle.encode(sam, fa); tc = time.time() - t
t = time.time(); recs, bases, dr = enc.decode_stream(stream, pb.contigs, rec_cap=N + 16); td = time.time() - t
print("%d reads as ONE stream: GPU encode %.2f s (kernel %.1f ms = %.1f Mbases/s, %.2f us per read), %d bytes == oracle: %s; oracle on one host core %.2f s (%.1f Mbases/s); "
      "GPU decode %.2f s (kernel %.1f ms), %d records" % (N, te, kms, pb.n_bases / kms / 1e3, kms * 1e3 / N, len(stream), stream == exp, tc, pb.n_bases / tc / 1e6, td, enc.last_kernel_ms(), len(recs)))
