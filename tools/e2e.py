"""End-to-end text path on the GPU box: threaded packer scaling + the cbc CLI with stage times."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, '.')
from cbc_amd import host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
t = time.time(); pb, sam, fa = host.synth(7, 200_000_000, n, want_text=True); print("synth+text %.2fs, sam %.0f MB" % (time.time() - t, len(sam) / 1e6), flush=True)
def arrays(p): return [np.asarray(a).tobytes() for a in (p.recs, p.seq, p.tok, p.blocks)]
ref = arrays(pb); del pb
os.environ["CBC_PACK_TIMES"] = "1"
for th in (1, 4, 8, 16, 16):
    t = time.time(); q = host.pack_sam(sam, fa, threads=th); dt = time.time() - t
    print("%2d threads %.3fs  %.0f MB/s  %.1f Mbases/s  same=%s" % (th, dt, len(sam) / dt / 1e6, n * 150 / dt / 1e6, arrays(q) == ref), flush=True)
    del q
d = "/tmp/cbc_e2e"; os.makedirs(d, exist_ok=True)
open(d + "/a.sam", "wb").write(sam); open(d + "/a.fa", "wb").write(fa)
exe = os.path.join("cbc_amd", "csrc", "cbc")
for rep in range(2):
    t = time.time(); r = subprocess.run([exe, "-c", d + "/a.sam", d + "/a.cbc", d + "/a.fa", "--verbose"], capture_output=True, text=True); dt = time.time() - t
    print(r.stdout.strip(), r.stderr.strip()); print("cbc -c wall %.2fs -> %.1f Mbases/s end to end" % (dt, n * 150 / dt / 1e6), flush=True)
for rep in range(2):       # the same with the text tokenised on the device (cbc_gpu_tokenise_sam): identical container
    t = time.time(); r = subprocess.run([exe, "-c", d + "/a.sam", d + "/b.cbc", d + "/a.fa", "--verbose", "--device-parse"], capture_output=True, text=True); dt = time.time() - t
    print(r.stdout.strip(), r.stderr.strip()); print("cbc -c --device-parse wall %.2fs -> %.1f Mbases/s end to end; same container: %s" % (
        dt, n * 150 / dt / 1e6, open(d + "/a.cbc", "rb").read() == open(d + "/b.cbc", "rb").read()), flush=True)
# tokeniser alone, in process (text already in memory): device vs the threaded host packer
from cbc_amd import gpu
enc = gpu.Encoder(0)
for rep in range(3):
    t = time.time(); pd, tr = enc.tokenise_sam(sam, fa); dt = time.time() - t
    print("device tokeniser + host block cutting %.3fs  %.0f MB/s of SAM text  %.1f Mbases/s (H2D of the text included)" % (dt, len(sam) / dt / 1e6, n * 150 / dt / 1e6), flush=True)
    enc.tokenise_free(tr); del pd
t = time.time(); r = subprocess.run([exe, "-d", d + "/a.cbc", d + "/a.txt", d + "/a.fa"], capture_output=True, text=True); dt = time.time() - t
print(r.stdout.strip(), r.stderr.strip()); print("cbc -d wall %.2fs" % dt)
want = b"".join(l.split(b"\t")[9] + b"\n" for l in sam.split(b"\n") if l)
print("round trip text identical:", open(d + "/a.txt", "rb").read() == want)
