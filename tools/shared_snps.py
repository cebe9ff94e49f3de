"""Encode / decode speed on data whose SNPs are SHARED variants (every read that covers a variant site carries it),
next to the same coverage with per-read random substitutions.  The bench workload has only the latter."""
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import synth
from cbc_amd import host, gpu

def make(seed, clen, nreads, L, site_every, err):
    rng = np.random.default_rng(seed)
    contig = synth.make_contig(rng, clen)
    alt = contig.copy()
    sites = np.sort(rng.choice(clen, size=max(clen // site_every, 1), replace=False)) if site_every else np.zeros(0, dtype=np.int64)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for s in sites:
        alt[s] = acgt[(int(np.where(acgt == contig[s])[0][0]) + 1 + int(rng.integers(0, 3))) % 4]
    starts = np.sort(rng.integers(0, clen - L - 8, size=nreads))
    out = []
    for i, s in enumerate(starts):
        s = int(s)
        seq = alt[s:s + L].copy()
        if err > 0:
            ne = rng.binomial(L, err)
            for q in rng.choice(L, size=ne, replace=False) if ne else []:
                seq[q] = acgt[(int(np.where(acgt == seq[q])[0][0]) + 1 + int(rng.integers(0, 3))) % 4]
        md, nm = synth._md_and_nm(contig, s, [("M", L)], seq)
        out.append(b"r%d\t%d\tc\t%d\t60\t%dM\t*\t0\t0\t%s\t%s\tMD:Z:%s\tNM:i:%d\n" % (i, 16 * int(rng.integers(0, 2)), s + 1, L, seq.tobytes(), b"I" * L, md.encode(), nm))
    return b"".join(out), synth.fasta_text([("c", contig)])

enc = gpu.Encoder(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
for name, site_every, err in (("random errors 0.3%", 0, 0.003), ("shared variants 1/300 bp + errors 0.1%", 300, 0.001), ("shared variants 1/100 bp", 100, 0.0)):
    t = time.time(); sam, fa = make(5, n * 5, n, 150, site_every, err); tg = time.time() - t
    pb = host.pack_sam(sam, fa, block_reads=4096)
    enc.upload_reference(pb.ref)
    ts = []
    for _ in range(3):
        payloads, res, offs, flat = enc.encode_blocks(pb); ts.append(enc.last_kernel_ms())
    assert (res["status"] == 0).all()
    blob = pb.container(flat, offs)
    plan = host.UnpackPlan(blob, fa); enc.upload_reference(plan.ref)
    td = []
    for _ in range(3):
        recs, seq, dres = enc.decode_blocks(plan); td.append(enc.last_kernel_ms())
    assert (dres["status"] == 0).all()
    text = plan.text(recs, seq)
    want = b"".join(l.split(b"\t")[9] + b"\n" for l in sam.split(b"\n") if l)
    print("%-42s reads %d blocks %d  encode %.2f ms  decode %.2f ms  %.1f bit/read  symbols/read %.2f  round trip %s  (gen %.0fs)" % (
        name, pb.n_recs, pb.n_blocks, min(ts), min(td), 8.0 * len(flat) / pb.n_recs, float(res["n_symbols"].sum()) / pb.n_recs, text == want, tg), flush=True)
