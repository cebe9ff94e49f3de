"""The long-read format extension (stream version 3).

NO REFERENCE PARITY EXISTS for this format: the reference cannot code a read longer than 252 bases.  Its two checks
are (1) the HIP kernels == oracle/cbc_long.c, the independent CPU statement of the specification, byte for byte, and
(2) decode(encode(x)) == x.  CPU part: packer long mode, the kernel bodies on the wave emulation.  GPU part: -m gpu."""
import os
import subprocess

import numpy as np
import pytest

import blockref
from cbc_amd import gpu, host
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    (0xCBC00005, 2_000_000, 200, 10_000, 0.05),     # cfg5's shape in small: 10 kb reads, 5 % indel + substitution mix
    (7, 500_000, 400, 1_000, 0.02),
    (9, 300_000, 30, 30_000, 0.10),                 # beyond 16 kb; gaps and counts that need the escapes
    (11, 100_000, 150, 200, 0.0),                   # no edits at all
    (15, 400_000, 12, 30_000, 0.45),                # ~13 k edits per read: more than the encoder's edit buffer holds (the read is walked twice)
    (13, 1_500_000, 300, 4_000, 0.002),             # sparse edits: most gaps beyond the 64 symbols held in LDS, many with the two gx bytes
    (17, 300_000, 20, 12_000, 0.45),                # ~5 k edits per read: more than one half of the edit buffer, less than both
]


def _break_a_cigar(pb, blk, read):
    """The CIGAR of one read no longer consumes the read (its first run is one base longer): what the walk must refuse."""
    b = pb.blocks[blk]
    t = int(b["tok_base"]) + int(pb.recs[int(b["rec_base"]) + read]["tok_off"]) + 2
    pb.tok[t] += 1 << 4


def _reads(sam):
    return b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines() if ln and not ln.startswith(b"@"))


def test_packer_long_mode(built):
    pb, sam, fa = host.synth_long(3, 600_000, 150, 5_000, 0.05, want_text=True, threads=3)
    assert pb.long_reads and pb.max_read_len == 5000 and pb.n_recs == 150
    # SAM lines of ~10 kB go through the text path (the reference's loader stops at 1023 bytes) and give the same arrays
    assert max(len(l) for l in sam.splitlines()) > 10_000
    pt = host.pack_sam(sam, fa, long_reads=True)
    for k in ("recs", "seq", "tok", "blocks", "info"):
        assert getattr(pt, k).tobytes() == getattr(pb, k).tobytes(), k
    # block cuts: 64 reads, or fewer when a block would pass 1 Mbase; POS rebased per block
    assert int(pb.blocks["n_reads"].max()) <= 64 and (pb.info["n_bases"] <= (1 << 20)).all()
    assert (pb.recs["pos"][pb.blocks["rec_base"].astype(np.int64)] == 1).all()
    # one generator thread or many: the same data
    p1, _, _ = host.synth_long(3, 600_000, 150, 5_000, 0.05, want_text=True, threads=1)
    assert p1.seq.tobytes() == pb.seq.tobytes() and p1.tok.tobytes() == pb.tok.tobytes()
    # refused: characters outside ACGTN (the base models code classes), CIGAR / SEQ length mismatch; block mode refuses 5 kb reads
    lines = sam.splitlines(keepends=True)
    f = lines[0].split(b"\t"); f[9] = f[9][:100] + b"R" + f[9][101:]
    with pytest.raises(host.CbcInputError, match="ACGTN"):
        host.pack_sam(b"\t".join(f) + b"".join(lines[1:]), fa, long_reads=True)
    f = lines[0].split(b"\t"); f[9] = f[9][:-1]
    with pytest.raises(host.CbcInputError, match="lengths differ"):
        host.pack_sam(b"\t".join(f) + b"".join(lines[1:]), fa, long_reads=True)
    with pytest.raises(host.CbcInputError):
        host.pack_sam(sam, fa)


@pytest.mark.parametrize("args", CASES)
def test_long_bodies_equal_the_cpu_statement_and_round_trip(built, args):
    pb, sam, fa = host.synth_long(*args, want_text=True, threads=4)
    ep, eres = blockref.emu_long_encode(pb)
    cp, cres = oracle.cpu_encode_blocks(pb, return_payloads=True, long_reads=True)
    assert (eres["status"] == 0).all() and ep == cp and (eres["n_symbols"] == cres["n_symbols"]).all()
    plan = host.UnpackPlan(blockref.container_from_payloads(pb, ep), fa)
    assert plan.long_reads and plan.n_recs == pb.n_recs
    recs, seq, dres = blockref.emu_long_decode(plan)
    assert (dres["status"] == 0).all() and (dres["n_symbols"] == eres["n_symbols"]).all()
    assert plan.text(recs, seq) == _reads(sam)
    assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all() and (recs["rlen"] == pb.recs["rlen"]).all()
    # the CPU statement's own decoder agrees
    b = pb.n_blocks - 1
    crecs, cseq = oracle.cpu_long_decode_block(ep[b], pb.ref, int(pb.blocks[b]["ref_off"]), int(pb.blocks[b]["n_reads"]) + 1,
                                               int(pb.info[b]["n_bases"]) + 16)
    nb = int(pb.info[b]["n_bases"]); s0 = int(pb.blocks[b]["seq_base"])
    assert len(crecs) == int(pb.blocks[b]["n_reads"]) and (cseq[:nb] == pb.seq[s0:s0 + nb]).all()


def test_long_soft_clips_and_out_full(built):
    """Soft clips are coded as insertions; an output area that is too small is reported as OUT_FULL, not overrun."""
    pb, sam, fa = host.synth_long(5, 200_000, 40, 2_000, 0.03, want_text=True, threads=1)
    lines = sam.splitlines(keepends=True)
    out = []
    for i, ln in enumerate(lines):
        f = ln.split(b"\t")
        if i % 3 == 0:                                            # turn the first 7 and last 5 aligned bases into clips
            f[5] = b"7S" + f[5] + b"5S"; f[9] = b"ACGTACG" + f[9] + b"TTTTT"; f[10] = f[10][:-1] + b"I" * 12 + b"\n"
        out.append(b"\t".join(f))
    sam2 = b"".join(out)
    pc = host.pack_sam(sam2, fa, long_reads=True)
    ep, eres = blockref.emu_long_encode(pc)
    cp, _ = oracle.cpu_encode_blocks(pc, return_payloads=True, long_reads=True)
    assert (eres["status"] == 0).all() and ep == cp
    plan = host.UnpackPlan(blockref.container_from_payloads(pc, ep), fa)
    recs, seq, dres = blockref.emu_long_decode(plan)
    assert plan.text(recs, seq) == _reads(sam2)
    big, _, _ = host.synth_long(6, 2_000_000, 64, 10_000, 0.05, want_text=True, threads=2)      # ~35 kB of payload in one block
    _, small = blockref.emu_long_encode(big, out_cap_per_base=0.001)
    assert (small["status"] == 1).all() and (small["nbytes"] == 0).all()


def test_long_walk_failure_is_the_blocks_status(built):
    pb, sam, fa = host.synth_long(23, 1_500_000, 150, 8_000, 0.05, want_text=True, threads=2)
    assert pb.n_blocks >= 2
    _break_a_cigar(pb, 1, 5)
    _, res = blockref.emu_long_encode(pb)
    assert res["status"][1] != 0 and res["nbytes"][1] == 0 and res["fail_read"][1] == 5
    assert (np.delete(res["status"], 1) == 0).all()


# ----------------------------------------------------------------------------------------- on the GPU
@pytest.fixture(scope="module")
def enc():
    e = gpu.Encoder(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("args", CASES + [(0xCBC00005, 20_000_000, 4_000, 10_000, 0.05)])
def test_gpu_long_encode_decode(enc, built, args):
    """HIP kernels == the CPU statement per block; GPU decode of the container == the reads."""
    pb, sam, fa = host.synth_long(*args, want_text=True)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_long_blocks(pb)
    assert (res["status"] == 0).all(), res[res["status"] != 0][:4]
    cp, cres = oracle.cpu_encode_blocks(pb, return_payloads=True, long_reads=True)
    assert payloads == cp and (res["n_symbols"] == cres["n_symbols"]).all()
    plan = host.UnpackPlan(pb.container(flat, offs), fa)
    enc.upload_reference(plan.ref)
    recs, seq, dres = enc.decode_long_blocks(plan)
    assert (dres["status"] == 0).all() and (dres["n_symbols"] == res["n_symbols"]).all()
    assert plan.text(recs, seq) == _reads(sam)
    assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all()


@pytest.mark.gpu
def test_gpu_long_walk_failure_reaches_the_result_and_every_wavefront_leaves(enc, built):
    """A failure of the walker wavefront travels walker -> model -> coder -> results; the other blocks are coded as ever."""
    pb, sam, fa = host.synth_long(23, 1_500_000, 150, 8_000, 0.05, want_text=True, threads=2)
    ref_payloads, _ = oracle.cpu_encode_blocks(pb, return_payloads=True, long_reads=True)
    _break_a_cigar(pb, 1, 5)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_long_blocks(pb)
    assert res["status"][1] != 0 and res["nbytes"][1] == 0 and res["fail_read"][1] == 5
    assert (np.delete(res["status"], 1) == 0).all()
    assert [p for k, p in enumerate(payloads) if k != 1] == [p for k, p in enumerate(ref_payloads) if k != 1]


@pytest.mark.gpu
def test_cli_long_round_trip(built, tmp_path):
    exe = os.path.join(ROOT, "cbc_amd", "csrc", "cbc")
    pb, sam, fa = host.synth_long(21, 3_000_000, 500, 8_000, 0.05, want_text=True)
    (tmp_path / "in.sam").write_bytes(sam); (tmp_path / "ref.fa").write_bytes(fa)
    r = subprocess.run([exe, "-c", "--long", str(tmp_path / "in.sam"), str(tmp_path / "out.cbc"), str(tmp_path / "ref.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "-d", str(tmp_path / "out.cbc"), str(tmp_path / "reads.txt"), str(tmp_path / "ref.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "reads.txt").read_bytes() == _reads(sam)
    # without --long the reference's limits apply and the input is refused with a message
    r = subprocess.run([exe, "-c", str(tmp_path / "in.sam"), str(tmp_path / "x.cbc"), str(tmp_path / "ref.fa")], capture_output=True, text=True)
    assert r.returncode != 0 and "long-read format" in r.stderr
