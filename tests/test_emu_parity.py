"""The kernel BODY (cbc_amd/csrc/cbc_encode_body.h) single-stepped on the CPU lock-step wave
emulation, against the oracle.  This is not the GPU parity test (tests/test_gpu_parity.py is): it
checks the body's logic and indexing without a GPU, so a kernel is never launched on hardware with
an out-of-range access the CPU could have caught."""
import glob
import json
import os

import numpy as np
import pytest

import blockref
import synth
from cbc_amd import host
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(pb, sam):
    payloads, res = blockref.emu_encode(pb)
    assert (res["status"] == 0).all(), res[res["status"] != 0]
    lines = blockref.mapped_sam_lines(sam)
    assert len(lines) == pb.n_recs
    for b in range(pb.n_blocks):
        bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
        exp, st = oracle.encode(bsam, bfa, return_stats=True)
        assert payloads[b] == exp, "block %d" % b
        assert int(res[b]["n_symbols"]) == st.n_symbols
    return payloads, res


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_L*.json"))))
def test_golden_block_payloads(built, path):
    g = json.load(open(path))
    pb = host.pack_sam(g["sam"].encode(), g["fasta"].encode(), block_reads=g["block_reads"])
    payloads, res = blockref.emu_encode(pb)
    assert (res["status"] == 0).all()
    assert [p.hex() for p in payloads] == g["block_payload_hex"]


@pytest.mark.parametrize("kw,L,br", [
    (dict(sub_rate=0.0, indel_frac=0.0), 100, 1000),
    (dict(), 150, 1024),
    (dict(sub_rate=0.01, indel_frac=0.3), 150, 700),
    (dict(sub_rate=0.02, indel_frac=0.5, trailing_s_frac=0.2, dup_pos_frac=0.1), 100, 512),
    (dict(sub_rate=0.05, indel_frac=0.0), 252, 300),
    (dict(flags=(0, 16, 83, 99, 147, 163, 1024 + 99, 2048)), 150, 2048),
    (dict(sub_rate=0.01, indel_frac=0.1), 33, 4096),
])
def test_blocks_match_oracle(built, kw, L, br):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [4000, 1500], L, **kw)
    pb = host.pack_sam(sam, fa, block_reads=br)
    _check(pb, sam)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kw,L,br", [
    (dict(), 150, 1024),
    (dict(sub_rate=0.02, indel_frac=0.5, trailing_s_frac=0.2, dup_pos_frac=0.1), 100, 512),
    (dict(sub_rate=0.45, indel_frac=0.3), 150, 256),              # dozens of edits per read: the queue drains inside a record
    (dict(sub_rate=0.003, indel_frac=0.02), 100, 1000),
])
def test_two_wavefront_hand_off_equals_fused(built, kw, L, br):
    """The kernel as it runs on the GPU: the model and coder roles of a block as two host threads with the LDS hand-off
    ring between them (publish / pull / seg_consume, group batches, the runs of SNP-only records, the record loop split at
    POS escapes).  Same payload bytes, status and symbol count as the fused emulation -- and no hang."""
    contigs = [300000, 120000] if L != 100 or br != 1000 else [40_000_000]
    fa, sam, _, _ = synth.dataset(5, contigs, [3000, 1200][:len(contigs)], L, **kw)
    pb = host.pack_sam(sam, fa, block_reads=br)
    p1, r1 = blockref.emu_encode(pb)
    p2, r2 = blockref.emu_encode(pb, two_wave=True)
    assert p1 == p2
    assert (r1["status"] == r2["status"]).all() and (r1["n_symbols"] == r2["n_symbols"]).all()


def test_md_last_column_quirk(built):
    """Q2: the '\\n' letter token -> phantom N->N SNP, same bytes as the oracle."""
    fa, _, rbc, _ = synth.dataset(9, [100000], [600], 100, sub_rate=0.02, indel_frac=0.0)
    sam = synth.sam_text(rbc, md_last=True)
    pb = host.pack_sam(sam, fa, block_reads=200)
    _check(pb, sam)


def test_md_last_with_deletion_aborts_like_the_reference(built):
    """Q2 + a SNP-free deletion read: the phantom SNP's gap equals the read length, which is outside
    the var alphabet -> assert(x < alphabetCard) in the reference; here: status ASSERT, oracle error."""
    fa, _, rbc, _ = synth.dataset(9, [100000], [600], 100, sub_rate=0.0, indel_frac=0.3)
    sam = synth.sam_text(rbc, md_last=True)
    pb = host.pack_sam(sam, fa, block_reads=4096)
    payloads, res = blockref.emu_encode(pb)
    assert int(res[0]["status"]) == 2
    with pytest.raises(oracle.OracleError):
        oracle.encode(sam, fa)


def test_sparse_positions_many_escapes(built):
    """Every POS delta distinct: the escape path and the derived pos_alpha byte models."""
    fa, sam, _, _ = synth.dataset(12, [40_000_000], [1500], 100, sub_rate=0.003, indel_frac=0.02)
    pb = host.pack_sam(sam, fa, block_reads=1000)
    assert pb.cap_pos >= 900
    _check(pb, sam)


def test_packer_cuts_blocks_at_table_caps(built):
    fa, sam, _, _ = synth.dataset(13, [20_000_000], [2000], 100, sub_rate=0.03, indel_frac=0.3)
    pb = host.pack_sam(sam, fa, block_reads=4096, max_cap_pos=256, max_cap_var=512)
    assert pb.n_blocks > 4 and pb.cap_pos <= 256 and pb.cap_var <= 512
    _check(pb, sam)


def test_c_generator_matches_oracle(built):
    pb, sam, fa = host.synth(0xCBC00002, 3_000_000, 12000, 150, want_text=True, block_reads=4096)
    _check(pb, sam)


def test_output_bound_holds(built):
    """cbc_gpu_plan_output's worst case (3 bytes per coded symbol) really bounds the payloads."""
    fa, sam, _, _ = synth.dataset(14, [200000], [3000], 150, sub_rate=0.08, indel_frac=0.8)
    pb = host.pack_sam(sam, fa, block_reads=500)
    payloads, res = blockref.emu_encode(pb)
    blocks = pb.blocks.copy()
    blockref.emu_lib().emu_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, pb.tok.ctypes.data)
    for b in range(pb.n_blocks):
        assert res[b]["status"] == 0
        assert len(payloads[b]) * 4 < int(blocks[b]["out_cap"]) * 3     # comfortably inside


def test_kernel_reports_reference_aborts(built):
    """Inputs the reference would abort on come back as a per-block status, never as bad bytes."""
    fa, sam, rbc, _ = synth.dataset(15, [100000], [100], 100, sub_rate=0.0, indel_frac=0.0)
    # MD claims a mismatch whose letter equals the read base: zero-count chars symbol (quirk Q4)
    recs = rbc[0][2]
    r = recs[50]
    seq = bytearray(r["seq"])
    seq[10] = ord("A") if seq[10] != ord("A") else ord("C")          # real mismatch -> imperfect read
    r["seq"] = bytes(seq)
    r["md"] = "10%s89" % chr(seq[10])                                  # ...but MD names the READ base
    r["nm"] = 1
    sam2 = synth.sam_text(rbc)
    pb = host.pack_sam(sam2, fa, block_reads=4096)
    payloads, res = blockref.emu_encode(pb)
    assert int(res[0]["status"]) == 2 and int(res[0]["fail_read"]) == 50 and payloads[0] == b""
    with pytest.raises(oracle.OracleError):
        oracle.encode(sam2, fa)


@pytest.mark.timeout(300)
def test_two_wavefront_failures_wind_up(built):
    """The two ways a block fails with both wavefronts running: a model-side assert in the middle of a segment (the model
    wavefront reports it through its LAST batch) and a record the coder wavefront refuses in its look-ahead (posted through
    the group mailbox; the model wavefront winds up).  Same status / record as the fused form, no hang, other blocks fine."""
    fa, sam, rbc, _ = synth.dataset(15, [100000], [300], 100, sub_rate=0.0, indel_frac=0.0)
    recs = rbc[0][2]
    r = recs[150]
    seq = bytearray(r["seq"])
    seq[10] = ord("A") if seq[10] != ord("A") else ord("C")
    r["seq"] = bytes(seq)
    r["md"] = "10%s89" % chr(seq[10])                                  # MD names the READ base: zero-count chars symbol (Q4)
    r["nm"] = 1
    pb = host.pack_sam(synth.sam_text(rbc), fa, block_reads=100)
    _, r1 = blockref.emu_encode(pb)
    _, r2 = blockref.emu_encode(pb, two_wave=True)
    assert [int(x) for x in r1["status"]] == [0, 2, 0] == [int(x) for x in r2["status"]]
    assert int(r1[1]["fail_read"]) == int(r2[1]["fail_read"]) == 50
    # a record the validation refuses (POS 0) in the second group of a block: found by the coder one group ahead
    pb = host.pack_sam(sam, fa, block_reads=300)
    pb.recs["pos"][100] = 0
    _, r1 = blockref.emu_encode(pb)
    _, r2 = blockref.emu_encode(pb, two_wave=True)
    assert int(r1[0]["status"]) == int(r2[0]["status"]) == 2
    assert int(r1[0]["fail_read"]) == int(r2[0]["fail_read"]) == 100
    # the CODER wavefront fails on its own (payload area too small, a POS alphabet beyond its cap): it must release the model
    # wavefront, which would otherwise wait for a match mask that never comes while the coder waits for its LAST batch
    pb = host.pack_sam(sam, fa, block_reads=300)                       # five groups: the coder fails in the second
    _, r1 = blockref.emu_encode(pb, shrink_block=0)
    _, r2 = blockref.emu_encode(pb, two_wave=True, shrink_block=0)
    assert int(r1[0]["status"]) == int(r2[0]["status"]) != 0


def _variable_length_sam(seed, n=1500):
    """Reads of mixed lengths (trimmed reads): exercises the rlength[0] cache switch / write-back."""
    rng = np.random.default_rng(seed)
    contig = synth.make_contig(rng, 200000)
    recs, pos = [], 50
    for i in range(n):
        pos += int(rng.integers(0, 60))
        L = int(rng.choice([100, 100, 100, 87, 64, 100, 99, 35]))
        if i % 400 == 1:
            L = 100        # the block's 2nd record sets the header read length in the oracle's block-alone run
        seq = contig[pos - 1: pos - 1 + L].copy()
        md, nm = str(L), 0
        if i % 5 == 0:
            q = int(rng.integers(1, L - 1))
            old = seq[q]
            seq[q] = synth._ACGT[(int(np.where(synth._ACGT == old)[0][0]) + 1) % 4]
            md, nm = "%d%s%d" % (q, chr(old), L - q - 1), 1
        recs.append(dict(pos=pos, flag=16 if i % 3 == 0 else 0, cigar="%dM" % L, seq=seq.tobytes(), md=md, nm=nm))
    rbc = [("chrV", 200000, recs)]
    return synth.fasta_text([("chrV", contig)]), synth.sam_text(rbc)


def test_variable_read_lengths(built):
    fa, sam = _variable_length_sam(17)
    pb = host.pack_sam(sam, fa, block_reads=400)
    assert pb.read_length == 100
    _check(pb, sam)


# ---- decode direction: kernel body through the emulation ----

def _roundtrip(pb, sam, fa):
    payloads, res = blockref.emu_encode(pb)
    assert (res["status"] == 0).all()
    blob = blockref.container_from_payloads(pb, payloads)
    plan = host.UnpackPlan(blob, fa)
    recs, seq, dres = blockref.emu_decode(plan)
    assert (dres["status"] == 0).all(), dres[dres["status"] != 0]
    assert (dres["n_symbols"] == res["n_symbols"]).all()           # same number of coder steps both ways
    expect = b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines() if not ln.startswith(b"@"))
    assert plan.text(recs, seq) == expect
    # POS / FLAG come back too
    lines = blockref.mapped_sam_lines(sam)
    b0 = 0
    for b in range(plan.n_blocks):
        n = int(plan.blocks[b]["n_reads"])
        w0 = int(plan.window_start[b])
        for k in (0, n // 2, n - 1):
            f = lines[b0 + k].split(b"\t")
            assert int(recs[b0 + k]["pos"]) + w0 == int(f[3]) and int(recs[b0 + k]["flag"]) == int(f[1])
        b0 += n
    return plan, payloads


@pytest.mark.parametrize("kw,L,br", [
    (dict(sub_rate=0.0, indel_frac=0.0), 100, 1000),
    (dict(), 150, 1024),
    (dict(sub_rate=0.02, indel_frac=0.5, trailing_s_frac=0.2, dup_pos_frac=0.1), 100, 512),
    (dict(sub_rate=0.05, indel_frac=0.0), 252, 300),
    (dict(flags=(0, 16, 83, 99, 147, 163)), 150, 2048),
])
def test_decode_round_trip(built, kw, L, br):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [4000, 1500], L, **kw)
    pb = host.pack_sam(sam, fa, block_reads=br)
    _roundtrip(pb, sam, fa)


def test_decode_equals_oracle_decoder_per_block(built):
    """The oracle's restatement of the reference DEcoder, run on each block's payload, gives the same reads."""
    fa, sam, _, _ = synth.dataset(6, [200000], [1500], 100, sub_rate=0.01, indel_frac=0.3)
    pb = host.pack_sam(sam, fa, block_reads=500)
    plan, payloads = _roundtrip(pb, sam, fa)
    lines = blockref.mapped_sam_lines(sam)
    for b in range(pb.n_blocks):
        bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
        text, nr = oracle.decode(payloads[b], bfa)
        assert nr == int(pb.blocks[b]["n_reads"])
        first = int(pb.blocks[b]["rec_base"])
        assert text == b"".join(lines[first + k].split(b"\t")[9] + b"\n" for k in range(nr))


def test_decode_variable_lengths_and_sparse_positions(built):
    fa, sam = _variable_length_sam(23)
    _roundtrip(host.pack_sam(sam, fa, block_reads=400), sam, fa)
    fa, sam, _, _ = synth.dataset(12, [40_000_000], [1500], 100)
    _roundtrip(host.pack_sam(sam, fa, block_reads=1000), sam, fa)


def test_decode_rejects_corrupt_payload(built):
    """Bit flips must end in a status (or different reads), never in an out-of-range access."""
    fa, sam, _, _ = synth.dataset(7, [100000], [600], 100, sub_rate=0.01, indel_frac=0.2)
    pb = host.pack_sam(sam, fa, block_reads=600)
    payloads, res = blockref.emu_encode(pb)
    rng = np.random.default_rng(5)
    for trial in range(12):
        bad = bytearray(payloads[0])
        for _ in range(3):
            i = int(rng.integers(70, len(bad)))
            bad[i] ^= 1 << int(rng.integers(0, 8))
        blob = blockref.container_from_payloads(pb, [bytes(bad)])
        plan = host.UnpackPlan(blob, fa)
        recs, seq, dres = blockref.emu_decode(plan)          # must simply return
        assert int(dres[0]["status"]) in (0, 2, 3, 4, 5, 6, 7)


def _soft_clip_sam(seed, n=600):
    """Leading / trailing / double soft clips, with and without a mismatch (quirk Q6 path)."""
    rng = np.random.default_rng(seed)
    contig = synth.make_contig(rng, 50000)
    recs, pos, L = [], 100, 100
    A = synth._ACGT
    for i in range(n):
        pos += int(rng.integers(1, 40))
        kind = i % 4
        fl = 16 if i % 3 == 0 else 0
        if kind == 0:
            k = int(rng.integers(1, 6))
            seq = np.concatenate([A[rng.integers(0, 4, size=k)], contig[pos - 1:pos - 1 + L - k]]).tobytes()
            recs.append(dict(pos=pos, flag=fl, cigar="%dS%dM" % (k, L - k), seq=seq, md=str(L - k), nm=0))
        elif kind == 1:
            k = int(rng.integers(1, 6))
            body = contig[pos - 1:pos - 1 + L - k].copy()
            q = int(rng.integers(5, L - k - 5))
            old = body[q]
            body[q] = A[(int(np.where(A == old)[0][0]) + 1) % 4]
            seq = np.concatenate([A[rng.integers(0, 4, size=k)], body]).tobytes()
            recs.append(dict(pos=pos, flag=fl, cigar="%dS%dM" % (k, L - k), seq=seq,
                             md="%d%s%d" % (q, chr(old), L - k - q - 1), nm=1))
        elif kind == 2:
            k, k2 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
            seq = np.concatenate([A[rng.integers(0, 4, size=k)], contig[pos - 1:pos - 1 + L - k - k2],
                                  A[rng.integers(0, 4, size=k2)]]).tobytes()
            recs.append(dict(pos=pos, flag=fl, cigar="%dS%dM%dS" % (k, L - k - k2, k2), seq=seq, md=str(L - k - k2), nm=0))
        else:
            recs.append(dict(pos=pos, flag=fl, cigar="%dM" % L, seq=contig[pos - 1:pos - 1 + L].tobytes(), md=str(L), nm=0))
    rbc = [("chrS", 50000, recs)]
    return synth.fasta_text([("chrS", contig)]), synth.sam_text(rbc)


def test_soft_clips_match_oracle_and_round_trip(built):
    """Leading soft clips go through the packer's restatement of the reference's in-place MD rebuild."""
    fa, sam = _soft_clip_sam(3)
    pb = host.pack_sam(sam, fa, block_reads=256)
    _check(pb, sam)
    _roundtrip(pb, sam, fa)


@pytest.mark.parametrize("site_every,err,br", [(100, 0.0, 1500), (60, 0.004, 4096), (300, 0.001, 700)])
def test_shared_variants_repeat_var_contexts(built, site_every, err, br):
    """Every read covering a variant site carries it: var contexts repeat with one symbol each (context cache
    hits), sequencing errors add second symbols (mixed entries -> filter + list scan) and other contexts
    evict them.  Bytes == oracle, and the decode round trip returns the reads."""
    fa, sam = synth.shared_variant_dataset(21, 60000, 3000, 100, site_every, err)
    pb = host.pack_sam(sam, fa, block_reads=br)
    payloads, _ = _check(pb, sam)
    blob = blockref.container_from_payloads(pb, payloads)
    plan = host.UnpackPlan(blob, fa)
    recs, seq, res = blockref.emu_decode(plan)
    assert (res["status"] == 0).all()
    assert plan.text(recs, seq) == b"".join(l.split(b"\t")[9] + b"\n" for l in sam.split(b"\n") if l)


def test_reads_with_dozens_of_edits(built):
    """~65 SNPs per read plus indels (the queue is drained many times inside one record)."""
    fa, sam, _, _ = synth.dataset(77, [60000], [600], 150, sub_rate=0.45, indel_frac=0.3)
    pb = host.pack_sam(sam, fa, block_reads=256)
    _check(pb, sam)


def test_block_larger_than_the_contract_is_refused(built):
    """The per-record models are coded as counting models, which holds while no total can reach the 2^20 rescale
    point, i.e. up to CBC_MAX_BLOCK_READS records per block: a descriptor that claims more comes back as
    CBC_ST_UNSUPPORTED (7) and the other blocks are unaffected (the decoder has the same check)."""
    pb = host.synth(12, 3_000_000, 40000, 150, block_reads=16384)
    assert pb.n_blocks == 3
    ok_payloads, ok_res = blockref.emu_encode(pb)
    assert (ok_res["status"] == 0).all()
    saved = int(pb.blocks[0]["n_reads"])
    pb.blocks[0]["n_reads"] = saved + 1                       # reaches into the next block's first record
    try:
        payloads, res = blockref.emu_encode(pb)
    finally:
        pb.blocks[0]["n_reads"] = saved
    assert int(res[0]["status"]) == 7 and payloads[0] == b""
    assert payloads[1:] == ok_payloads[1:]


def test_blocks_at_the_cap_limits(built):
    """16 384-record blocks and blocks cut by the var-symbol cap (32 768): the largest the packer makes.  No
    adaptive total reaches 2^20 inside them (the block contract); kernel body == the CPU port."""
    for args in ((77, 3_000_000, 40_000, 150, 0.003, 0.02), (78, 400_000, 1_200, 150, 0.45, 0.3)):
        p = host.synth(*args, block_reads=16384, max_cap_var=32768)
        ep, eres = blockref.emu_encode(p)
        cp, cres = oracle.cpu_encode_blocks(p, return_payloads=True)
        assert (eres["status"] == 0).all() and ep == cp and (eres["n_symbols"] == cres["n_symbols"]).all()
        p.close()
