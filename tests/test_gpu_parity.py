"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle."""
import glob
import json
import os

import numpy as np
import pytest

import blockref
import synth
from cbc_amd import gpu, host
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def enc():
    e = gpu.Encoder(0)
    yield e
    e.close()


def _check_blocks(enc, pb, sam):
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all(), res[res["status"] != 0]
    lines = blockref.mapped_sam_lines(sam)
    assert len(lines) == pb.n_recs
    for b in range(pb.n_blocks):
        bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
        exp = oracle.encode(bsam, bfa)
        assert payloads[b] == exp, "block %d: %d bytes vs oracle %d" % (b, len(payloads[b]), len(exp))
    return payloads


def test_kat_prefix(enc, built):
    """SURVEY.md 8a known-answer: L=100 stream starts 00 00 00 64 55 ff ff d4 85 79 db 94."""
    fa, sam, _, _ = synth.dataset(1, [200000], [4], 100, sub_rate=0.0, indel_frac=0.0)
    pb = host.pack_sam(sam, fa)
    p = _check_blocks(enc, pb, sam)
    assert p[0][:12].hex(" ") == "00 00 00 64 55 ff ff d4 85 79 db 94"
    assert len(p[0]) == 121


@pytest.mark.parametrize("kw,L,br", [
    (dict(sub_rate=0.0, indel_frac=0.0), 100, 1000),
    (dict(), 150, 1024),
    (dict(sub_rate=0.01, indel_frac=0.3), 150, 700),
    (dict(sub_rate=0.02, indel_frac=0.5, trailing_s_frac=0.2, dup_pos_frac=0.1), 100, 512),
    (dict(flags=(0, 16, 83, 99, 147, 163)), 150, 2048),
])
def test_blocks_match_oracle(enc, built, kw, L, br):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [4000, 1500], L, **kw)
    pb = host.pack_sam(sam, fa, block_reads=br)
    _check_blocks(enc, pb, sam)


def test_c_generator_matches_oracle(enc, built):
    pb, sam, fa = host.synth(0xCBC00002, 3_000_000, 20000, 150, want_text=True, block_reads=4096)
    _check_blocks(enc, pb, sam)


def test_gpu_equals_emulation_large(enc, built):
    """Larger, sparser input (many distinct POS deltas): GPU result == lock-step emulation result."""
    pb = host.synth(3, 50_000_000, 100_000, 150, block_reads=4096)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all()
    ep, eres = blockref.emu_encode(pb)
    assert payloads == ep
    assert (res["n_symbols"] == eres["n_symbols"]).all()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_L*.json"))))
def test_golden_block_payloads(enc, built, path):
    """Committed vectors (tests/golden/make_golden.py): per-block payloads for block_reads=128."""
    g = json.load(open(path))
    pb = host.pack_sam(g["sam"].encode(), g["fasta"].encode(), block_reads=g["block_reads"])
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all()
    assert [p.hex() for p in payloads] == g["block_payload_hex"]


def test_reference_aborts_come_back_as_status(enc, built):
    fa, sam, rbc, _ = synth.dataset(15, [100000], [100], 100, sub_rate=0.0, indel_frac=0.0)
    r = rbc[0][2][50]
    seq = bytearray(r["seq"])
    seq[10] = ord("A") if seq[10] != ord("A") else ord("C")
    r["seq"] = bytes(seq)
    r["md"] = "10%s89" % chr(seq[10])
    pb = host.pack_sam(synth.sam_text(rbc), fa)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert int(res[0]["status"]) == 2 and int(res[0]["fail_read"]) == 50 and payloads[0] == b""


def test_full_size_properties(enc, built):
    """cfg2-shaped launch (1M-read slice of it): size-independent checks.
    - determinism: two launches give identical bytes
    - every payload starts with the 4 header bytes of its read length; offsets are consistent
    - sampled blocks (first, middle, last) equal the lock-step emulation byte for byte, i.e. a
      block's payload does not depend on the other 240+ blocks of the launch."""
    pb = host.synth(0xCBC00002, 248_956_422 // 10, 1_000_000, 150, block_reads=4096)
    enc.upload_reference(pb.ref)
    p1, r1, o1, f1 = enc.encode_blocks(pb)
    p2, r2, o2, f2 = enc.encode_blocks(pb)
    assert (r1["status"] == 0).all() and f1.tobytes() == f2.tobytes()
    assert all(p[:4] == bytes([0, 0, 0, 150]) for p in p1)
    assert int(o1[-1]) == sum(len(p) for p in p1)
    ep = blockref.emu_encode_blocks(pb, [0, pb.n_blocks // 2, pb.n_blocks - 1])
    for b, payload, res in ep:
        assert p1[b] == payload and int(r1[b]["n_symbols"]) == int(res["n_symbols"])


def test_million_reads_sampled_blocks_vs_the_oracle_text_path(enc, built):
    """The full-size tests check the GPU against the CPU port, which consumes the packer's own tokens: a packer or generator
    fault would be common to both sides.  Here, at 1 M reads (245 blocks), every 12th block is also coded by the oracle's TEXT
    path from the block's own SAM lines and FASTA window -- its tokeniser, not the packer's."""
    pb, sam, fa = host.synth(0xCBC00002, 248_956_422 // 10, 1_000_000, 150, want_text=True, block_reads=4096)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all()
    lines = blockref.mapped_sam_lines(sam)
    assert len(lines) == pb.n_recs
    which = sorted(set(range(0, pb.n_blocks, 12)) | {pb.n_blocks - 1})
    for b in which:
        bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
        assert payloads[b] == oracle.encode(bsam, bfa), b
    assert len(which) >= 20


def test_variable_read_lengths(enc, built):
    from test_emu_parity import _variable_length_sam
    fa, sam = _variable_length_sam(17)
    pb = host.pack_sam(sam, fa, block_reads=400)
    _check_blocks(enc, pb, sam)


# ---- decode direction on the GPU ----

def _gpu_roundtrip(enc, pb, sam, fa):
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all()
    blob = pb.container(flat, offs)
    plan = host.UnpackPlan(blob, fa)
    enc.upload_reference(plan.ref)
    recs, seq, dres = enc.decode_blocks(plan)
    assert (dres["status"] == 0).all(), dres[dres["status"] != 0]
    assert (dres["n_symbols"] == res["n_symbols"]).all()
    expect = b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines() if not ln.startswith(b"@"))
    assert plan.text(recs, seq) == expect
    return blob


@pytest.mark.parametrize("kw,L,br", [
    (dict(), 150, 1024),
    (dict(sub_rate=0.02, indel_frac=0.5, trailing_s_frac=0.2, dup_pos_frac=0.1), 100, 512),
    (dict(flags=(0, 16, 83, 99, 147, 163)), 150, 2048),
])
def test_decode_round_trip(enc, built, kw, L, br):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [4000, 1500], L, **kw)
    _gpu_roundtrip(enc, host.pack_sam(sam, fa, block_reads=br), sam, fa)


def test_decode_equals_emulation(enc, built):
    pb, sam, fa = host.synth(9, 2_000_000, 30000, 150, want_text=True, block_reads=4096)
    blob = _gpu_roundtrip(enc, pb, sam, fa)
    plan = host.UnpackPlan(blob, fa)
    er, es, eres = blockref.emu_decode(plan)
    enc.upload_reference(plan.ref)
    gr, gs, gres = enc.decode_blocks(plan)
    n = plan.n_recs * plan.seq_stride
    assert gr.tobytes() == er.tobytes() and gs[:n].tobytes() == es[:n].tobytes()


def test_full_size_round_trip(enc, built):
    """Size-independent property at cfg2 scale (1M-read slice): encode -> decode == the packed bases."""
    pb = host.synth(0xCBC00002, 248_956_422 // 10, 1_000_000, 150, block_reads=4096)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all()
    # decode against the device reference already uploaded: lay the launch out from the packed batch
    import ctypes
    blocks = np.zeros(pb.n_blocks, dtype=host.DEC_BLOCK_DTYPE)
    stride = 152
    nrec = 0
    for b in range(pb.n_blocks):
        blocks[b]["in_off"] = int(offs[b]); blocks[b]["in_bytes"] = int(offs[b + 1] - offs[b])
        blocks[b]["ref_off"] = int(pb.blocks[b]["ref_off"]); blocks[b]["rec_base"] = nrec
        blocks[b]["seq_base"] = nrec * stride; blocks[b]["n_reads"] = int(pb.blocks[b]["n_reads"])
        blocks[b]["read_length"] = 150; blocks[b]["seq_stride"] = stride
        nrec += int(pb.blocks[b]["n_reads"])
    recs = np.zeros(nrec, dtype=host.REC_DTYPE)
    seq = np.zeros(nrec * stride + 8, dtype=np.uint8)
    dres = np.zeros(pb.n_blocks, dtype=host.RESULT_DTYPE)
    caps = host.LdsCaps(pb.cap_pos, pb.cap_var)
    pay = np.ascontiguousarray(flat)
    rc = gpu.lib().cbc_gpu_decode_blocks(enc._ctx, pay.ctypes.data, pay.size, blocks.ctypes.data, pb.n_blocks,
                                         ctypes.byref(caps), recs.ctypes.data, nrec, seq.ctypes.data, seq.size, dres.ctypes.data)
    assert rc == 0 and (dres["status"] == 0).all()
    got = seq[:nrec * stride].reshape(nrec, stride)[:, :150]
    want = pb.seq[:nrec * 150].reshape(nrec, 150)
    assert (got == want).all()
    assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all()


def test_cli_compress_decompress(built, tmp_path):
    """The `cbc` binary end to end: -c then -d gives back the SEQ column."""
    import subprocess, os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cbc_amd", "csrc", "cbc")
    fa, sam, _, _ = synth.dataset(8, [200000, 80000], [3000, 1000], 100, sub_rate=0.01, indel_frac=0.2)
    (tmp_path / "in.sam").write_bytes(sam); (tmp_path / "ref.fa").write_bytes(fa)
    r = subprocess.run([exe, "-c", "1", str(tmp_path / "in.sam"), str(tmp_path / "out.cbc"), str(tmp_path / "ref.fa"),
                        "--block-reads", "1000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Final Size:" in r.stdout
    r = subprocess.run([exe, "-d", str(tmp_path / "out.cbc"), str(tmp_path / "reads.txt"), str(tmp_path / "ref.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    expect = b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines() if not ln.startswith(b"@"))
    assert (tmp_path / "reads.txt").read_bytes() == expect
    # same container as the library path
    pb = host.pack_sam(sam, fa, block_reads=1000)
    enc = gpu.Encoder(0); enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (tmp_path / "out.cbc").read_bytes() == pb.container(flat, offs)
    enc.close()


def test_soft_clips(enc, built):
    from test_emu_parity import _soft_clip_sam
    fa, sam = _soft_clip_sam(3)
    pb = host.pack_sam(sam, fa, block_reads=256)
    _check_blocks(enc, pb, sam)
    _gpu_roundtrip(enc, pb, sam, fa)


# ------------------------------------------------------------------ two-wavefront hand-off, edge shapes
def _gpu_vs_emu(enc, pb):
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    ep, eres = blockref.emu_encode(pb)
    assert (res["status"] == eres["status"]).all() and (res["fail_read"] == eres["fail_read"]).all()
    assert payloads == ep
    assert (res["n_symbols"] == eres["n_symbols"]).all()
    return payloads, res


@pytest.mark.parametrize("n_reads", [2, 63, 64, 65, 128, 129, 1000])
def test_group_boundaries(enc, built, n_reads):
    """Record counts around the 64-record group size: the GROUP batches and the last, partial group."""
    fa, sam, _, _ = synth.dataset(40 + n_reads, [40000], [n_reads], 100, sub_rate=0.01, indel_frac=0.1)
    pb = host.pack_sam(sam, fa, block_reads=4096)
    _check_blocks(enc, pb, sam)


def test_all_perfect_and_all_imperfect_blocks(enc, built):
    """Groups with no imperfect record (GROUP batches only, no segments) and groups where every record has
    many edits (segments spill over several batches while the coder is still on earlier ones)."""
    fa, sam, _, _ = synth.dataset(71, [200000], [3000], 150, sub_rate=0.0, indel_frac=0.0)
    pb = host.pack_sam(sam, fa, block_reads=1024)
    _check_blocks(enc, pb, sam)
    fa, sam, _, _ = synth.dataset(72, [200000], [3000], 150, sub_rate=0.06, indel_frac=0.9)
    pb = host.pack_sam(sam, fa, block_reads=1024)
    _check_blocks(enc, pb, sam)


def test_long_contig_name_segment(enc, built):
    """A 120-character RNAME: the name segment spans several batches before the first record's symbols."""
    name = "chr_" + "x" * 116
    fa, sam, _, _ = synth.dataset(73, [50000], [500], 100, names=[name])
    pb = host.pack_sam(sam, fa, block_reads=256)
    _check_blocks(enc, pb, sam)


def test_many_small_blocks_fill_the_gpu(enc, built):
    """Thousands of 64..200-record blocks: every CU holds many workgroups, each with its own ring."""
    pb = host.synth(9, 30_000_000, 400_000, 150, block_reads=128)
    assert pb.n_blocks > 3000
    _gpu_vs_emu(enc, pb)


def test_failures_on_either_wavefront_end_the_block_cleanly(enc, built):
    """A model-side abort (MD inconsistent with the read: reference assert) and a coder-side one (POS table
    cap) in a batch of otherwise good blocks: statuses equal the emulation's, the other blocks are intact."""
    fa, sam, rbc, _ = synth.dataset(15, [100000], [700], 100, sub_rate=0.0, indel_frac=0.0)
    r = rbc[0][2][350]
    seq = bytearray(r["seq"]); seq[10] = ord("A") if seq[10] != ord("A") else ord("C")
    r["seq"] = bytes(seq); r["md"] = "10%s89" % chr(seq[10])
    pb = host.pack_sam(synth.sam_text(rbc), fa, block_reads=100)
    payloads, res = _gpu_vs_emu(enc, pb)
    assert int(res[3]["status"]) == 2 and int(res[3]["fail_read"]) == 50 and payloads[3] == b""
    assert all(int(s) == 0 for i, s in enumerate(res["status"]) if i != 3)
    # coder side: sparse positions with a POS table smaller than the block needs
    pb = host.synth(4, 200_000_000, 3000, 150, block_reads=1000, max_cap_pos=4096)
    assert pb.cap_pos > 64
    pb.cap_pos = 64                                   # both the GPU call and the emulation take the caps from here
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    ep, eres = blockref.emu_encode(pb)
    assert (res["status"] == eres["status"]).all() and (res["fail_read"] == eres["fail_read"]).all()
    assert (res["status"] == 3).all() and payloads == ep            # CBC_ST_CAP_POS


def test_piece_by_piece_bit_packing_on_the_gpu(built):
    """pack() places strings longer than 32 bits (long E3 runs) piece by piece; that path is rare.  A test
    build of the same kernels takes it for every string longer than 9 bits: its payloads must equal the
    oracle's too.  Runs in a child process because the library is chosen at import time."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "cbc_amd", "csrc", "libcbc_gpu_serialpack.so")
    assert os.path.exists(lib)
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import synth, blockref
        from cbc_amd import gpu, host
        from oracle import oracle
        fa, sam, _, _ = synth.dataset(91, [300000], [6000], 150, sub_rate=0.01, indel_frac=0.2)
        pb = host.pack_sam(sam, fa, block_reads=1500)
        enc = gpu.Encoder(0); enc.upload_reference(pb.ref)
        payloads, res, offs, flat = enc.encode_blocks(pb)
        assert (res["status"] == 0).all()
        lines = blockref.mapped_sam_lines(sam)
        for b in range(pb.n_blocks):
            bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
            assert payloads[b] == oracle.encode(bsam, bfa), b
        print("SERIALPACK_OK", pb.n_blocks)
    """ % (root, os.path.join(root, "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CBC_GPU_LIB=lib), capture_output=True, text=True)
    assert "SERIALPACK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("site_every,err", [(100, 0.0), (300, 0.001)])
def test_shared_variants(enc, built, site_every, err):
    """Shared variant sites (var contexts that repeat with one symbol: the context cache) on the GPU: bytes ==
    oracle per block, decode round trip == the reads."""
    fa, sam = synth.shared_variant_dataset(33, 400000, 20000, 150, site_every, err)
    pb = host.pack_sam(sam, fa, block_reads=4096)
    payloads = _check_blocks(enc, pb, sam)
    blob = blockref.container_from_payloads(pb, payloads)
    plan = host.UnpackPlan(blob, fa)
    enc.upload_reference(plan.ref)
    recs, seq, res = enc.decode_blocks(plan)
    assert (res["status"] == 0).all()
    assert plan.text(recs, seq) == b"".join(l.split(b"\t")[9] + b"\n" for l in sam.split(b"\n") if l)


def test_reads_with_dozens_of_edits(enc, built):
    """~65 SNPs per read plus indels: one record's edit segment is longer than a hand-off batch, so it spans
    several batches while the coder wave is inside it."""
    fa, sam, _, _ = synth.dataset(77, [60000], [600], 150, sub_rate=0.45, indel_frac=0.3)
    pb = host.pack_sam(sam, fa, block_reads=256)
    _check_blocks(enc, pb, sam)


# ------------------------------------------------------------------ BASELINE.json configs at their full sizes
def _decode_all(enc, pb, flat, offs, L):
    """Decode every block of a packed batch from the compacted payloads; returns (recs, bases[n, L])."""
    import ctypes
    stride = (L + 3) // 4 * 4
    nb = pb.n_blocks
    blocks = np.zeros(nb, dtype=host.DEC_BLOCK_DTYPE)
    rb = np.concatenate([[0], np.cumsum(pb.blocks["n_reads"].astype(np.uint64))])
    blocks["in_off"] = offs[:-1]; blocks["in_bytes"] = (offs[1:] - offs[:-1]).astype(np.uint32)
    blocks["ref_off"] = pb.blocks["ref_off"]; blocks["rec_base"] = rb[:-1]; blocks["seq_base"] = rb[:-1] * stride
    blocks["n_reads"] = pb.blocks["n_reads"]; blocks["read_length"] = L; blocks["seq_stride"] = stride
    nrec = int(rb[-1])
    recs = np.zeros(nrec, dtype=host.REC_DTYPE)
    seq = np.zeros(nrec * stride + 8, dtype=np.uint8)
    dres = np.zeros(nb, dtype=host.RESULT_DTYPE)
    caps = host.LdsCaps(pb.cap_pos, pb.cap_var)
    pay = np.ascontiguousarray(flat)
    rc = gpu.lib().cbc_gpu_decode_blocks(enc._ctx, pay.ctypes.data, pay.size, blocks.ctypes.data, nb, ctypes.byref(caps),
                                         recs.ctypes.data, nrec, seq.ctypes.data, seq.size, dres.ctypes.data)
    assert rc == 0 and (dres["status"] == 0).all(), dres[dres["status"] != 0][:4]
    return recs, seq[:nrec * stride].reshape(nrec, stride)[:, :L], dres


def test_cfg1_shape_every_block_vs_oracle(enc, built):
    """BASELINE config 1's shape (15 072 434 bp contig, 100 k x 100 bp) on the HIP path: every one of the 25
    blocks byte-equal to the oracle run on that block alone (SAM text path), and the GPU round trip."""
    pb, sam, fa = host.synth(0xCBC00001, 15_072_434, 100_000, 100, want_text=True, block_reads=4096)
    assert pb.n_blocks == 25
    _check_blocks(enc, pb, sam)
    assert enc.last_kernel_variant() == 5
    _gpu_roundtrip(enc, pb, sam, fa)


def test_cfg2_full_size_every_block_vs_cpu_port(enc, built):
    """BASELINE config 2 in full (10 M x 150 bp vs a chr1-sized contig, 2442 blocks): every block's payload and
    coder-step count == the oracle's packed-input CPU port run on that block alone; decode returns the packed
    bases, POS and FLAG of all 10 M records."""
    pb = host.synth(0xCBC00002, 248_956_422, 10_000_000, 150, block_reads=4096)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all()
    assert enc.last_kernel_variant() == 5                    # 2442 blocks <= 10 per CU: the 5-waves-per-SIMD build
    cp, cres = oracle.cpu_encode_blocks(pb, return_payloads=True)
    assert (cres["status"] == 0).all()
    bad = [b for b in range(pb.n_blocks) if payloads[b] != cp[b]]
    assert not bad, bad[:8]
    assert (res["n_symbols"] == cres["n_symbols"]).all()
    recs, bases, dres = _decode_all(enc, pb, flat, offs, 150)
    assert (bases == pb.seq[:pb.n_recs * 150].reshape(pb.n_recs, 150)).all()
    assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all()
    assert (dres["n_symbols"] == res["n_symbols"]).all()


def test_cfg3_full_size_encode_decode_vs_cpu_port(enc, built):
    """BASELINE config 3 (chr1-sized contig at 30x: 49 791 284 x 150 bp, 12 157 blocks -> the 6-waves-per-SIMD
    build of the kernel): encode, decode, diff against the packed bases / POS / FLAG of every record, and every
    8th block (plus the first and last 16) byte-equal to the CPU port run on the block alone."""
    pb = host.synth(0xCBC00003, 248_956_422, 49_791_284, 150, block_reads=4096)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)
    assert (res["status"] == 0).all()
    assert pb.n_blocks > 12000 and enc.last_kernel_variant() == 6
    recs, bases, dres = _decode_all(enc, pb, flat, offs, 150)
    assert (bases == pb.seq[:pb.n_recs * 150].reshape(pb.n_recs, 150)).all()
    assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all()
    assert (dres["n_symbols"] == res["n_symbols"]).all()
    which = sorted(set(range(0, pb.n_blocks, 8)) | set(range(16)) | set(range(pb.n_blocks - 16, pb.n_blocks)))
    cp, cres = oracle.cpu_encode_blocks(pb, blocks=which, return_payloads=True)
    bad = [b for i, b in enumerate(which) if payloads[b] != cp[i] or int(res[b]["n_symbols"]) != int(cres[i]["n_symbols"])]
    assert len(which) > 1500 and not bad, bad[:8]


def test_blocks_at_the_cap_limits(enc, built):
    """The largest blocks the packer makes: 16 384 records, and ~65 edits per read so that the var-symbol cap
    (32 768 per block) is what cuts them.  No adaptive total can reach the 2^20 rescale point inside such a block
    (chars: 41 + 8 * 32768; snps: L + 10 * 16384; indels: L + 48 * 16384; flag: 65536 + 8 * 16384), which is the
    block contract -- rescales are exercised by the whole-file stream tests.  GPU == CPU port, encode and decode."""
    pb = host.synth(77, 3_000_000, 60_000, 150, 0.003, 0.02, block_reads=16384, max_cap_var=32768)
    assert int(pb.blocks["n_reads"].max()) == 16384
    pd = host.synth(78, 400_000, 4_000, 150, 0.45, 0.3, block_reads=16384, max_cap_var=32768)
    assert pd.cap_var > 30000
    for p in (pb, pd):
        enc.upload_reference(p.ref)
        payloads, res, offs, flat = enc.encode_blocks(p)
        assert (res["status"] == 0).all()
        cp, cres = oracle.cpu_encode_blocks(p, return_payloads=True)
        assert payloads == cp and (res["n_symbols"] == cres["n_symbols"]).all()
        recs, bases, dres = _decode_all(enc, p, flat, offs, 150)
        assert (bases == p.seq[:p.n_recs * 150].reshape(p.n_recs, 150)).all()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launched_bare_with_two_ranks(built, scaling):
    """`python bench.py --gpus 2` exactly as the driver starts it (no torch.distributed.run): the parent spawns
    its own ranks.  Two ranks share the one GPU of this box, so the collectives go over gloo; strong scaling
    checks the gathered container against the single-GPU container inside the run."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--reads", "300000",
                        "--contig-len", "8000000", "--steps", "2", "--warmup", "1", "--scaling", scaling, "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == scaling and j["value"] > 0
    if scaling == "strong":
        assert j["config"]["gathered_equals_single_gpu"] is True


def test_cli_several_devices(built, tmp_path):
    """`cbc --devices a,b`: one host thread and one context per listed device, contigs dealt largest-first (encode),
    contiguous block ranges (decode).  This box has one GPU, so both contexts live on device 0; the container and
    the decoded text must not depend on the device count."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cbc_amd", "csrc", "cbc")
    fa, sam, _, _ = synth.dataset(18, [60000, 200000, 90000], [900, 3000, 1400], 100, sub_rate=0.01, indel_frac=0.2)
    (tmp_path / "in.sam").write_bytes(sam); (tmp_path / "ref.fa").write_bytes(fa)
    outs = []
    for tag, extra in (("one", []), ("two", ["--devices", "0,0"]), ("three", ["--devices", "0,0,0"])):
        o = tmp_path / ("out_%s.cbc" % tag)
        r = subprocess.run([exe, "-c", "1", str(tmp_path / "in.sam"), str(o), str(tmp_path / "ref.fa"), "--block-reads", "256", "--verbose"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(o.read_bytes())
    assert outs[0] == outs[1] == outs[2]
    expect = b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines() if not ln.startswith(b"@"))
    for extra in ([], ["--devices", "0,0"]):
        r = subprocess.run([exe, "-d", str(tmp_path / "out_two.cbc"), str(tmp_path / "reads.txt"), str(tmp_path / "ref.fa")] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "reads.txt").read_bytes() == expect


# ------------------------------------------------------------------ BASELINE config 4 through bench.py's own generator
def _cfg4_inputs(scale, L=150):
    """bench.py's cfg4 workload (24 GRCh38-shaped contigs, 30x, the C generator per contig with bench's seeds) as ONE
    SAM + FASTA pair, packed into one 24-contig batch by the host packer."""
    import bench
    lens, reads = bench.cfg4_workload(scale, L)
    sams, fas = [], []
    for c in range(24):
        pb, sam, fa = host.synth(0xCBC00004 + c, lens[c], reads[c], L, 0.003, 0.02, bench.GRCH38_NAMES[c].encode(), want_text=True)
        pb.close()
        sams.append(b"".join(l for l in sam.splitlines(keepends=True) if not l.startswith(b"@")))
        fas.append(fa)
    return b"".join(sams), b"".join(fas), reads


def _cfg4_worker(rank, world, port, q, scale):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    import bench
    from cbc_amd import gpu as G, host as H, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sam, fa, reads = _cfg4_inputs(scale)
    pb = H.pack_sam(sam, fa, block_reads=4096)
    enc = G.Encoder(0)                                        # both ranks share the one GPU of this box
    enc.upload_reference(pb.ref)
    ran = []

    def encode_blocks(which):
        ran.extend(which)
        if not which:
            return torch.zeros(0, dtype=torch.uint8), torch.zeros(0, dtype=torch.int64)
        payloads, res, offs, flat = enc.encode_blocks(pb, which=which)
        assert (res["status"] == 0).all()
        return torch.from_numpy(flat.copy()), torch.from_numpy(np.diff(offs.astype(np.int64)))

    mine, allp, alls = shard.encode_sharded_by_contig(dist, pb, encode_blocks, torch.device("cpu"), dst=0)
    part = [int(x) for x in pb.assign_contigs(world)]
    ok = ran == mine and part == bench.assign_largest_first(reads, world)
    if rank == 0:
        offs = np.concatenate([[0], np.cumsum(alls.numpy())]).astype(np.uint64)
        blob = pb.container(allp.numpy(), offs)
        q.put((ok, blob, part, len(mine), pb.n_blocks))
    enc.close()
    dist.destroy_process_group()


def test_cfg4_contig_sharded_two_ranks_vs_oracle(enc, built):
    """BASELINE config 4 (24 contigs with GRCh38's primary lengths, 30x, 150 bp) at a small scale through bench.py's own
    generator and assignment rule: two gloo ranks share this box's GPU, each codes the blocks of the contigs it is dealt
    (shard.encode_sharded_by_contig, the function `bench.py --workload cfg4` and `cbc --devices` mirror), rank 0 puts the
    gathered bitstreams back into block order.  Every block == the oracle's packed-input CPU port on that block alone,
    sampled blocks == the oracle on the block's own SAM text, the container == the one-rank container, and the decode of
    the gathered container == the SEQ column."""
    import torch.multiprocessing as mp
    scale = 0.0007                                            # ~430 k reads, 24 contigs of 174 k .. 33 k bases
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_cfg4_worker, args=(r, 2, port, q, scale)) for r in range(2)]
    for p in procs:
        p.start()
    ok, blob, part, n_mine, nb = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok and len(set(part)) == 2 and 0 < n_mine < nb and nb >= 100
    sam, fa, reads = _cfg4_inputs(scale)
    pb = host.pack_sam(sam, fa, block_reads=4096)
    assert len(pb.contigs) == 24 and pb.n_blocks == nb and pb.n_recs == sum(reads)
    enc.upload_reference(pb.ref)
    payloads, res, offs, flat = enc.encode_blocks(pb)         # the one-rank result
    assert (res["status"] == 0).all()
    assert blob == pb.container(flat, offs)
    cp, cres = oracle.cpu_encode_blocks(pb, return_payloads=True)
    assert payloads == cp and (res["n_symbols"] == cres["n_symbols"]).all()
    lines = blockref.mapped_sam_lines(sam)
    for b in sorted(set(range(0, nb, 9)) | {nb - 1}):
        bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
        assert payloads[b] == oracle.encode(bsam, bfa), b
    plan = host.UnpackPlan(blob, fa)
    enc.upload_reference(plan.ref)
    recs, seq, dres = enc.decode_blocks(plan)
    assert (dres["status"] == 0).all()
    assert plan.text(recs, seq) == b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines())


def test_paired_end_flag_sets_of_more_than_64_values(enc, built):
    """A paired-end file with secondary / supplementary / duplicate flags: well over 64 distinct FLAG values.  Block mode
    keeps at most CBC_CAP_FLAG distinct values per block (the packer cuts on the 65th); every block == the oracle."""
    flags = tuple(sorted({f | x for f in (65, 81, 83, 97, 99, 113, 129, 145, 147, 161, 163, 177) for x in (0, 256, 1024, 2048, 512, 1280, 2304, 3072, 768)}))
    assert len(flags) >= 100
    fa, sam, _, _ = synth.dataset(61, [400000], [9000], 100, sub_rate=0.005, indel_frac=0.05, flags=flags)
    pb = host.pack_sam(sam, fa, block_reads=4096)
    assert len({int(f) for f in pb.recs["flag"]}) >= 100 and pb.n_blocks > 3
    for b in range(pb.n_blocks):
        r0, n = int(pb.blocks[b]["rec_base"]), int(pb.blocks[b]["n_reads"])
        assert len({int(f) for f in pb.recs["flag"][r0:r0 + n]}) <= 64
    _check_blocks(enc, pb, sam)
    _gpu_roundtrip(enc, pb, sam, fa)


def test_checksum_kernel_equals_host_twin(built):
    """cbc_gpu_checksum_device == cbc_checksum64 (what the two sides of the bitstream gather compute), on aligned and
    unaligned device ranges, empty input included.  In a child process: torch owns the device memory there and has to
    initialise HIP before the library does (as in bench.py), which this process's Encoder fixture has already done."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, ctypes
        sys.path.insert(0, %r)
        import numpy as np, torch
        torch.cuda.init()
        from cbc_amd import gpu, host
        enc = gpu.Encoder(0)
        rng = np.random.default_rng(3)
        a = rng.integers(0, 256, 5_000_011, dtype=np.uint8)
        t = torch.from_numpy(a).cuda()
        d_sum = torch.zeros(1, dtype=torch.int64, device="cuda")
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for off, n in ((0, len(a)), (0, 4096), (1, 100_000), (7, 17), (16, 0), (3, 1)):
            enc.checksum_device(t.data_ptr() + off, n, d_sum.data_ptr(), s)
            torch.cuda.synchronize()
            assert int(d_sum.item()) & (2 ** 64 - 1) == host.checksum64(a[off:off + n]), (off, n)
        print("CHECKSUM_OK")
    """ % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "CHECKSUM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_cli_rccl_exchange_one_member_self_test(built, tmp_path):
    """`cbc --devices 0 --rccl`: the multi-device path with its RCCL exchange (cbc_gpu_group_gather: grouped ncclSend /
    ncclRecv, checksums taken on the sending and the receiving device) on the one GPU of this box -- a one-member group sends
    to itself, which runs every call site.  The container equals the plain one-device container; with two contexts on one
    device (`--devices 0,0`) no RCCL group can exist and the bitstreams come back over PCIe instead (same container).
    Real multi-device RCCL runs only on the driver's 8-GPU node."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cbc_amd", "csrc", "cbc")
    fa, sam, _, _ = synth.dataset(19, [60000, 200000, 90000], [900, 3000, 1400], 100, sub_rate=0.01, indel_frac=0.2)
    (tmp_path / "in.sam").write_bytes(sam); (tmp_path / "ref.fa").write_bytes(fa)
    outs = {}
    for tag, extra in (("plain", []), ("rccl", ["--devices", "0", "--rccl"]), ("dup", ["--devices", "0,0"])):
        o = tmp_path / ("out_%s.cbc" % tag)
        r = subprocess.run([exe, "-c", "1", str(tmp_path / "in.sam"), str(o), str(tmp_path / "ref.fa"), "--block-reads", "256", "--verbose"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = (o.read_bytes(), r.stdout)
    assert outs["plain"][0] == outs["rccl"][0] == outs["dup"][0]
    assert "exchange: RCCL" in outs["rccl"][1] and "sent == received" in outs["rccl"][1]
    assert "exchange: one D2H per device" in outs["dup"][1]
