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
