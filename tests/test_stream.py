"""The whole-file stream ("compat" mode): the reference's own file format, one arithmetic stream per file, every
model in its general (rescaling) form (cbc_amd/csrc/cbc_stream_body.h).  CPU part: the packer's whole-file mode and
the kernel body on the lock-step emulation against the oracle's whole-file encode / the SEQ column.  GPU part
(-m gpu): the same through the C ABI and the `cbc --compat` command line."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

import blockref
import synth
from cbc_amd import gpu, host
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _seq_column(sam):
    return [ln.split(b"\t")[9] for ln in sam.splitlines() if ln and not ln.startswith(b"@") and not (int(ln.split(b"\t")[1]) & 4)]


def _record_lines(sam):
    return [ln for ln in sam.splitlines(keepends=True) if ln.strip() and not ln.startswith(b"@")]


def _snps_per_reference_base(sam):
    """How many SNP symbols each chars row (reference base) receives: the MD letters of the file."""
    import re
    n = {}
    for ln in _record_lines(sam):
        f = ln.rstrip(b"\n").split(b"\t")
        md = [x for x in f[11:] if x.startswith(b"MD:Z:")][0][5:]
        for m in re.finditer(rb"(\^[A-Z]+)|([A-Z])", md):
            if m.group(2):
                n[m.group(2)] = n.get(m.group(2), 0) + 1
    return n


# rescale-crossing inputs: (synth arguments, what must have been rescaled inside the one stream)
RESCALE_CASES = {
    "per_record_models": (31, 1_500_000, 140_000, 100, 0.003, 0.02),    # flag at 122 880 records; same_ref, rlength, pos near 104 858
    "chars_and_var": (32, 400_000, 15_000, 100, 0.45, 0.3),             # a chars row at ~131 k SNPs, the var context (L+2, 0) at ~104 k
}


def _rescale_input(name):
    pb0, sam, fa = host.synth(*RESCALE_CASES[name], want_text=True)
    pb0.close()
    if name == "per_record_models":
        assert len(_seq_column(sam)) > 122_880 + 1000
    else:
        per_row = _snps_per_reference_base(sam)
        assert max(per_row.values()) > 131_072 + 2000, per_row          # (2^20 - 41) / 8 symbols rescale a chars row
    return sam, fa


def test_packer_whole_file_mode(built):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000, 50000], [400, 150, 70], 100, sub_rate=0.02, indel_frac=0.3)
    pb = host.pack_sam(sam, fa, whole_file=True)
    blk = host.pack_sam(sam, fa, block_reads=100)
    assert pb.whole_file and not blk.whole_file
    assert pb.n_blocks == 3 and list(pb.blocks["n_reads"]) == [400, 150, 70]         # one segment per contig
    want_pos = [int(ln.split(b"\t")[3]) for ln in _record_lines(sam)]
    assert list(pb.recs["pos"]) == want_pos                                          # POS as in the SAM, not rebased
    assert [int(x) for x in pb.blocks["ref_off"]] == [int(c["ref_off"]) for c in pb.contigs]
    assert pb.seq.tobytes() == blk.seq.tobytes() and pb.tok.tobytes() == blk.tok.tobytes()
    # the threaded text path cuts the same segments
    pt = host.pack_sam(sam, fa, whole_file=True, threads=4)
    assert pt.blocks.tobytes() == pb.blocks.tobytes() and pt.recs.tobytes() == pb.recs.tobytes()
    # what the reference cannot represent as one stream is refused, with the way out named
    fa2, sam2, _, _ = synth.dataset(6, [6_000_000], [30], 100, sub_rate=0.0, indel_frac=0.0)
    lines = _record_lines(sam2)
    first = lines[0].split(b"\t"); first[3] = b"5200000"
    seq = b"".join(fa2.split(b"\n")[1:])[5200000 - 1:5200000 - 1 + 100]
    first[9] = seq
    only_far = b"\t".join(first)
    with pytest.raises(host.CbcInputError, match="MAX_ALPHA"):
        host.pack_sam(only_far + only_far, fa2, whole_file=True)
    host.pack_sam(only_far + only_far, fa2)                                         # block mode rebases: fine


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_L*.json"))))
def test_stream_body_on_golden_vectors(built, path):
    """Every committed vector's whole-file stream (oracle-generated, tests/golden/make_golden.py), two contigs included."""
    g = json.load(open(path))
    sam, fa = g["sam"].encode(), g["fasta"].encode()
    pb = host.pack_sam(sam, fa, whole_file=True)
    payloads, res = blockref.emu_encode_stream(pb)
    assert int(res[0]["status"]) == 0 and payloads[0].hex() == g["stream_hex"]
    recs, bases, dres = blockref.emu_decode_stream(payloads[0], pb.ref, pb.contigs, pb.n_recs + 3)
    assert int(dres["status"]) == 0 and len(recs) == g["n_reads"]
    want = _seq_column(sam)
    assert all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want))
    assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all()


@pytest.mark.parametrize("case", sorted(RESCALE_CASES))
def test_stream_body_across_model_rescales(built, case):
    """The general form of the models where it matters: totals cross 2^20 inside the stream (the inputs are checked
    to be large enough, so this cannot silently stop covering the rescale code).  Encode bytes == oracle, decode
    of the oracle's bytes == the reads."""
    sam, fa = _rescale_input(case)
    pb = host.pack_sam(sam, fa, whole_file=True)
    expect, st = oracle.encode(sam, fa, return_stats=True)
    payloads, res = blockref.emu_encode_stream(pb)
    assert int(res[0]["status"]) == 0 and payloads[0] == expect and int(res[0]["n_symbols"]) == st.n_symbols
    recs, bases, dres = blockref.emu_decode_stream(expect, pb.ref, pb.contigs, pb.n_recs + 3)
    want = _seq_column(sam)
    assert int(dres["status"]) == 0 and len(recs) == len(want) and int(dres["n_symbols"]) == st.n_symbols
    assert all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want))


def test_stream_body_three_contigs_and_sparse_positions(built):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000, 50000], [4000, 1500, 700], 100, sub_rate=0.02, indel_frac=0.5,
                                  trailing_s_frac=0.2, dup_pos_frac=0.1)
    pb = host.pack_sam(sam, fa, whole_file=True)
    payloads, res = blockref.emu_encode_stream(pb)
    assert payloads[0] == oracle.encode(sam, fa)
    pb0, sam, fa = host.synth(33, 30_000_000, 20_000, 150, want_text=True)             # ~4700 distinct POS steps
    pb = host.pack_sam(sam, fa, whole_file=True)
    assert pb.cap_pos > 4000
    payloads, res = blockref.emu_encode_stream(pb)
    assert payloads[0] == oracle.encode(sam, fa)
    # a decode buffer that is too small is reported, not overrun
    recs, bases, dres = blockref.emu_decode_stream(payloads[0], pb.ref, pb.contigs, 1000)
    assert int(dres["status"]) == 1 and int(dres["nbytes"]) == 1000


def test_general_form_over_blocks(built):
    """The same body, one stream per block: equals the block kernels' payloads on ordinary blocks, and codes a block
    of more than CBC_MAX_BLOCK_READS records (which the block kernels refuse) exactly as the oracle does."""
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [2500, 900], 100, sub_rate=0.02, indel_frac=0.4)
    pb = host.pack_sam(sam, fa, block_reads=700)
    payloads, res = blockref.emu_encode_stream(pb, per_segment=True)
    ep, eres = blockref.emu_encode(pb)
    assert (res["status"] == 0).all() and payloads == ep and (res["n_symbols"] == eres["n_symbols"]).all()
    pb0, sam, fa = host.synth(41, 2_000_000, 40_000, 100, want_text=True)
    big = host.pack_sam(sam, fa, whole_file=True)                # one contig = one segment of 40 000 records
    assert big.n_blocks == 1 and int(big.blocks[0]["n_reads"]) == 40_000
    _, refused = blockref.emu_encode(big)
    assert int(refused[0]["status"]) == 7                        # CBC_ST_UNSUPPORTED from the block kernel body
    payloads, res = blockref.emu_encode_stream(big, per_segment=True)
    assert int(res[0]["status"]) == 0 and payloads[0] == oracle.encode(sam, fa)


# ----------------------------------------------------------------------------------------- on the GPU
@pytest.fixture(scope="module")
def enc():
    e = gpu.Encoder(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_L*.json"))))
def test_gpu_stream_on_golden_vectors(enc, built, path):
    g = json.load(open(path))
    sam, fa = g["sam"].encode(), g["fasta"].encode()
    pb = host.pack_sam(sam, fa, whole_file=True)
    enc.upload_reference(pb.ref)
    stream, sr = enc.encode_stream(pb)
    assert sr.status == 0 and stream.hex() == g["stream_hex"]
    recs, bases, dr = enc.decode_stream(stream, pb.contigs)
    want = _seq_column(sam)
    assert dr.status == 0 and len(recs) == len(want)
    assert all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want))
    assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(RESCALE_CASES))
def test_gpu_stream_across_model_rescales(enc, built, case):
    """In-stream rescales of every model family on the HIP path: bytes == oracle's whole-file encode (what the
    reference's own -x reads), GPU decode == the reads, symbol counts equal."""
    sam, fa = _rescale_input(case)
    pb = host.pack_sam(sam, fa, whole_file=True)
    expect, st = oracle.encode(sam, fa, return_stats=True)
    enc.upload_reference(pb.ref)
    stream, sr = enc.encode_stream(pb)
    assert sr.status == 0 and stream == expect and sr.n_symbols == st.n_symbols
    recs, bases, dr = enc.decode_stream(stream, pb.contigs, rec_cap=1000)       # starts too small: grows on OUT_FULL
    want = _seq_column(sam)
    assert dr.status == 0 and len(recs) == len(want) and dr.n_symbols == st.n_symbols
    assert all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want))


@pytest.mark.gpu
def test_gpu_stream_contigs_and_fallback(enc, built):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000, 50000], [4000, 1500, 700], 100, sub_rate=0.02, indel_frac=0.5,
                                  trailing_s_frac=0.2, dup_pos_frac=0.1)
    pb = host.pack_sam(sam, fa, whole_file=True)
    enc.upload_reference(pb.ref)
    stream, sr = enc.encode_stream(pb)
    assert stream == oracle.encode(sam, fa)
    text, nr = oracle.decode(stream, fa)                          # the oracle's decoder reads the GPU's file
    assert nr == 6200 and text == b"".join(w + b"\n" for w in _seq_column(sam))
    # general-form coder over ordinary blocks == the block kernel; over a 40 000-record block == the oracle
    blk = host.pack_sam(sam, fa, block_reads=700)
    enc.upload_reference(blk.ref)
    p_gen, r_gen = enc.encode_stream_blocks(blk)
    p_blk, r_blk, _, _ = enc.encode_blocks(blk)
    assert (r_gen["status"] == 0).all() and p_gen == p_blk and (r_gen["n_symbols"] == r_blk["n_symbols"]).all()
    pb0, sam, fa = host.synth(41, 2_000_000, 40_000, 100, want_text=True)
    big = host.pack_sam(sam, fa, whole_file=True)
    enc.upload_reference(big.ref)
    p_blk, r_blk, _, _ = enc.encode_blocks(big)
    assert int(r_blk[0]["status"]) == 7
    p_gen, r_gen = enc.encode_stream_blocks(big)
    assert int(r_gen[0]["status"]) == 0 and p_gen[0] == oracle.encode(sam, fa)
    # ... and the decode twin of the general form: such a block back to its reads (the block decoder refuses it), and the
    # ordinary blocks' general-form payloads too
    recs, bases, dres = enc.decode_stream_blocks(p_gen, big)
    want = _seq_column(sam)
    assert int(dres[0]["status"]) == 0 and int(dres[0]["nbytes"]) == 40_000 and int(dres[0]["n_symbols"]) == int(r_gen[0]["n_symbols"])
    assert all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want)) and (recs["pos"] == big.recs["pos"]).all()
    text, nr = oracle.decode(p_gen[0], fa)
    assert nr == 40_000 and text == b"".join(w + b"\n" for w in want)
    fa5, sam5, _, _ = synth.dataset(5, [300000, 120000, 50000], [4000, 1500, 700], 100, sub_rate=0.02, indel_frac=0.5,
                                    trailing_s_frac=0.2, dup_pos_frac=0.1)
    enc.upload_reference(blk.ref)
    p5, r5 = enc.encode_stream_blocks(blk)
    recs, bases, dres = enc.decode_stream_blocks(p5, blk)
    assert (dres["status"] == 0).all() and (recs["flag"] == blk.recs["flag"]).all() and (recs["pos"] == blk.recs["pos"]).all()
    want5 = _seq_column(sam5)
    assert len(want5) == blk.n_recs and all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want5))


@pytest.mark.gpu
def test_cli_compat_round_trip(built, tmp_path):
    """`cbc -c --compat` writes the reference's own format: the file equals the oracle's whole-file encode byte for
    byte; `cbc -d` recognises it (no container magic) and returns the SEQ column."""
    exe = os.path.join(ROOT, "cbc_amd", "csrc", "cbc")
    fa, sam, _, _ = synth.dataset(8, [200000, 80000], [3000, 1000], 100, sub_rate=0.01, indel_frac=0.2)
    (tmp_path / "in.sam").write_bytes(sam); (tmp_path / "ref.fa").write_bytes(fa)
    r = subprocess.run([exe, "-c", "1", "--compat", str(tmp_path / "in.sam"), str(tmp_path / "out.cbc"), str(tmp_path / "ref.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "out.cbc").read_bytes() == oracle.encode(sam, fa)
    r = subprocess.run([exe, "-x", str(tmp_path / "out.cbc"), str(tmp_path / "reads.txt"), str(tmp_path / "ref.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "reads.txt").read_bytes() == b"".join(w + b"\n" for w in _seq_column(sam))


# ------------------------------------------------------------------ what the reference's tables hold and no register / LDS budget does
PAIRED_FLAGS = tuple(sorted({f | x for f in (65, 81, 83, 97, 99, 113, 129, 145, 147, 161, 163, 177)
                             for x in (0, 256, 1024, 2048, 512, 1280, 2304, 3072, 768)}))


def _with_many_flags(sam, n_first=0):
    """Rewrite the FLAG column with the 108 paired-end / secondary / supplementary / duplicate combinations above (no flag
    with bit 4 = unmapped).  The first n_first records keep the generator's two flags so that the register pairs fill
    with the common values first, as in a real file."""
    out = []
    k = 0
    for ln in sam.splitlines(keepends=True):
        if ln.startswith(b"@") or not ln.strip():
            out.append(ln); continue
        f = ln.split(b"\t")
        if k >= n_first:
            f[1] = str(PAIRED_FLAGS[(k * 2654435761 >> 7) % len(PAIRED_FLAGS)]).encode()
        k += 1
        out.append(b"\t".join(f))
    return b"".join(out)


def _many_flags_input():
    """> 122 880 records (the flag model's first rescale, sam_models.c:96-130 + stream_model.c:41-48) with 110 FLAG values."""
    pb0, sam, fa = host.synth(35, 1_200_000, 126_000, 100, 0.003, 0.02, want_text=True)
    pb0.close()
    sam = _with_many_flags(sam, n_first=500)
    assert len(_seq_column(sam)) > 122_880 + 1000
    return sam, fa


def _many_pos_steps_input():
    """> 104 858 records (the pos model's first rescale) whose POS steps take > 10 000 distinct values: more than the
    8192 alphabet entries the stream kernels keep in LDS."""
    pb0, sam, fa = host.synth(36, 290_000_000, 112_000, 100, 0.003, 0.02, want_text=True)
    pb0.close()
    return sam, fa


def test_whole_file_stream_takes_what_the_reference_takes(built):
    """More than CBC_CAP_FLAG distinct FLAG values and more distinct POS steps than the LDS alphabet holds, each across the
    model's rescale: the packer accepts them in whole-file mode (it used to refuse) and the stream body == the oracle,
    encode and decode (emulation of the kernel body)."""
    assert len(PAIRED_FLAGS) >= 100
    sam, fa = _many_flags_input()
    pb = host.pack_sam(sam, fa, whole_file=True)
    assert len({int(f) for f in pb.recs["flag"]}) >= 100
    expect, st = oracle.encode(sam, fa, return_stats=True)
    payloads, res = blockref.emu_encode_stream(pb)
    assert int(res[0]["status"]) == 0 and payloads[0] == expect and int(res[0]["n_symbols"]) == st.n_symbols
    recs, bases, dres = blockref.emu_decode_stream(expect, pb.ref, pb.contigs, pb.n_recs + 3)
    want = _seq_column(sam)
    assert int(dres["status"]) == 0 and len(recs) == len(want) and int(dres["n_symbols"]) == st.n_symbols
    assert all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want)) and (recs["flag"] == pb.recs["flag"]).all()
    pb.close()

    sam, fa = _many_pos_steps_input()
    pb = host.pack_sam(sam, fa, whole_file=True)
    pos = pb.recs["pos"].astype(np.int64)
    n_steps = len(set(np.diff(np.concatenate([[0], pos])).tolist()))
    assert n_steps > 10_000 and pb.cap_pos > blockref.STREAM_POS_LDS + 1500 and pb.n_recs > 104_858 + 1000, (n_steps, pb.cap_pos)
    expect, st = oracle.encode(sam, fa, return_stats=True)
    payloads, res = blockref.emu_encode_stream(pb)
    assert int(res[0]["status"]) == 0 and payloads[0] == expect and int(res[0]["n_symbols"]) == st.n_symbols
    recs, bases, dres = blockref.emu_decode_stream(expect, pb.ref, pb.contigs, pb.n_recs + 3)
    want = _seq_column(sam)
    assert int(dres["status"]) == 0 and len(recs) == len(want) and int(dres["n_symbols"]) == st.n_symbols
    assert (recs["pos"] == pb.recs["pos"]).all() and all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want))


@pytest.mark.gpu
def test_gpu_whole_file_stream_takes_what_the_reference_takes(enc, built):
    """The same two inputs on the HIP path: `cbc_gpu_encode_stream` bytes == the oracle's whole-file encode, the oracle's
    decoder reads the GPU's file, and `cbc_gpu_decode_stream` returns the reads."""
    for make in (_many_flags_input, _many_pos_steps_input):
        sam, fa = make()
        pb = host.pack_sam(sam, fa, whole_file=True)
        expect, st = oracle.encode(sam, fa, return_stats=True)
        enc.upload_reference(pb.ref)
        stream, sr = enc.encode_stream(pb)
        assert sr.status == 0 and stream == expect and sr.n_symbols == st.n_symbols
        recs, bases, dr = enc.decode_stream(stream, pb.contigs)
        want = _seq_column(sam)
        assert dr.status == 0 and len(recs) == len(want) and dr.n_symbols == st.n_symbols
        assert (recs["pos"] == pb.recs["pos"]).all() and (recs["flag"] == pb.recs["flag"]).all()
        assert all(bases[i, :len(w)].tobytes() == w for i, w in enumerate(want))
        pb.close()
