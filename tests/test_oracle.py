"""The oracle (CPU restatement of the reference) against everything that can pin it here:
the known-answer prefix recorded from the real reference, the committed regression vectors,
and encode -> decode round trips over every edit class the reference handles."""
import glob
import json
import os

import pytest

import synth
from cbc_amd import host
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _seqs(rbc):
    return b"".join(r["seq"] + b"\n" for c in rbc for r in c[2])


def test_survey_known_answer_prefix():
    kat = json.load(open(os.path.join(GOLDEN, "survey_kat.json")))
    fa, sam, rbc, _ = synth.dataset(1, [200000], [4], kat["read_length"], sub_rate=0.0, indel_frac=0.0)
    out = oracle.encode(sam, fa)
    assert out[:12].hex(" ") == kat["stream_prefix_hex"]
    assert len(out) == kat["file_bytes_for_4_reads"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_L*.json"))))
def test_golden_regression_vectors(path):
    g = json.load(open(path))
    sam, fa = g["sam"].encode(), g["fasta"].encode()
    stream = oracle.encode(sam, fa)
    assert stream.hex() == g["stream_hex"]
    text, nr = oracle.decode(stream, fa)
    assert nr == g["n_reads"]
    expect = b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines() if not ln.startswith(b"@"))
    assert text == expect


@pytest.mark.parametrize("kw", [
    dict(sub_rate=0.0, indel_frac=0.0),
    dict(sub_rate=0.01, indel_frac=0.0),
    dict(sub_rate=0.01, indel_frac=0.5),
    dict(sub_rate=0.02, indel_frac=0.5, trailing_s_frac=0.3, dup_pos_frac=0.1),
    dict(sub_rate=0.05, indel_frac=0.0),
])
@pytest.mark.parametrize("L", [100, 150])
def test_round_trip(kw, L):
    fa, sam, rbc, _ = synth.dataset(7, [200000, 90000], [3000, 1000], L, **kw)
    stream, st = oracle.encode(sam, fa, return_stats=True)
    assert st.n_records == 4000 and st.read_length == L
    text, nr = oracle.decode(stream, fa)
    assert nr == 4000 and text == _seqs(rbc)


def test_round_trip_across_model_rescales():
    """> 122 880 records: the flag model (n = 65536 + 8/record) and the same_ref / rlength / pos
    models (10/record) all cross the 2^20 rescale inside one stream."""
    pb, sam, fa = host.synth(31, 1_500_000, 140_000, 100, 0.003, 0.02, want_text=True)
    stream, st = oracle.encode(sam, fa, return_stats=True)
    assert st.n_records == 140_000
    text, nr = oracle.decode(stream, fa)
    expect = b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines())
    assert nr == 140_000 and text == expect
    pb.close()


def test_unmapped_records_are_dropped():
    fa, sam, rbc, _ = synth.dataset(8, [100000], [50], 100)
    lines = sam.splitlines(keepends=True)
    hdr = [l for l in lines if l.startswith(b"@")]
    recs = [l for l in lines if not l.startswith(b"@")]
    um = recs[10].split(b"\t")
    um[1] = b"4"
    with_um = b"".join(hdr + recs[:20] + [b"\t".join(um)] + recs[20:])
    assert oracle.encode(with_um, fa) == oracle.encode(sam, fa)


def test_md_last_column_quirk_changes_stream():
    """Quirk Q2: MD as the last column keeps its newline -> a phantom N->N SNP on imperfect reads."""
    rng_fa, sam, rbc, contigs = synth.dataset(9, [100000], [200], 100, sub_rate=0.02, indel_frac=0.0)
    sam_last = synth.sam_text(rbc, md_last=True)
    a, sa = oracle.encode(sam, rng_fa, return_stats=True)
    b, sb = oracle.encode(sam_last, rng_fa, return_stats=True)
    n_imperfect = sum(1 for r in rbc[0][2] if r["nm"] > 0)
    assert sb.n_symbols == sa.n_symbols + 2 * n_imperfect      # one var + one chars symbol each
    assert a != b


def test_leading_soft_clip_alone_round_trips():
    """Quirk Q6: a leading S with no other edit goes through the reference's in-place MD rebuild."""
    import numpy as np
    rng = np.random.default_rng(3)
    contig = synth.make_contig(rng, 50000)
    recs = []
    pos = 100
    for i in range(300):
        pos += int(rng.integers(1, 40))
        L = 100
        if i % 3 == 0:
            k = int(rng.integers(1, 6))
            body = contig[pos - 1: pos - 1 + L - k]
            clip = synth._ACGT[rng.integers(0, 4, size=k)]
            seq = np.concatenate([clip, body]).tobytes()
            recs.append(dict(pos=pos, flag=0 if i % 2 else 16, cigar="%dS%dM" % (k, L - k), seq=seq, md=str(L - k), nm=0))
        else:
            recs.append(dict(pos=pos, flag=0 if i % 2 else 16, cigar="%dM" % L, seq=contig[pos - 1:pos - 1 + L].tobytes(),
                             md=str(L), nm=0))
    rbc = [("chrS", 50000, recs)]
    fa = synth.fasta_text([("chrS", contig)])
    sam = synth.sam_text(rbc)
    stream = oracle.encode(sam, fa)
    text, nr = oracle.decode(stream, fa)
    assert nr == 300 and text == _seqs(rbc)


# ---- the packed-input CPU port (oracle/cbc_cpu.c: the cbc_cpu_* set with the C ABI's signatures) is pinned to
# ---- the text path above: same bytes and same coder-step counts for every block
def _cpu_port_equals_text_path(pb, sam):
    import blockref
    payloads, res = oracle.cpu_encode_blocks(pb, return_payloads=True)
    assert (res["status"] == 0).all()
    lines = blockref.mapped_sam_lines(sam)
    for b in range(pb.n_blocks):
        bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
        exp, st = oracle.encode(bsam, bfa, return_stats=True)
        assert payloads[b] == exp, "block %d" % b
        assert int(res[b]["n_symbols"]) == st.n_symbols


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_L*.json"))))
def test_cpu_port_on_golden_vectors(built, path):
    g = json.load(open(path))
    pb = host.pack_sam(g["sam"].encode(), g["fasta"].encode(), block_reads=g["block_reads"])
    payloads, res = oracle.cpu_encode_blocks(pb, return_payloads=True)
    assert [p.hex() for p in payloads] == g["block_payload_hex"]


@pytest.mark.parametrize("kw,L,br", [
    (dict(), 150, 1024),
    (dict(sub_rate=0.02, indel_frac=0.5, trailing_s_frac=0.2, dup_pos_frac=0.1), 100, 512),
    (dict(flags=(0, 16, 83, 99, 147, 163)), 150, 2048),
    (dict(sub_rate=0.45, indel_frac=0.3), 150, 256),
])
def test_cpu_port_equals_text_path(built, kw, L, br):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [2500, 900], L, **kw)
    _cpu_port_equals_text_path(host.pack_sam(sam, fa, block_reads=br), sam)


def test_cpu_port_soft_clips_and_subset(built):
    from test_emu_parity import _soft_clip_sam
    fa, sam = _soft_clip_sam(3)
    pb = host.pack_sam(sam, fa, block_reads=256)
    _cpu_port_equals_text_path(pb, sam)
    allp, _ = oracle.cpu_encode_blocks(pb, return_payloads=True)
    some, _ = oracle.cpu_encode_blocks(pb, blocks=[2, 0], return_payloads=True)
    assert some == [allp[2], allp[0]]
