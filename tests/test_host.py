"""Host logic without a GPU: the packer's parsing rules and limits, block cutting, the container,
and that libcbc_gpu.so loads and exports every symbol include/cbc_gpu.h declares."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

import synth
from cbc_amd import gpu, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tokens(pb, i):
    r = pb.recs[i]
    b = next(k for k in range(pb.n_blocks) if pb.blocks[k]["rec_base"] <= i < pb.blocks[k]["rec_base"] + pb.blocks[k]["n_reads"])
    base = int(pb.blocks[b]["tok_base"]) + int(r["tok_off"])
    hdr = int(pb.tok[base])
    nc, nm = hdr & 0xffff, hdr >> 16
    cig = [(int(t) >> 4, int(t) & 15) for t in pb.tok[base + 2: base + 2 + nc]]
    md = [(int(t) >> 8, chr(int(t) & 0xff)) for t in pb.tok[base + 2 + nc: base + 2 + nc + nm]]
    return cig, md


def _one(fields, fa_seq="ACGT" * 100):
    """SAM with two identical-length records so the header read length is well defined."""
    fa = (">c\n%s\n" % fa_seq).encode()
    lines = []
    for f in fields:
        lines.append("\t".join(str(x) for x in f) + "\n")
    return "".join(lines).encode(), fa


def test_gpu_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "cbc_gpu.h")).read()
    declared = sorted(set(re.findall(r"\b(cbc_(?:gpu|stream)_\w+)\s*\(", hdr)))
    assert len(declared) >= 12
    assert sorted(gpu.EXPORTS) == declared
    L = ctypes.CDLL(gpu.GPU_LIB)                         # loads without a GPU; no compute call made
    for name in declared:
        assert getattr(L, name) is not None
    assert gpu.lib().cbc_gpu_abi_version() == 1


def test_host_library_exports(built):
    hdr = open(os.path.join(ROOT, "include", "cbc_host.h")).read()
    declared = sorted(set(re.findall(r"\b(cbc_(?:pack|packed|synth|free|container)\w*)\s*\(", hdr)))
    L = ctypes.CDLL(host.HOST_LIB)
    for name in declared:
        assert getattr(L, name) is not None


def test_no_gpu_means_loud_failure(built):
    """The product has no CPU fallback: without a device the encoder refuses to exist."""
    if gpu.lib().cbc_gpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(gpu.CbcGpuError):
        gpu.Encoder(0)


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(host.ReadRec) == 16 and ctypes.sizeof(host.BlockDesc) == 64
    assert ctypes.sizeof(host.BlockResult) == 16 and ctypes.sizeof(gpu.DeviceBatch) == ctypes.sizeof(ctypes.c_void_p) * 8 + 8 * 5 + 8 + 8


def test_rebase_and_block_cut(built):
    fa, sam, rbc, _ = synth.dataset(2, [300000, 100000], [2500, 700], 100)
    pb = host.pack_sam(sam, fa, block_reads=1000)
    assert pb.n_recs == 3200 and pb.read_length == 100
    assert [int(x) for x in pb.info["n_reads"]] == [1000, 1000, 500, 700]
    assert [int(x) for x in pb.info["contig"]] == [0, 0, 0, 1]
    recs = [r for c in rbc for r in c[2]]
    k = 0
    for b in range(pb.n_blocks):
        w0 = int(pb.info[b]["window_start"])
        first = int(pb.blocks[b]["rec_base"])
        assert int(pb.recs[first]["pos"]) == 1 and w0 == recs[first]["pos"] - 1
        c = pb.contigs[int(pb.info[b]["contig"])]
        assert int(pb.blocks[b]["ref_off"]) == int(c["ref_off"]) + w0
        for j in range(int(pb.blocks[b]["n_reads"])):
            assert int(pb.recs[first + j]["pos"]) + w0 == recs[first + j]["pos"]
            k += 1
    assert k == 3200
    assert pb.contig_name(0) == b"chr1" and pb.contig_name(1) == b"chr2"
    # reference: upper-cased, each contig followed by CBC_REF_PAD zero bytes
    c0 = pb.contigs[0]
    assert int(c0["length"]) == 300000 and not pb.ref[int(c0["length"]): int(c0["length"]) + host.CBC_REF_PAD].any()


def test_cigar_and_md_tokens_follow_the_reference_scanners(built):
    seq = "A" * 50
    sam, fa = _one([
        ["r0", 0, "c", 1, 60, "50M", "*", 0, 0, seq, "I" * 50, "MD:Z:50", "NM:i:0"],
        ["r1", 16, "c", 3, 60, "10M2I38M", "*", 0, 0, seq, "I" * 50, "MD:Z:5C10^AC0T31", "NM:i:5"],
        # ops the reference ignores (H, N, =, X) do not consume the number: the next recognised op
        # takes atoi() of the whole unconsumed segment, i.e. the FIRST number in it
        ["r2", 0, "c", 5, 60, "5H45M5S", "*", 0, 0, seq, "I" * 50, "NM:i:0", "MD:Z:45\n".strip("\n")],
        # MD as the last column keeps its newline as a letter token (quirk Q2)
        ["r3", 0, "c", 7, 60, "50M", "*", 0, 0, seq, "I" * 50, "NM:i:1", "MD:Z:20G29"],
    ])
    pb = host.pack_sam(sam, fa)
    assert _tokens(pb, 0) == ([(50, 0)], [])
    assert _tokens(pb, 1) == ([(10, 0), (2, 1), (38, 0)], [(5, "C"), (10, "T")])   # 10 + 0 around ^AC
    assert _tokens(pb, 2)[0] == [(5, 0), (5, 3)]                                   # "5H45M" -> atoi = 5
    assert _tokens(pb, 3)[1] == [(20, "G"), (29, "\n")]


def test_record_without_md_keeps_previous_md(built):
    """load_sam_line only strcpy()s edits when an MD/XD field is present (sam_file_allocation.c:507-511)."""
    seq = "A" * 40
    sam, fa = _one([
        ["r0", 0, "c", 1, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:7C32", "NM:i:1"],
        ["r1", 0, "c", 2, 60, "40M", "*", 0, 0, seq, "I" * 40, "NM:i:0"],
    ])
    pb = host.pack_sam(sam, fa)
    assert _tokens(pb, 1)[1] == [(7, "C")]


def test_unmapped_skipped_and_headers_ignored(built):
    seq = "A" * 40
    sam, fa = _one([
        ["r0", 0, "c", 1, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"],
        ["u", 4, "*", 0, 0, "*", "*", 0, 0, seq, "I" * 40],
        ["r1", 0, "c", 2, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"],
    ])
    pb = host.pack_sam(b"@HD\tVN:1.6\n@SQ\tSN:c\tLN:400\n" + sam, fa)
    assert pb.n_recs == 2 and pb.n_skipped_unmapped == 1


@pytest.mark.parametrize("mutate,msg", [
    (lambda f: f.__setitem__(3, 0), "POS"),
    (lambda f: f.__setitem__(5, "*"), "CIGAR '*'"),
    (lambda f: f.__setitem__(9, "A" * 253), "read length"),
])
def test_inputs_outside_the_reference_limits_are_rejected(built, mutate, msg):
    seq = "A" * 40
    good = ["r0", 0, "c", 1, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"]
    bad = list(good)
    bad[3] = 2
    mutate(bad)
    sam, fa = _one([good, bad], fa_seq="ACGT" * 200)
    with pytest.raises(host.CbcInputError) as e:
        host.pack_sam(sam, fa)
    assert msg in str(e.value)


def test_unsorted_and_missing_contig_rejected(built):
    seq = "A" * 40
    n0 = ["r0", 0, "n" * 200, 10, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"]
    sam, fa = _one([n0, n0])
    with pytest.raises(host.CbcInputError, match="longer than"):
        host.pack_sam(sam, fa)
    a = ["r0", 0, "c", 10, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"]
    b = ["r1", 0, "c", 5, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"]
    sam, fa = _one([a, b])
    with pytest.raises(host.CbcInputError, match="not sorted"):
        host.pack_sam(sam, fa)
    c = ["r1", 0, "d", 5, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"]
    sam, fa = _one([a, c])
    with pytest.raises(host.CbcInputError, match="FASTA has fewer"):
        host.pack_sam(sam, fa)
    long_line = ["r0", 0, "c", 10, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40", "XX:Z:" + "y" * 1100]
    sam, fa = _one([a, long_line])
    with pytest.raises(host.CbcInputError, match="1023"):
        host.pack_sam(sam, fa)


def test_fasta_is_consumed_in_order_and_upper_cased(built):
    fa = b">first\nacgtn\nACGT\n>second\nGGGG\nCC\n"
    seq = "ACGTN"
    sam = ("r0\t0\tzzz\t1\t60\t5M\t*\t0\t0\t%s\tIIIII\tMD:Z:5\n" % seq +
           "r1\t0\tzzz\t1\t60\t5M\t*\t0\t0\t%s\tIIIII\tMD:Z:5\n" % seq +
           "r2\t0\tother\t1\t60\t5M\t*\t0\t0\tGGGGC\tIIIII\tMD:Z:5\n")
    # read length 5 < 30 is fine for the packer (only the generator has a lower bound)
    pb = host.pack_sam(sam.encode(), fa)
    c0, c1 = pb.contigs[0], pb.contigs[1]
    assert pb.ref[int(c0["ref_off"]): int(c0["ref_off"]) + 9].tobytes() == b"ACGTNACGT"
    assert pb.ref[int(c1["ref_off"]): int(c1["ref_off"]) + 6].tobytes() == b"GGGGCC"
    assert pb.contig_name(0) == b"zzz" and pb.contig_name(1) == b"other"      # FASTA header text is ignored


def test_header_read_length_rule(built):
    """get_read_length: SEQ length of the SECOND record; -l takes the maximum."""
    fa = (">c\n" + "A" * 400 + "\n").encode()
    mk = lambda n, p: "r\t0\tc\t%d\t60\t%dM\t*\t0\t0\t%s\t%s\tMD:Z:%d\n" % (p, n, "A" * n, "I" * n, n)
    sam = (mk(40, 1) + mk(36, 2) + mk(50, 3)).encode()
    assert host.pack_sam(sam, fa).read_length == 36
    assert host.pack_sam(sam, fa, var_length=True).read_length == 50


def test_container_layout(built):
    fa, sam, _, _ = synth.dataset(3, [100000, 50000], [300, 100], 100)
    pb = host.pack_sam(sam, fa, block_reads=128)
    sizes = np.arange(10, 10 + pb.n_blocks, dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    payload = np.arange(int(offs[-1]), dtype=np.uint64).astype(np.uint8)
    blob = pb.container(payload, offs)
    magic, ver, L0, nc, nb, nbytes, cap_pos, cap_var, max_rl = struct.unpack_from("<9I", blob, 0)
    assert magic == 0x42434243 and ver == 2 and L0 == 100 and nc == 2 and nb == pb.n_blocks
    assert (cap_pos, cap_var, max_rl) == (pb.cap_pos, pb.cap_var, 100)
    p = 36 + ((nbytes + 3) & ~3)
    p += 16 * nc
    for b in range(nb):
        contig, nreads, w0, poff, pbytes, _ = struct.unpack_from("<IIQQII", blob, p + 32 * b)
        assert (contig, nreads, w0) == (int(pb.info[b]["contig"]), int(pb.info[b]["n_reads"]), int(pb.info[b]["window_start"]))
        assert poff == int(offs[b]) and pbytes == int(sizes[b])
    assert blob[p + 32 * nb:] == payload.tobytes()


def test_synth_is_seeded_and_sorted(built):
    a = host.synth(5, 500000, 5000, 150)
    b = host.synth(5, 500000, 5000, 150)
    c = host.synth(6, 500000, 5000, 150)
    assert a.seq.tobytes() == b.seq.tobytes() and a.recs.tobytes() == b.recs.tobytes()
    assert a.seq.tobytes() != c.seq.tobytes()
    assert a.n_bases == 5000 * 150 and set(np.unique(a.recs["flag"]).tolist()) <= {0, 16}


def test_unpack_plan_matches_the_container(built):
    fa, sam, _, _ = synth.dataset(3, [100000, 50000], [300, 100], 100)
    pb = host.pack_sam(sam, fa, block_reads=128)
    sizes = np.arange(10, 10 + pb.n_blocks, dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    payload = np.arange(int(offs[-1]), dtype=np.uint64).astype(np.uint8)
    plan = host.UnpackPlan(pb.container(payload, offs), fa)
    assert plan.n_blocks == pb.n_blocks and plan.n_recs == pb.n_recs and plan.seq_stride == 100
    assert (plan.blocks["ref_off"] == pb.blocks["ref_off"]).all() and (plan.blocks["in_bytes"] == sizes).all()
    assert plan.ref.tobytes() == pb.ref.tobytes()
    with pytest.raises(host.CbcInputError):
        host.UnpackPlan(b"nonsense" * 10, fa)
    with pytest.raises(host.CbcInputError, match="different length"):
        host.UnpackPlan(pb.container(payload, offs), fa.replace(b"A", b"", 1))


def test_cli_argument_surface(built, tmp_path):
    """`cbc` keeps the reference's argv conventions (src/main.c:85-204, README.md:58-72); on a box
    without a GPU compression stops with a clear error instead of falling back to a CPU path."""
    import subprocess
    exe = os.path.join(ROOT, "cbc_amd", "csrc", "cbc")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cbc_amd", "csrc"), "cbc"], stdout=subprocess.DEVNULL)
    run = lambda *a: subprocess.run([exe, *a], capture_output=True, text=True)
    r = run()
    assert r.returncode == 1 and "Missing required filenames" in r.stderr and "usage:" in r.stderr
    r = run("-u", "1", "a", "b", "c")
    assert r.returncode == 1 and "out of scope" in r.stderr
    r = run("-c", "0.5", "a", "b", "c")
    assert r.returncode == 1 and "lossy" in r.stderr
    r = run("-c", "a", "b", "c", "d")
    assert r.returncode == 1 and "Garbage argument" in r.stderr
    r = run("-d", "user@host:file", "b", "c")
    assert r.returncode == 1 and "out of scope" in r.stderr
    fa, sam, _, _ = synth.dataset(8, [50000], [200], 100)
    (tmp_path / "in.sam").write_bytes(sam)
    (tmp_path / "ref.fa").write_bytes(fa)
    if gpu.lib().cbc_gpu_device_count() == 0:
        for form in (["-c", "1"], ["-c"]):                       # main.c form and README form
            r = run(*form, str(tmp_path / "in.sam"), str(tmp_path / "out.cbc"), str(tmp_path / "ref.fa"))
            assert r.returncode == 1 and "no usable MI355X" in r.stderr and not (tmp_path / "out.cbc").exists()
    bad = sam.replace(b"\t100M\t", b"\t*\t", 1)
    (tmp_path / "bad.sam").write_bytes(bad)
    r = run("-c", str(tmp_path / "bad.sam"), str(tmp_path / "out.cbc"), str(tmp_path / "ref.fa"))
    assert r.returncode == 1 and "CIGAR" in r.stderr


def test_cfg1_shape_round_trip_on_the_oracle(built):
    """BASELINE.json configs[0]: chrI-sized contig (15 072 434 bp), 100 k x 100 bp reads, CPU encode + decode
    round trip -- on the oracle, and the packer + kernel body agree with it on sampled blocks."""
    import blockref
    from oracle import oracle
    pb, sam, fa = host.synth(0xCBC00001, 15_072_434, 100_000, 100, want_text=True, block_reads=4096)
    stream, st = oracle.encode(sam, fa, return_stats=True)
    assert st.n_records == 100_000 and st.read_length == 100
    text, nr = oracle.decode(stream, fa)
    assert nr == 100_000 and text == b"".join(ln.split(b"\t")[9] + b"\n" for ln in sam.splitlines())
    lines = blockref.mapped_sam_lines(sam)
    for b, payload, res in blockref.emu_encode_blocks(pb, [0, pb.n_blocks // 2, pb.n_blocks - 1]):
        bsam, bfa = blockref.block_alone_inputs(pb, lines, b)
        assert int(res["status"]) == 0 and payload == oracle.encode(bsam, bfa)


# ---------------------------------------------------------------- multi-threaded text path
def _packed_arrays(pb):
    return [np.asarray(a).tobytes() for a in (pb.recs, pb.seq, pb.tok, pb.blocks, pb.names)] + [
        np.asarray(pb.info).tobytes(), np.asarray(pb.contigs).tobytes(),
        pb.n_recs, pb.n_blocks, pb.n_tok, pb.n_bases, pb.n_skipped_unmapped, pb.read_length, pb.cap_pos, pb.cap_var,
        int(pb.c_ptr.contents.max_read_len)]


@pytest.mark.parametrize("threads", [2, 3, 8, 64])
def test_threaded_packer_equals_the_serial_one(built, threads):
    """Three contigs (the second one tiny so a chunk sees two RNAME changes), unmapped records in between,
    indels and trailing soft clips: every array of the threaded pack is the serial pack's."""
    fa, sam, rbc, _ = synth.dataset(11, [30000, 600, 20000], [900, 7, 700], 100, indel_frac=0.2, trailing_s_frac=0.1,
                                     sub_rate=0.01, dup_pos_frac=0.2)
    lines = sam.split(b"\n")
    unm = b"u\t4\t*\t0\t0\t*\t*\t0\t0\t" + b"A" * 100 + b"\t" + b"I" * 100 + b"\tMD:Z:100"
    lines = lines[:40] + [unm] + lines[40:900] + [unm, unm] + lines[900:]
    sam = b"\n".join(lines)
    a = host.pack_sam(sam, fa, threads=1, block_reads=128)
    b = host.pack_sam(sam, fa, threads=threads, block_reads=128)
    assert a.n_skipped_unmapped == 3 and a.n_blocks > 12
    assert _packed_arrays(a) == _packed_arrays(b)


def test_threaded_packer_leading_soft_clips_and_stale_md(built):
    """Leading soft clips rebuild the MD text in place (quirk Q6) and a record without MD inherits the
    rebuilt text: same arrays whether the stale record's chunk saw an MD before it (threaded path) or a
    chunk starts with it (falls back to the serial path)."""
    rng = np.random.default_rng(5)
    contig = synth.make_contig(rng, 5000)
    recs = synth.make_reads(rng, contig, 300, 60, sub_rate=0.02, indel_frac=0.1)
    out = []
    for i, r in enumerate(recs):
        seq = bytearray(r["seq"]); cigar = r["cigar"]; md = r["md"]
        if i % 7 == 3 and cigar == "60M":
            s0 = r["pos"] - 1                                  # clip + 56 aligned bases; the packer rebuilds MD for these
            seq = bytearray(b"TTTT" + contig[s0:s0 + 56].tobytes()); seq[30] = ord("A") if seq[30] != ord("A") else ord("C")
            cigar = "4S56M"; md = "26%s29" % chr(contig[s0 + 26])
        tags = b"NM:i:0" if i % 11 == 5 else ("MD:Z:%s\tNM:i:%d" % (md, r["nm"])).encode()
        out.append(b"r%d\t%d\tc\t%d\t60\t%s\t*\t0\t0\t%s\t%s\t%s\n" % (i, r["flag"], r["pos"], cigar.encode(), bytes(seq), b"I" * 60, tags))
    fa = synth.fasta_text([("c", contig)])
    sam = b"".join(out)
    a = host.pack_sam(sam, fa, threads=1)
    for th in (2, 5, 16):
        assert _packed_arrays(host.pack_sam(sam, fa, threads=th)) == _packed_arrays(a)


def test_threaded_packer_reports_the_same_input_errors(built):
    seq = "A" * 40
    good = lambda i, pos, name="c": ["r%d" % i, 0, name, pos, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"]
    rows = [good(i, 1 + i) for i in range(200)]
    fa_seq = "A" * 600
    cases = []
    bad = [list(r) for r in rows]; bad[150][3] = 3;                   cases.append((bad, "not sorted"))
    bad = [list(r) for r in rows]; bad[120][5] = "*";                 cases.append((bad, "CIGAR '*'"))
    bad = [list(r) for r in rows]; bad[199][2] = "d";                 cases.append((bad, "FASTA has fewer"))
    bad = [list(r) for r in rows]; bad[77] = bad[77][:9];             cases.append((bad, "fewer than 11"))
    bad = [list(r) for r in rows]; bad[60][11] = "MD:Z:10C40";        cases.append((bad, "inconsistent"))
    for table, msg in cases:
        sam, fa = _one(table, fa_seq=fa_seq)
        for th in (1, 4):
            with pytest.raises(host.CbcInputError, match=re.escape(msg)):
                host.pack_sam(sam, fa, threads=th)


def test_threaded_packer_tiny_inputs(built):
    """More threads than lines, a body of one record, an empty body."""
    seq = "A" * 40
    row = ["r0", 0, "c", 1, 60, "40M", "*", 0, 0, seq, "I" * 40, "MD:Z:40"]
    sam, fa = _one([row, row])
    assert _packed_arrays(host.pack_sam(sam, fa, threads=8)) == _packed_arrays(host.pack_sam(sam, fa, threads=1))
    sam1, _ = _one([row])
    for s in (sam1, b"@HD\tVN:1.6\n"):
        res = []
        for th in (1, 8):
            try:
                res.append(_packed_arrays(host.pack_sam(s, fa, threads=th)))
            except host.CbcInputError as e:
                res.append(str(e))
        assert res[0] == res[1]


@pytest.mark.parametrize("threads", [2, 5, 16])
def test_threaded_fasta_loader_equals_the_serial_one(built, threads):
    """Four contigs with lower-case bases, an empty line, lines of different widths, a header that is the
    last line, no trailing newline: reference bytes and the contig table are the serial loader's."""
    rng = np.random.default_rng(3)
    c = [synth.make_contig(rng, n) for n in (5000, 37, 12000, 800)]
    parts = [b">one extra words\n"]
    parts += [c[0][i:i + 70].tobytes().lower() + b"\n" for i in range(0, 5000, 70)]
    parts += [b">two\n", c[1].tobytes() + b"\n", b"\n", b">three\n"]
    parts += [c[2][i:i + 50].tobytes() + b"\n" for i in range(0, 12000, 50)]
    parts += [b">four\n", c[3].tobytes()[:400] + b"\n", c[3].tobytes()[400:]]
    fa = b"".join(parts)
    seq = c[0][:40].tobytes()
    row = ["r0", 0, "one", 1, 60, "40M", "*", 0, 0, seq.decode(), "I" * 40, "MD:Z:40"]
    sam = ("\t".join(str(x) for x in row) + "\n").encode() * 2
    a = host.pack_sam(sam, fa, threads=1)
    b = host.pack_sam(sam, fa, threads=threads)
    assert np.array_equal(np.asarray(a.ref), np.asarray(b.ref))
    lens = lambda pb, n: [(int(pb.c_ptr.contents.contigs[i].ref_off), int(pb.c_ptr.contents.contigs[i].length)) for i in range(n)]
    assert lens(a, 4) == lens(b, 4) and [l for _, l in lens(a, 4)] == [5000, 37, 12000, 800]
    # a FASTA whose first line is not a '>' line: the reference still treats it as the header
    fa2 = b"not a header\n" + c[1].tobytes() + b"\n>x\n" + c[3].tobytes() + b"\n"
    r1 = host.pack_sam(sam, fa2.replace(c[1].tobytes(), c[0][:100].tobytes()), threads=1)
    r2 = host.pack_sam(sam, fa2.replace(c[1].tobytes(), c[0][:100].tobytes()), threads=threads)
    assert np.array_equal(np.asarray(r1.ref), np.asarray(r2.ref))


# ---------------------------------------------------------------- hostile inputs (advisor findings of round 1)
def _container_with_block_entry(poff, pbytes, nreads=10):
    """A syntactically valid one-contig, one-block container whose index entry holds the given values."""
    import struct
    fa, sam, _, _ = synth.dataset(3, [5000], [10], 100, sub_rate=0.0, indel_frac=0.0)
    pb = host.pack_sam(sam, fa)
    blob = bytearray(pb.container(np.zeros(4200, dtype=np.uint8), np.array([0, 4200], dtype=np.uint64)))
    names_pad = (len(pb.names) + 3) & ~3
    e = 36 + names_pad + 16 * 1                     # the block index follows the contig table
    struct.pack_into("<IIQQII", blob, e, 0, nreads, 0, poff, pbytes, 0)
    return bytes(blob), fa


@pytest.mark.parametrize("poff,pbytes,nreads", [
    (0xfffffffffffff000, 4100, 10),                 # poff + pbytes wraps to 4
    (0xfffffffffffffff8, 16, 10),                   # wraps to 8
    (0, 4200, 16385),                               # more records than a block may hold
    (4000, 300, 10),                                # plain overrun
])
def test_unpack_plan_rejects_wrapping_block_index(built, poff, pbytes, nreads):
    blob, fa = _container_with_block_entry(poff, pbytes, nreads)
    with pytest.raises(host.CbcInputError, match="corrupt block index"):
        host.UnpackPlan(blob, fa)
    blob, fa = _container_with_block_entry(0, 4200, 10)      # the same entry with honest values parses
    host.UnpackPlan(blob, fa).close()


def test_decoder_refuses_descriptors_that_wrap(built):
    """A descriptor whose in_off / rec_base / seq_base would wrap a 64-bit sum: the kernel body reports ASSERT
    (emulation under bounds checks: no access happens)."""
    import blockref
    pb, sam, fa = host.synth(5, 300_000, 600, 100, want_text=True, block_reads=256)
    payloads, res = blockref.emu_encode(pb)
    plan = host.UnpackPlan(blockref.container_from_payloads(pb, payloads), fa)
    good = plan.blocks.copy()
    for field, val in (("in_off", 0xfffffffffffff000), ("rec_base", 0xffffffffffffff00), ("seq_base", 0xffffffffffff0000)):
        plan.blocks[:] = good
        plan.blocks[1][field] = val
        recs, seq, dres = blockref.emu_decode(plan)
        assert int(dres[1]["status"]) == 2, field
        assert int(dres[0]["status"]) == 0 and int(dres[2]["status"]) == 0
    plan.blocks[:] = good
    plan.close()


def test_soft_clip_cigar_past_the_contig_is_rejected_under_asan(built, tmp_path):
    """CIGARs that walk off the reference / SEQ after a leading soft clip (the MD is derived from the alignment
    there): input errors, checked on an AddressSanitizer build of the packer in a child process."""
    import subprocess, sys, textwrap
    csrc = os.path.join(ROOT, "cbc_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "libcbc_host_asan.so"], stdout=subprocess.DEVNULL)
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np, synth
        from cbc_amd import host
        host.HOST_LIB = %r
        rng = np.random.default_rng(1)
        contig = synth.make_contig(rng, 3000)
        fa = synth.fasta_text([("c", contig)])
        def rec(cigar, md="5"):
            seq = bytearray(contig[2800 - 1:2800 - 1 + 100].tobytes()); seq[0] = ord("A") if seq[0] != ord("A") else ord("C")
            return synth.sam_text([("c", 3000, [dict(pos=2800, flag=0, cigar=cigar, seq=bytes(seq), md=md, nm=1)])])
        n = 0
        for cigar in ("1S" + "4096M" * 40, "1S10M4000D89M", "1S99M" + "300D" * 8, "1S200M"):
            for threads in (1, 4):
                try:
                    host.pack_sam(rec(cigar), fa, threads=threads)
                except host.CbcInputError as e:
                    n += 1
        print("REJECTED", n)
    """ % (ROOT, os.path.join(ROOT, "tests"), os.path.join(csrc, "libcbc_host_asan.so")))
    env = dict(os.environ, LD_PRELOAD=subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip(),
               ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert "REJECTED 8" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
