"""The device SAM tokeniser: cbc_amd/csrc/cbc_tok_core.h holds its per-line / per-record
functions; the same functions run line by line on the CPU here (tests/emu) and one thread per line on the GPU (-m gpu).
Parity = the resulting packed batch is IDENTICAL, array for array, to the host packer's (cbc_pack_sam)."""
import os
import subprocess

import numpy as np
import pytest

import blockref
import synth
from cbc_amd import gpu, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARRAYS = ("recs", "seq", "tok", "blocks", "info", "names", "contigs")


def _same(pd, ph):
    for k in ARRAYS:
        assert getattr(pd, k).tobytes() == getattr(ph, k).tobytes(), k
    assert (pd.cap_pos, pd.cap_var, pd.read_length, pd.n_bases, pd.n_skipped_unmapped) == (ph.cap_pos, ph.cap_var, ph.read_length, ph.n_bases, ph.n_skipped_unmapped)


def _emu_pack(sam, fa, **kw):
    t = blockref.emu_tokenise(sam)
    return host.pack_from_device_tokens(sam, fa, t["summaries"], t["rname_change"], t["change_off"], t["change_len"], t["n_unmapped"],
                                        seq=t["seq"], tok=t["tok"], seq_bytes=t["seq_bytes"], n_tok=t["n_tok"], **kw)


SHAPES = [
    (dict(), 150, 1024),
    (dict(sub_rate=0.02, indel_frac=0.5, dup_pos_frac=0.1, trailing_s_frac=0.2), 100, 512),
    (dict(flags=(0, 16, 83, 99, 147, 163, 4, 77)), 150, 2048),          # unmapped records (FLAG & 4) are dropped
    (dict(sub_rate=0.45, indel_frac=0.3), 150, 256),
]


@pytest.mark.parametrize("kw,L,br", SHAPES)
def test_core_equals_host_packer(built, kw, L, br):
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [2500, 900], L, **kw)
    _same(_emu_pack(sam, fa, block_reads=br), host.pack_sam(sam, fa, block_reads=br, threads=1))


def test_core_quirks_and_limits(built):
    # Q2: MD as the last column keeps its newline -> the phantom N token
    fa, _, rbc, _ = synth.dataset(9, [100000], [600], 100, sub_rate=0.02, indel_frac=0.0)
    sam = synth.sam_text(rbc, md_last=True)
    _same(_emu_pack(sam, fa, block_reads=200), host.pack_sam(sam, fa, block_reads=200, threads=1))
    # header lines, a last line without '\n', consecutive tabs (strtok skips empty columns)
    fa, sam, _, _ = synth.dataset(10, [60000], [50], 100)
    sam2 = b"@HD\tVN:1.6\n@SQ\tSN:x\tLN:60000\n" + sam.rstrip(b"\n").replace(b"\t60\t", b"\t\t60\t", 5)
    _same(_emu_pack(sam2, fa), host.pack_sam(sam2, fa, threads=1))
    # what the device path hands to the host packer (status 3) and what it refuses like the host packer does
    from test_emu_parity import _soft_clip_sam
    fas, sams = _soft_clip_sam(3)
    with pytest.raises(ValueError) as e:
        blockref.emu_tokenise(sams)
    assert e.value.args[0][0] == 3
    fa, sam, _, _ = synth.dataset(10, [60000], [50], 100, sub_rate=0.03, indel_frac=0.0)
    lines = sam.splitlines(keepends=True)
    import re
    k = next(i for i, ln in enumerate(lines) if re.search(rb"MD:Z:(\d+)[ACGT]", ln) and int(re.search(rb"MD:Z:(\d+)[ACGT]", ln).group(1)) >= 20)
    for mutate, status in ((lambda f: f[:5] + [b"*"] + f[6:], 7),
                           (lambda f: f[:9], 4),
                           (lambda f: f[:9] + [f[9][:10]] + f[10:], 9),                       # the MD names a mismatch beyond the SEQ
                           (lambda f: f[:9] + [f[9] * 3] + f[10:], 10)):
        f = lines[k].rstrip(b"\n").split(b"\t")
        bad = b"".join(lines[:k]) + b"\t".join(mutate(f)) + b"\n" + b"".join(lines[k + 1:])
        with pytest.raises(ValueError) as e:
            blockref.emu_tokenise(bad)
        assert e.value.args[0] == (status, k), (status, e.value.args)
        if status != 3:
            with pytest.raises(host.CbcInputError):
                host.pack_sam(bad, fa, threads=1)
    # a record without an MD field inherits the text of the nearest earlier line that has one (read_line_t.edits persists):
    # on the device path too, and an unmapped line's MD counts.  Here: perfect reads made MD-less (they inherit a perfect or an
    # imperfect neighbour's text -- the latter is refused as inconsistent by both packers unless the texts happen to fit), so
    # strip MD only where the previous line's MD is a plain number
    fa3, sam3, _, _ = synth.dataset(12, [80000], [400], 100, sub_rate=0.004, indel_frac=0.0, flags=(0, 16, 4))
    ls = sam3.splitlines(keepends=True)
    stripped = 0
    for i in range(1, len(ls)):
        f, g = ls[i].rstrip(b"\n").split(b"\t"), ls[i - 1].rstrip(b"\n").split(b"\t")
        if len(f) > 11 and f[11] == b"MD:Z:100" and len(g) > 11 and g[11] == b"MD:Z:100" and i % 3 == 0:
            ls[i] = b"\t".join(f[:11] + f[12:]) + b"\n"; stripped += 1
    assert stripped > 20
    sam3 = b"".join(ls)
    _same(_emu_pack(sam3, fa3, block_reads=100), host.pack_sam(sam3, fa3, block_reads=100, threads=1))
    # ... and with no MD anywhere before it within the look-back, the device path hands the file to the host packer
    body = [l for l in ls if not l.startswith(b"@") and int(l.split(b"\t")[1]) & 4 == 0 and b"MD:Z:100" in l]
    no_md_at_all = b"\t".join(body[0].rstrip(b"\n").split(b"\t")[:11]) + b"\n"
    with pytest.raises(ValueError) as e:
        blockref.emu_tokenise(body[0] + no_md_at_all * 5000)
    assert e.value.args[0][0] == 3 and e.value.args[0][1] == 4097
    blockref.emu_tokenise(body[0] + no_md_at_all * 4096)                              # within the look-back: fine
    long_line = b"".join(lines[:2]) + lines[2].rstrip(b"\n") + b"\tXX:Z:" + b"y" * 900 + b"\n"
    with pytest.raises(ValueError) as e:
        blockref.emu_tokenise(long_line)
    assert e.value.args[0] == (5, 2)


# ----------------------------------------------------------------------------------------- on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kw,L,br", SHAPES)
def test_gpu_tokeniser_equals_host_packer(built, kw, L, br):
    enc = gpu.Encoder(0)
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [2500, 900], L, **kw)
    pd, tr = enc.tokenise_sam(sam, fa, fetch=True, block_reads=br)
    ph = host.pack_sam(sam, fa, block_reads=br)
    _same(pd, ph)
    # encode straight from the tokeniser's device arrays == encode of the host-packed batch
    enc.upload_reference(ph.ref)
    p_dev, r_dev, _, _ = enc.encode_blocks_tokenised(pd, tr)
    p_host, r_host, _, _ = enc.encode_blocks(ph)
    assert (r_dev["status"] == 0).all() and p_dev == p_host
    # ... and == the oracle run on each block's own SAM text + FASTA window: the device tokeniser against the checker,
    # not against the product's other packer (the block cuts come from the device-tokenised batch itself)
    from oracle import oracle
    lines = blockref.mapped_sam_lines(sam)
    assert len(lines) == pd.n_recs
    for b in range(pd.n_blocks):
        bsam, bfa = blockref.block_alone_inputs(pd, lines, b)
        assert p_dev[b] == oracle.encode(bsam, bfa), "block %d of the device-tokenised batch differs from the oracle" % b
    enc.tokenise_free(tr)
    enc.close()


@pytest.mark.gpu
def test_gpu_tokeniser_md_less_records_stay_on_the_device(built):
    """Records without an MD field (they inherit the nearest earlier line's text, an unmapped line's included): tokenised on
    the device, arrays == the host packer's, every block == the oracle on the block's own text."""
    from oracle import oracle
    enc = gpu.Encoder(0)
    fa, sam, _, _ = synth.dataset(12, [80000], [4000], 100, sub_rate=0.004, indel_frac=0.0, flags=(0, 16, 4))
    ls = sam.splitlines(keepends=True)
    stripped = 0
    for i in range(1, len(ls)):
        f, g = ls[i].rstrip(b"\n").split(b"\t"), ls[i - 1].rstrip(b"\n").split(b"\t")
        if len(f) > 11 and f[11] == b"MD:Z:100" and len(g) > 11 and g[11] == b"MD:Z:100" and i % 3 == 0:
            ls[i] = b"\t".join(f[:11] + f[12:]) + b"\n"; stripped += 1
    assert stripped > 200
    sam = b"".join(ls)
    pd, tr = enc.tokenise_sam(sam, fa, fetch=True, block_reads=500)
    ph = host.pack_sam(sam, fa, block_reads=500)
    _same(pd, ph)
    enc.upload_reference(ph.ref)
    p_dev, r_dev, _, _ = enc.encode_blocks_tokenised(pd, tr)
    assert (r_dev["status"] == 0).all()
    lines = blockref.mapped_sam_lines(sam)
    for b in range(pd.n_blocks):
        bsam, bfa = blockref.block_alone_inputs(pd, lines, b)
        assert p_dev[b] == oracle.encode(bsam, bfa), b
    enc.tokenise_free(tr)
    enc.close()


@pytest.mark.gpu
def test_gpu_tokeniser_large_and_errors(built):
    enc = gpu.Encoder(0)
    pb, sam, fa = host.synth(0xCBC00002, 30_000_000, 1_000_000, 150, want_text=True)       # 358 MB of text, 1 M lines
    pd, tr = enc.tokenise_sam(sam, fa, fetch=True)
    _same(pd, pb)
    enc.tokenise_free(tr)
    from test_emu_parity import _soft_clip_sam
    fas, sams = _soft_clip_sam(3)
    with pytest.raises(host.CbcInputError, match="needs the host packer"):
        enc.tokenise_sam(sams, fas)
    lines = sam.splitlines(keepends=True)[:100]
    f = lines[40].split(b"\t"); f[5] = b"*"
    with pytest.raises(host.CbcInputError, match="line 41"):
        enc.tokenise_sam(b"".join(lines[:40]) + b"\t".join(f) + b"".join(lines[41:]), fa)
    enc.close()


@pytest.mark.gpu
def test_cli_device_parse(built, tmp_path):
    """`cbc -c --device-parse` writes the same container as the host-parsed run; with leading soft clips it falls back."""
    exe = os.path.join(ROOT, "cbc_amd", "csrc", "cbc")
    fa, sam, _, _ = synth.dataset(8, [200000, 80000], [3000, 1000], 100, sub_rate=0.01, indel_frac=0.2)
    (tmp_path / "in.sam").write_bytes(sam); (tmp_path / "ref.fa").write_bytes(fa)
    outs = []
    for extra in ([], ["--device-parse"]):
        o = tmp_path / ("o%d.cbc" % len(outs))
        r = subprocess.run([exe, "-c", str(tmp_path / "in.sam"), str(o), str(tmp_path / "ref.fa"), "--block-reads", "1000"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(o.read_bytes())
    assert outs[0] == outs[1]
    from test_emu_parity import _soft_clip_sam
    fas, sams = _soft_clip_sam(3)
    (tmp_path / "s.sam").write_bytes(sams); (tmp_path / "s.fa").write_bytes(fas)
    r = subprocess.run([exe, "-c", str(tmp_path / "s.sam"), str(tmp_path / "s1.cbc"), str(tmp_path / "s.fa"), "--device-parse", "--verbose"], capture_output=True, text=True)
    assert r.returncode == 0 and "parsing on the host" in r.stdout, r.stdout + r.stderr
    r2 = subprocess.run([exe, "-c", str(tmp_path / "s.sam"), str(tmp_path / "s2.cbc"), str(tmp_path / "s.fa")], capture_output=True, text=True)
    assert (tmp_path / "s1.cbc").read_bytes() == (tmp_path / "s2.cbc").read_bytes()
