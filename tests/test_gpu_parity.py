"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle."""
import numpy as np
import pytest

import blockref
import synth
from cbc_amd import gpu, host
from oracle import oracle

pytestmark = pytest.mark.gpu


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
