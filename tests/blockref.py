"""Helpers shared by the parity tests: run a packed batch through an encoder back end and build,
for every block, the "block alone" SAM + FASTA the reference would be run on, so the oracle can
produce the expected payload (SURVEY.md section 7 hard part 1: payload == reference on that block)."""
import ctypes
import os
import subprocess

import numpy as np

from cbc_amd import host

_HERE = os.path.dirname(os.path.abspath(__file__))
_EMU_DIR = os.path.join(_HERE, "emu")


class DeviceBatch(ctypes.Structure):
    _fields_ = [
        ("d_recs", ctypes.c_void_p), ("d_seq", ctypes.c_void_p), ("d_tok", ctypes.c_void_p),
        ("d_names", ctypes.c_void_p), ("d_blocks", ctypes.c_void_p), ("n_blocks", ctypes.c_uint32),
        ("d_ref", ctypes.c_void_p), ("ref_bytes", ctypes.c_uint64),
        ("d_out", ctypes.c_void_p), ("out_bytes", ctypes.c_uint64),
        ("d_results", ctypes.c_void_p),
        ("seq_bytes", ctypes.c_uint64), ("n_tok", ctypes.c_uint64), ("n_recs", ctypes.c_uint64),
        ("caps", host.LdsCaps),
    ]


class DecDeviceBatch(ctypes.Structure):
    _fields_ = [
        ("d_in", ctypes.c_void_p), ("in_bytes", ctypes.c_uint64),
        ("d_blocks", ctypes.c_void_p), ("n_blocks", ctypes.c_uint32),
        ("d_ref", ctypes.c_void_p), ("ref_bytes", ctypes.c_uint64),
        ("d_recs", ctypes.c_void_p), ("n_recs", ctypes.c_uint64),
        ("d_seq", ctypes.c_void_p), ("seq_bytes", ctypes.c_uint64),
        ("d_results", ctypes.c_void_p),
        ("d_var_scratch", ctypes.c_void_p), ("var_scratch_words", ctypes.c_uint64),
        ("caps", host.LdsCaps),
    ]


_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        subprocess.check_call(["make", "-C", _EMU_DIR, "libcbc_emu.so"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(os.environ.get("CBC_EMU_LIB") or os.path.join(_EMU_DIR, "libcbc_emu.so"))   # override: debugging builds
        L.emu_encode_blocks.restype = ctypes.c_int
        L.emu_encode_blocks.argtypes = [ctypes.POINTER(DeviceBatch)]
        L.emu_decode_blocks.restype = ctypes.c_int
        L.emu_decode_blocks.argtypes = [ctypes.POINTER(DecDeviceBatch)]
        L.emu_plan_output.restype = ctypes.c_uint64
        L.emu_plan_output.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        _emu = L
    return _emu


def emu_encode(pb, two_wave=False, shrink_block=None):
    """Run the kernel body on the CPU wave emulation.  Returns (payload list, results array).
    two_wave: the model and coder roles of a block as two host threads with the LDS hand-off ring between them."""
    L = emu_lib()
    blocks = pb.blocks.copy()
    total = L.emu_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, pb.tok.ctypes.data)
    if shrink_block is not None:                     # a payload area far too small for the block: the coder stops with OUT_FULL
        blocks[shrink_block]["reserved"] = 256
    out = np.full(int(total), 0xAA, dtype=np.uint8)
    res = np.zeros(pb.n_blocks, dtype=host.RESULT_DTYPE)
    db = DeviceBatch(pb.recs.ctypes.data, pb.seq.ctypes.data, pb.tok.ctypes.data, pb.names.ctypes.data,
                     blocks.ctypes.data, pb.n_blocks, pb.ref.ctypes.data, len(pb.ref), out.ctypes.data, int(total),
                     res.ctypes.data, len(pb.seq), pb.n_tok, pb.n_recs, host.LdsCaps(pb.cap_pos, pb.cap_var))
    rc = (L.emu_encode_blocks_two_wave if two_wave else L.emu_encode_blocks)(ctypes.byref(db))
    if rc != 0:
        raise RuntimeError("emulation reported an invariant violation (rc=%d)" % rc)
    payloads = []
    for b in range(pb.n_blocks):
        o = int(blocks[b]["out_off"])
        payloads.append(out[o:o + int(res[b]["nbytes"])].tobytes())
    return payloads, res


def emu_decode(plan):
    """Run the decoder body on the CPU wave emulation.  Returns (recs, seq, results)."""
    L = emu_lib()
    blocks = plan.blocks.copy()
    pay = np.concatenate([np.ascontiguousarray(plan.payloads), np.zeros(16, dtype=np.uint8)])
    recs = np.zeros(plan.n_recs, dtype=host.REC_DTYPE)
    seq = np.zeros(plan.n_recs * plan.seq_stride + 16, dtype=np.uint8)
    res = np.zeros(plan.n_blocks, dtype=host.RESULT_DTYPE)
    vs = np.zeros(max(plan.n_blocks * plan.cap_var, 1), dtype=np.uint32)
    db = DecDeviceBatch(pay.ctypes.data, pay.size, blocks.ctypes.data, plan.n_blocks, plan.ref.ctypes.data, len(plan.ref),
                        recs.ctypes.data, plan.n_recs, seq.ctypes.data, seq.size, res.ctypes.data,
                        vs.ctypes.data, vs.size, host.LdsCaps(plan.cap_pos, plan.cap_var))
    if L.emu_decode_blocks(ctypes.byref(db)) != 0:
        raise RuntimeError("emulation reported an invariant violation")
    return recs, seq, res


def container_from_payloads(pb, payloads):
    offs = np.zeros(len(payloads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(p) for p in payloads])
    flat = np.frombuffer(b"".join(payloads), dtype=np.uint8) if payloads else np.zeros(0, dtype=np.uint8)
    return pb.container(flat, offs)


def emu_encode_blocks(pb, which):
    """Emulate only the listed blocks (for large batches).  Returns [(b, payload, result)]."""
    L = emu_lib()
    out = []
    for b in which:
        blocks = pb.blocks[b:b + 1].copy()
        total = L.emu_plan_output(blocks.ctypes.data, 1, pb.recs.ctypes.data, pb.tok.ctypes.data)
        buf = np.zeros(int(total), dtype=np.uint8)
        res = np.zeros(1, dtype=host.RESULT_DTYPE)
        db = DeviceBatch(pb.recs.ctypes.data, pb.seq.ctypes.data, pb.tok.ctypes.data, pb.names.ctypes.data,
                         blocks.ctypes.data, 1, pb.ref.ctypes.data, len(pb.ref), buf.ctypes.data, int(total),
                         res.ctypes.data, len(pb.seq), pb.n_tok, pb.n_recs, host.LdsCaps(pb.cap_pos, pb.cap_var))
        if L.emu_encode_blocks(ctypes.byref(db)) != 0:
            raise RuntimeError("emulation reported an invariant violation")
        out.append((b, buf[:int(res[0]["nbytes"])].tobytes(), res[0]))
    return out


def block_alone_inputs(pb, sam_lines, b):
    """SAM + FASTA text of block b alone: its records with POS rebased to the block window, and the
    window (from the block's first base to the end of the contig, capped) as a one-contig FASTA."""
    info = pb.info[b]
    bd = pb.blocks[b]
    ci = int(info["contig"])
    w0 = int(info["window_start"])
    first = int(bd["rec_base"])
    n = int(bd["n_reads"])
    c = pb.contigs[ci]
    coff, clen = int(c["ref_off"]), int(c["length"])
    last_pos = int(pb.recs[first + n - 1]["pos"])
    wend = min(clen, w0 + last_pos + 2 * 256 + 64)
    window = pb.ref[coff + w0: coff + wend].tobytes()
    fa = [b">blk\n"]
    for i in range(0, len(window), 60):
        fa.append(window[i:i + 60] + b"\n")
    lines = []
    for k in range(n):
        f = sam_lines[first + k].split(b"\t")
        f[3] = b"%d" % (int(f[3]) - w0)
        lines.append(b"\t".join(f))
    return b"".join(lines), b"".join(fa)


def mapped_sam_lines(sam: bytes):
    """Record lines (with their newline) of mapped reads, in file order."""
    out = []
    for ln in sam.splitlines(keepends=True):
        if ln.startswith(b"@") or not ln.strip():
            continue
        f = ln.split(b"\t")
        if int(f[1]) & 4:
            continue
        out.append(ln)
    return out


STREAM_POS_LDS, STREAM_POS_MAX = 8192, 5_000_000         # cbc_stream_body.h


def stream_aux_words(cap_pos):
    """cbc_stream_aux_words(): flag pairs + the pos alphabet beyond the LDS part, per stream slot."""
    return 2 * 65536 + 2 * max(cap_pos - STREAM_POS_LDS, 0) + 64


class StreamArgs(ctypes.Structure):
    """cbc_stream_args of cbc_amd/csrc/cbc_stream_body.h."""
    _fields_ = [("recs", ctypes.c_void_p), ("seq", ctypes.c_void_p), ("tok", ctypes.c_void_p), ("names", ctypes.c_void_p),
                ("segs", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("out", ctypes.c_void_p), ("results", ctypes.c_void_p),
                ("vtab", ctypes.c_void_p), ("aux", ctypes.c_void_p),
                ("ref_bytes", ctypes.c_uint64), ("out_bytes", ctypes.c_uint64), ("seq_bytes", ctypes.c_uint64),
                ("n_tok", ctypes.c_uint64), ("n_recs", ctypes.c_uint64),
                ("n_segs", ctypes.c_uint32), ("cap_pos", ctypes.c_uint32), ("cap_name", ctypes.c_uint32),
                ("names_bytes", ctypes.c_uint32), ("per_segment", ctypes.c_uint32), ("n_vtab", ctypes.c_uint32)]


def emu_encode_stream(pb, per_segment=False):
    """The whole-file stream body on the CPU wave emulation.  pb: packed with whole_file=True (one stream) or an
    ordinary block batch with per_segment=True (one general-form stream per block).  Returns (payload list, results)."""
    L = emu_lib()
    L.emu_encode_stream.restype = ctypes.c_int
    L.emu_encode_stream.argtypes = [ctypes.POINTER(StreamArgs)]
    segs = pb.blocks.copy()
    n_streams = pb.n_blocks if per_segment else 1
    off = 0
    for b in range(pb.n_blocks):                       # each stream's area starts at its first segment's out_off
        cap = (4096 + 48 * int(segs[b]["n_reads"]) + 8 * int(segs[b]["n_tok"]) + 255) & ~255
        if not per_segment:
            cap = (4096 + 48 * pb.n_recs + 8 * pb.n_tok + 255) & ~255 if b == 0 else 0
        segs[b]["out_off"] = off; segs[b]["out_cap"] = cap
        off += cap
    out = np.full(off, 0xAA, dtype=np.uint8)
    res = np.zeros(n_streams, dtype=host.RESULT_DTYPE)
    vtab = np.zeros(65535 * 256, dtype=np.uint32)
    cap_pos = max(pb.cap_pos, 64)
    aux = np.full(stream_aux_words(cap_pos), 0xA5A5A5A5, dtype=np.uint32)      # needs no initial content
    a = StreamArgs(pb.recs.ctypes.data, pb.seq.ctypes.data, pb.tok.ctypes.data, pb.names.ctypes.data, segs.ctypes.data,
                   pb.ref.ctypes.data, out.ctypes.data, res.ctypes.data, vtab.ctypes.data, aux.ctypes.data,
                   len(pb.ref), off, len(pb.seq), pb.n_tok, pb.n_recs,
                   pb.n_blocks, cap_pos, len(pb.names) + 2 * pb.n_blocks + 16, len(pb.names), 1 if per_segment else 0, 1)
    if L.emu_encode_stream(ctypes.byref(a)) != 0:
        raise RuntimeError("emulation reported an invariant violation")
    payloads = []
    for s in range(n_streams):
        o = int(segs[s]["out_off"])
        payloads.append(out[o:o + int(res[s]["nbytes"])].tobytes())
    return payloads, res


class DStreamArgs(ctypes.Structure):
    """cbc_dstream_args of cbc_amd/csrc/cbc_stream_body.h."""
    _fields_ = [("in_", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("contig_off", ctypes.c_void_p), ("contig_len", ctypes.c_void_p),
                ("recs", ctypes.c_void_p), ("seq", ctypes.c_void_p), ("results", ctypes.c_void_p), ("vtab", ctypes.c_void_p),
                ("aux", ctypes.c_void_p),
                ("in_bytes", ctypes.c_uint64), ("ref_bytes", ctypes.c_uint64), ("rec_cap", ctypes.c_uint64), ("seq_bytes", ctypes.c_uint64),
                ("n_contigs", ctypes.c_uint32), ("cap_pos", ctypes.c_uint32), ("cap_name", ctypes.c_uint32),
                ("seq_stride", ctypes.c_uint32), ("read_length", ctypes.c_uint32)]


def emu_decode_stream(stream: bytes, ref, contigs, rec_cap, cap_pos=STREAM_POS_MAX, cap_name=2048):
    """The whole-file stream decoder body on the CPU wave emulation.  ref/contigs: the packer's reference layout
    (pb.ref, pb.contigs).  Returns (recs, bases[n, stride], result)."""
    L = emu_lib()
    L.emu_decode_stream.restype = ctypes.c_int
    L.emu_decode_stream.argtypes = [ctypes.POINTER(DStreamArgs)]
    L0 = int.from_bytes(stream[:4], "big")
    stride = 256
    pay = np.concatenate([np.frombuffer(stream, dtype=np.uint8), np.zeros(16, dtype=np.uint8)])
    co = np.ascontiguousarray(contigs["ref_off"], dtype=np.uint64); cl = np.ascontiguousarray(contigs["length"], dtype=np.uint64)
    recs = np.zeros(rec_cap, dtype=host.REC_DTYPE)
    seq = np.zeros(rec_cap * stride + 16, dtype=np.uint8)
    res = np.zeros(1, dtype=host.RESULT_DTYPE)
    vtab = np.zeros(65535 * 256, dtype=np.uint32)
    aux = np.empty(stream_aux_words(cap_pos), dtype=np.uint32)
    a = DStreamArgs(pay.ctypes.data, ref.ctypes.data, co.ctypes.data, cl.ctypes.data, recs.ctypes.data, seq.ctypes.data,
                    res.ctypes.data, vtab.ctypes.data, aux.ctypes.data, len(stream), len(ref), rec_cap, seq.size, len(co), cap_pos, cap_name, stride, L0)
    if L.emu_decode_stream(ctypes.byref(a)) != 0:
        raise RuntimeError("emulation reported an invariant violation")
    n = int(res[0]["nbytes"])
    return recs[:n], seq[:n * stride].reshape(n, stride), res[0]


class LongArgs(ctypes.Structure):
    """cbc_long_args of cbc_amd/csrc/cbc_long_body.h."""
    _fields_ = [("recs", ctypes.c_void_p), ("seq", ctypes.c_void_p), ("tok", ctypes.c_void_p), ("names", ctypes.c_void_p),
                ("blocks", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("out", ctypes.c_void_p), ("results", ctypes.c_void_p),
                ("ref_bytes", ctypes.c_uint64), ("out_bytes", ctypes.c_uint64), ("seq_bytes", ctypes.c_uint64),
                ("n_tok", ctypes.c_uint64), ("n_recs", ctypes.c_uint64),
                ("n_blocks", ctypes.c_uint32), ("cap_pos", ctypes.c_uint32), ("names_bytes", ctypes.c_uint32),
                ("scratch", ctypes.c_void_p)]                      # set by the emulation driver


def emu_long_encode(pb, out_cap_per_base=2.0):
    """The long-read encoder body on the CPU wave emulation.  Returns (payload list, results)."""
    L = emu_lib()
    L.emu_long_encode_blocks.restype = ctypes.c_int
    L.emu_long_encode_blocks.argtypes = [ctypes.POINTER(LongArgs)]
    blocks = pb.blocks.copy()
    off = 0
    for b in range(pb.n_blocks):
        cap = (4096 + 64 * int(blocks[b]["n_reads"]) + int(out_cap_per_base * int(pb.info[b]["n_bases"])) + 255) & ~255
        blocks[b]["out_off"] = off; blocks[b]["out_cap"] = cap
        off += cap
    out = np.full(off, 0xAA, dtype=np.uint8)
    res = np.zeros(pb.n_blocks, dtype=host.RESULT_DTYPE)
    a = LongArgs(pb.recs.ctypes.data, pb.seq.ctypes.data, pb.tok.ctypes.data, pb.names.ctypes.data, blocks.ctypes.data,
                 pb.ref.ctypes.data, out.ctypes.data, res.ctypes.data, len(pb.ref), off, len(pb.seq), pb.n_tok, pb.n_recs,
                 pb.n_blocks, pb.cap_pos, len(pb.names), None)
    if L.emu_long_encode_blocks(ctypes.byref(a)) != 0:
        raise RuntimeError("emulation reported an invariant violation")
    return [out[int(blocks[b]["out_off"]):int(blocks[b]["out_off"]) + int(res[b]["nbytes"])].tobytes() for b in range(pb.n_blocks)], res


def emu_long_decode(plan):
    """The long-read decoder body on the CPU wave emulation.  Returns (recs, flat bases, results)."""
    L = emu_lib()
    L.emu_long_decode_blocks.restype = ctypes.c_int
    L.emu_long_decode_blocks.argtypes = [ctypes.POINTER(DecDeviceBatch)]
    blocks = plan.blocks.copy()
    pay = np.concatenate([np.ascontiguousarray(plan.payloads), np.zeros(16, dtype=np.uint8)])
    recs = np.zeros(plan.n_recs, dtype=host.REC_DTYPE)
    seq = np.zeros(plan.seq_total + 16, dtype=np.uint8)
    res = np.zeros(plan.n_blocks, dtype=host.RESULT_DTYPE)
    db = DecDeviceBatch(pay.ctypes.data, pay.size, blocks.ctypes.data, plan.n_blocks, plan.ref.ctypes.data, len(plan.ref),
                        recs.ctypes.data, plan.n_recs, seq.ctypes.data, seq.size, res.ctypes.data, None, 0,
                        host.LdsCaps(plan.cap_pos, plan.cap_var))
    if L.emu_long_decode_blocks(ctypes.byref(db)) != 0:
        raise RuntimeError("emulation reported an invariant violation")
    return recs, seq, res


def emu_tokenise(sam: bytes):
    """cbc_tok_core.h (the device tokeniser's per-line functions) run line by line on the CPU.  Returns a dict like
    cbc_tok_result, or raises with (status, line)."""
    L = emu_lib()
    L.emu_tokenise.restype = ctypes.c_int
    L.emu_tokenise.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64] + [ctypes.c_void_p] * 7
    n_max = sam.count(b"\n") + 2
    summ = np.zeros(n_max, dtype=host.SUMMARY_DTYPE); chg = np.zeros(n_max, dtype=np.uint8)
    coff = np.zeros(n_max, dtype=np.uint64); clen = np.zeros(n_max, dtype=np.uint32)
    seq = np.zeros(len(sam) + 16, dtype=np.uint8); tok = np.zeros(len(sam) // 2 + 4096, dtype=np.uint32)
    counts = np.zeros(8, dtype=np.uint64)
    st = L.emu_tokenise(sam, len(sam), host.sam_body_offset(sam), summ.ctypes.data, chg.ctypes.data, coff.ctypes.data, clen.ctypes.data,
                        seq.ctypes.data, tok.ctypes.data, counts.ctypes.data)
    if st != 0:
        raise ValueError((st, int(counts[6])))
    n_lines, n_recs, n_unm, sb, nt, nc = (int(x) for x in counts[:6])
    return dict(n_lines=n_lines, n_recs=n_recs, n_unmapped=n_unm, seq_bytes=sb, n_tok=nt, summaries=summ[:n_recs], rname_change=chg[:n_recs],
                change_off=coff[:nc], change_len=clen[:nc], seq=seq[:sb + 8], tok=tok[:nt])
