"""ctypes view of libcbc_host.so (the plain-C host side: packer, container, synthetic workload).

Mirrors, for the hot path only, the reference's record loader / FASTA loader seam
(src/sam_file_allocation.c:437-529, src/read_decompression.c:17-53): text in, packed blocks out.
Python is plumbing here; the work is done by cbc_amd/csrc/cbc_pack.c.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
HOST_LIB = os.path.join(_CSRC, "libcbc_host.so")

CBC_CAP_FLAG = 64
CBC_CAP_NAME = 128
CBC_REF_PAD = 512
CBC_MAX_READ_LEN = 252


class ReadRec(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_uint32), ("flag", ctypes.c_uint16), ("rlen", ctypes.c_uint16),
                ("seq_off", ctypes.c_uint32), ("tok_off", ctypes.c_uint32)]


class BlockDesc(ctypes.Structure):
    _fields_ = [("rec_base", ctypes.c_uint64), ("seq_base", ctypes.c_uint64), ("tok_base", ctypes.c_uint64),
                ("ref_off", ctypes.c_uint64), ("out_off", ctypes.c_uint64), ("out_cap", ctypes.c_uint32),
                ("n_reads", ctypes.c_uint32), ("name_off", ctypes.c_uint32), ("read_length", ctypes.c_uint32),
                ("n_tok", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class BlockResult(ctypes.Structure):
    _fields_ = [("nbytes", ctypes.c_uint32), ("status", ctypes.c_uint32), ("n_symbols", ctypes.c_uint32),
                ("fail_read", ctypes.c_uint32)]


class BlockInfo(ctypes.Structure):
    _fields_ = [("contig", ctypes.c_uint32), ("n_reads", ctypes.c_uint32), ("window_start", ctypes.c_uint64),
                ("n_bases", ctypes.c_uint64)]


class ContigInfo(ctypes.Structure):
    _fields_ = [("ref_off", ctypes.c_uint64), ("length", ctypes.c_uint64), ("name_off", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


class LdsCaps(ctypes.Structure):
    _fields_ = [("cap_pos", ctypes.c_uint32), ("cap_var", ctypes.c_uint32)]


class Packed(ctypes.Structure):
    _fields_ = [
        ("recs", ctypes.POINTER(ReadRec)), ("n_recs", ctypes.c_uint64),
        ("seq", ctypes.POINTER(ctypes.c_uint8)), ("seq_bytes", ctypes.c_uint64),
        ("tok", ctypes.POINTER(ctypes.c_uint32)), ("n_tok", ctypes.c_uint64),
        ("names", ctypes.POINTER(ctypes.c_uint8)), ("names_bytes", ctypes.c_uint32),
        ("blocks", ctypes.POINTER(BlockDesc)), ("n_blocks", ctypes.c_uint32),
        ("info", ctypes.POINTER(BlockInfo)),
        ("contigs", ctypes.POINTER(ContigInfo)), ("n_contigs", ctypes.c_uint32),
        ("ref", ctypes.POINTER(ctypes.c_uint8)), ("ref_bytes", ctypes.c_uint64),
        ("caps", LdsCaps),
        ("read_length", ctypes.c_uint32),
        ("n_bases", ctypes.c_uint64),
        ("n_skipped_unmapped", ctypes.c_uint64),
        ("max_read_len", ctypes.c_uint32), ("whole_file", ctypes.c_uint32),
        ("cap_recs", ctypes.c_uint64), ("cap_seq", ctypes.c_uint64), ("cap_tok", ctypes.c_uint64),
        ("cap_ref", ctypes.c_uint64), ("cap_names", ctypes.c_uint32), ("cap_blocks", ctypes.c_uint32),
        ("cap_contigs", ctypes.c_uint32),
    ]


class DecBlockDesc(ctypes.Structure):
    _fields_ = [("in_off", ctypes.c_uint64), ("ref_off", ctypes.c_uint64), ("rec_base", ctypes.c_uint64),
                ("seq_base", ctypes.c_uint64), ("in_bytes", ctypes.c_uint32), ("n_reads", ctypes.c_uint32),
                ("read_length", ctypes.c_uint32), ("seq_stride", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 4)]


class UnpackPlanC(ctypes.Structure):
    _fields_ = [("blocks", ctypes.POINTER(DecBlockDesc)), ("n_blocks", ctypes.c_uint32),
                ("payloads", ctypes.POINTER(ctypes.c_uint8)), ("payload_bytes", ctypes.c_uint64),
                ("ref", ctypes.POINTER(ctypes.c_uint8)), ("ref_bytes", ctypes.c_uint64),
                ("window_start", ctypes.POINTER(ctypes.c_uint64)),
                ("caps", LdsCaps), ("read_length", ctypes.c_uint32), ("seq_stride", ctypes.c_uint32),
                ("n_recs", ctypes.c_uint64), ("long_reads", ctypes.c_uint32), ("max_read_len", ctypes.c_uint32),
                ("seq_total", ctypes.c_uint64)]


class PackOpts(ctypes.Structure):
    _fields_ = [("block_reads", ctypes.c_uint32), ("max_cap_pos", ctypes.c_uint32),
                ("max_cap_var", ctypes.c_uint32), ("var_length", ctypes.c_uint32),
                ("n_threads", ctypes.c_uint32), ("whole_file", ctypes.c_uint32), ("long_reads", ctypes.c_uint32)]


class SynthOpts(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("contig_len", ctypes.c_uint64), ("n_reads", ctypes.c_uint64),
                ("read_len", ctypes.c_uint32), ("sub_rate", ctypes.c_double), ("indel_frac", ctypes.c_double),
                ("name", ctypes.c_char_p)]


class TwoBitRun(ctypes.Structure):
    _fields_ = [("start", ctypes.c_uint64), ("length", ctypes.c_uint32), ("byte", ctypes.c_uint32)]


class TwoBitC(ctypes.Structure):
    _fields_ = [("codes", ctypes.POINTER(ctypes.c_uint32)), ("n_bases", ctypes.c_uint64),
                ("runs", ctypes.POINTER(TwoBitRun)), ("n_runs", ctypes.c_uint64)]


RUN_DTYPE = np.dtype([("start", "<u8"), ("length", "<u4"), ("byte", "<u4")])


class CbcInputError(ValueError):
    """The input violates a limit of the reference format (message from the packer)."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("libcbc_host.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C cbc_amd/csrc`")
        L = ctypes.CDLL(HOST_LIB)
        L.cbc_pack_default_opts.argtypes = [ctypes.POINTER(PackOpts)]
        L.cbc_pack_sam.restype = ctypes.c_int
        L.cbc_pack_sam.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.POINTER(PackOpts), ctypes.POINTER(ctypes.POINTER(Packed)),
                                   ctypes.c_char_p, ctypes.c_size_t]
        L.cbc_packed_free.argtypes = [ctypes.POINTER(Packed)]
        L.cbc_synth_packed.restype = ctypes.c_int
        L.cbc_synth_packed.argtypes = [ctypes.POINTER(SynthOpts), ctypes.POINTER(PackOpts),
                                       ctypes.POINTER(ctypes.POINTER(Packed)),
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.c_char_p, ctypes.c_size_t]
        L.cbc_synth_long.restype = ctypes.c_int
        L.cbc_synth_long.argtypes = L.cbc_synth_packed.argtypes
        L.cbc_free.argtypes = [ctypes.c_void_p]
        L.cbc_container_size.restype = ctypes.c_int64
        L.cbc_container_size.argtypes = [ctypes.POINTER(Packed), ctypes.POINTER(ctypes.c_uint64)]
        L.cbc_container_write.restype = ctypes.c_int64
        L.cbc_container_write.argtypes = [ctypes.POINTER(Packed), ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64),
                                          ctypes.c_void_p, ctypes.c_uint64]
        L.cbc_pack_from_device_tokens.restype = ctypes.c_int
        L.cbc_pack_from_device_tokens.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(PackOpts),
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                                  ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                                  ctypes.POINTER(ctypes.POINTER(Packed)), ctypes.c_char_p, ctypes.c_size_t]
        L.cbc_sam_body_offset.restype = ctypes.c_uint64
        L.cbc_sam_body_offset.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.cbc_2bit_pack.restype = ctypes.c_int
        L.cbc_2bit_pack.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.POINTER(TwoBitC))]
        L.cbc_2bit_unpack.restype = ctypes.c_int
        L.cbc_2bit_unpack.argtypes = [ctypes.POINTER(TwoBitC), ctypes.c_void_p]
        L.cbc_2bit_free.argtypes = [ctypes.POINTER(TwoBitC)]
        L.cbc_assign_contigs.restype = ctypes.c_int
        L.cbc_assign_contigs.argtypes = [ctypes.POINTER(Packed), ctypes.c_uint32, ctypes.c_void_p]
        L.cbc_checksum64.restype = ctypes.c_uint64
        L.cbc_checksum64.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.cbc_unpack_plan_create.restype = ctypes.c_int
        L.cbc_unpack_plan_create.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                                             ctypes.POINTER(ctypes.POINTER(UnpackPlanC)), ctypes.c_char_p, ctypes.c_size_t]
        L.cbc_unpack_plan_free.argtypes = [ctypes.POINTER(UnpackPlanC)]
        L.cbc_unpack_write_text.restype = ctypes.c_int64
        L.cbc_unpack_write_text.argtypes = [ctypes.POINTER(UnpackPlanC), ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint64]
        _lib = L
    return _lib


def _opts(block_reads=None, max_cap_pos=None, max_cap_var=None, var_length=False, threads=None, whole_file=False,
          long_reads=False):
    o = PackOpts()
    lib().cbc_pack_default_opts(ctypes.byref(o))
    if block_reads is not None:
        o.block_reads = block_reads
    if max_cap_pos is not None:
        o.max_cap_pos = max_cap_pos
    if max_cap_var is not None:
        o.max_cap_var = max_cap_var
    o.var_length = 1 if var_length else 0
    if threads is not None:
        o.n_threads = threads                  # 0 = one per online CPU, 1 = serial text path
    o.whole_file = 1 if whole_file else 0      # "compat": one stream per file, the reference's own format
    o.long_reads = 1 if long_reads else 0      # the long-read format extension (stream version 3)
    if long_reads and block_reads is None:
        o.block_reads = 64
    return o


def _np_view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    addr = ctypes.addressof(ptr.contents)
    nbytes = int(n) * np.dtype(dtype).itemsize
    buf = (ctypes.c_uint8 * nbytes).from_address(addr)
    return np.frombuffer(buf, dtype=dtype)


REC_DTYPE = np.dtype([("pos", "<u4"), ("flag", "<u2"), ("rlen", "<u2"), ("seq_off", "<u4"), ("tok_off", "<u4")])
BLOCK_DTYPE = np.dtype([("rec_base", "<u8"), ("seq_base", "<u8"), ("tok_base", "<u8"), ("ref_off", "<u8"),
                        ("out_off", "<u8"), ("out_cap", "<u4"), ("n_reads", "<u4"), ("name_off", "<u4"),
                        ("read_length", "<u4"), ("n_tok", "<u4"), ("reserved", "<u4")])
RESULT_DTYPE = np.dtype([("nbytes", "<u4"), ("status", "<u4"), ("n_symbols", "<u4"), ("fail_read", "<u4")])
INFO_DTYPE = np.dtype([("contig", "<u4"), ("n_reads", "<u4"), ("window_start", "<u8"), ("n_bases", "<u8")])
CONTIG_DTYPE = np.dtype([("ref_off", "<u8"), ("length", "<u8"), ("name_off", "<u4"), ("reserved", "<u4")])
DEC_BLOCK_DTYPE = np.dtype([("in_off", "<u8"), ("ref_off", "<u8"), ("rec_base", "<u8"), ("seq_base", "<u8"),
                            ("in_bytes", "<u4"), ("n_reads", "<u4"), ("read_length", "<u4"), ("seq_stride", "<u4"),
                            ("reserved", "<u4", (4,))])
assert REC_DTYPE.itemsize == 16 and BLOCK_DTYPE.itemsize == 64 and RESULT_DTYPE.itemsize == 16
assert DEC_BLOCK_DTYPE.itemsize == 64


class PackedBatch:
    """Owns a cbc_packed* and exposes its arrays as numpy views (zero copy)."""

    def __init__(self, ptr):
        self._ptr = ptr
        p = ptr.contents
        self.recs = _np_view(p.recs, p.n_recs, REC_DTYPE)
        # a batch from the device tokeniser may carry sizes only: the bases and tokens are resident on the GPU
        self.seq = _np_view(p.seq, p.seq_bytes, np.uint8) if p.seq else np.zeros(0, dtype=np.uint8)
        self.tok = _np_view(p.tok, max(int(p.n_tok), 1), np.uint32) if p.tok else np.zeros(0, dtype=np.uint32)
        self.names = _np_view(p.names, p.names_bytes, np.uint8)
        self.blocks = _np_view(p.blocks, p.n_blocks, BLOCK_DTYPE)
        self.info = _np_view(p.info, p.n_blocks, INFO_DTYPE)
        self.contigs = _np_view(p.contigs, p.n_contigs, CONTIG_DTYPE)
        self.ref = _np_view(p.ref, p.ref_bytes, np.uint8)
        self.cap_pos = int(p.caps.cap_pos)
        self.cap_var = int(p.caps.cap_var)
        self.read_length = int(p.read_length)
        self.n_bases = int(p.n_bases)
        self.n_recs = int(p.n_recs)
        self.n_tok = int(p.n_tok)
        self.n_blocks = int(p.n_blocks)
        self.n_skipped_unmapped = int(p.n_skipped_unmapped)
        self.whole_file = int(p.whole_file) == 1
        self.long_reads = int(p.whole_file) == 2
        self.max_read_len = int(p.max_read_len)

    @property
    def c_ptr(self):
        return self._ptr

    def contig_name(self, ci):
        off = int(self.contigs[ci]["name_off"])
        raw = self.names[off:].tobytes()
        return raw[:raw.index(b"\0")]

    def assign_contigs(self, n_parts):
        """Whole contigs dealt to n_parts, largest first to the least loaded part (cbc_assign_contigs).  Returns the
        part of every contig; blocks_of_part() turns it into block index lists."""
        out = np.zeros(max(len(self.contigs), 1), dtype=np.uint32)
        rc = lib().cbc_assign_contigs(self._ptr, n_parts, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("cbc_assign_contigs failed: %d" % rc)
        return out[:len(self.contigs)]

    def blocks_of_part(self, part_of_contig, part):
        return [b for b in range(self.n_blocks) if int(part_of_contig[int(self.info[b]["contig"])]) == part]

    def container(self, payloads: np.ndarray, out_offsets: np.ndarray) -> bytes:
        offs = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        po = offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        n = lib().cbc_container_size(self._ptr, po)
        if n < 0:
            raise RuntimeError("cbc_container_size failed: %d" % n)
        dst = np.zeros(int(n), dtype=np.uint8)
        pl = np.ascontiguousarray(payloads, dtype=np.uint8)
        w = lib().cbc_container_write(self._ptr, pl.ctypes.data, po, dst.ctypes.data, int(n))
        if w != n:
            raise RuntimeError("cbc_container_write failed: %d" % w)
        return dst.tobytes()

    def close(self):
        if self._ptr is not None:
            # drop the numpy views before the C memory goes away
            for k in ("recs", "seq", "tok", "names", "blocks", "info", "contigs", "ref"):
                setattr(self, k, None)
            lib().cbc_packed_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SUMMARY_DTYPE = np.dtype([("pos", "<u4"), ("flag", "<u2"), ("rl", "<u2"), ("nt_ev", "<u4"), ("line", "<u4")])


def sam_body_offset(sam: bytes) -> int:
    return int(lib().cbc_sam_body_offset(sam, len(sam)))


def pack_from_device_tokens(sam: bytes, fasta: bytes, summaries, rname_change, change_off, change_len, n_unmapped,
                            seq=None, tok=None, seq_bytes=None, n_tok=None, **kw) -> PackedBatch:
    """The serial half of packing after the device tokeniser (cbc_pack_from_device_tokens): contig numbering and block
    cutting over the record summaries.  seq / tok: numpy copies of the tokeniser's arrays (the C side keeps its own
    copies) or None when they stay on the device."""
    o = _opts(**kw)
    out = ctypes.POINTER(Packed)()
    err = ctypes.create_string_buffer(512)
    summaries = np.ascontiguousarray(summaries, dtype=SUMMARY_DTYPE)
    rname_change = np.ascontiguousarray(rname_change, dtype=np.uint8)
    change_off = np.ascontiguousarray(change_off, dtype=np.uint64); change_len = np.ascontiguousarray(change_len, dtype=np.uint32)
    n = len(summaries)
    cseq = ctok = None
    if seq is not None:                       # hand the C side malloc'ed copies it may own
        libc = ctypes.CDLL(None)
        libc.malloc.restype = ctypes.c_void_p
        seq_bytes, n_tok = int(seq_bytes if seq_bytes is not None else len(seq) - 8), int(n_tok if n_tok is not None else len(tok))
        cseq = libc.malloc(seq_bytes + 16); ctok = libc.malloc(max(n_tok, 1) * 4)
        ctypes.memmove(cseq, np.ascontiguousarray(seq, dtype=np.uint8).ctypes.data, seq_bytes)
        ctypes.memmove(ctok, np.ascontiguousarray(tok, dtype=np.uint32).ctypes.data, n_tok * 4)
    rc = lib().cbc_pack_from_device_tokens(sam, len(sam), fasta, len(fasta), ctypes.byref(o), summaries.ctypes.data,
                                           rname_change.ctypes.data, change_off.ctypes.data, change_len.ctypes.data, n, n_unmapped,
                                           cseq, int(seq_bytes), ctok, int(n_tok), ctypes.byref(out), err, 512)
    if rc != 0:
        raise CbcInputError("cbc_pack_from_device_tokens failed (%d): %s" % (rc, err.value.decode(errors="replace")))
    return PackedBatch(out)


def pack_2bit(bases: np.ndarray, threads=0):
    """Bases -> 2-bit transport form (cbc_2bit_pack): (codes uint32 array, exception runs array of RUN_DTYPE)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    out = ctypes.POINTER(TwoBitC)()
    rc = lib().cbc_2bit_pack(bases.ctypes.data, bases.size, threads, ctypes.byref(out))
    if rc != 0:
        raise RuntimeError("cbc_2bit_pack failed: %d" % rc)
    p = out.contents
    codes = _np_view(p.codes, (int(p.n_bases) + 15) // 16, np.uint32).copy()
    runs = _np_view(p.runs, p.n_runs, RUN_DTYPE).copy() if p.n_runs else np.zeros(0, dtype=RUN_DTYPE)
    lib().cbc_2bit_free(out)
    return codes, runs


def unpack_2bit(codes: np.ndarray, runs: np.ndarray, n_bases: int) -> np.ndarray:
    """Host inverse of pack_2bit (cbc_2bit_unpack)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint32)
    runs = np.ascontiguousarray(runs, dtype=RUN_DTYPE)
    c = TwoBitC(codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n_bases,
                ctypes.cast(runs.ctypes.data, ctypes.POINTER(TwoBitRun)) if len(runs) else None, len(runs))
    out = np.zeros(n_bases, dtype=np.uint8)
    rc = lib().cbc_2bit_unpack(ctypes.byref(c), out.ctypes.data)
    if rc != 0:
        raise RuntimeError("cbc_2bit_unpack failed: %d" % rc)
    return out


def checksum64(data) -> int:
    """cbc_checksum64 over a bytes object or a uint8 numpy array (host twin of the device checksum)."""
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)
    return int(lib().cbc_checksum64(a.ctypes.data if a.size else None, a.size))


def pack_sam(sam: bytes, fasta: bytes, **kw) -> PackedBatch:
    """SAM text + FASTA text -> packed blocks (the load_sam_line / store_reference_in_memory seam)."""
    o = _opts(**kw)
    out = ctypes.POINTER(Packed)()
    err = ctypes.create_string_buffer(512)
    rc = lib().cbc_pack_sam(sam, len(sam), fasta, len(fasta), ctypes.byref(o), ctypes.byref(out), err, 512)
    if rc != 0:
        raise CbcInputError("cbc_pack_sam failed (%d): %s" % (rc, err.value.decode(errors="replace")))
    return PackedBatch(out)


def synth_long(seed, contig_len, n_reads, read_len=10_000, edit_rate=0.05, name=b"chrL", want_text=False, **kw):
    """cfg5 workload: long reads with an indel + substitution mix, packed for the long-read format."""
    return synth(seed, contig_len, n_reads, read_len, edit_rate, 0.0, name, want_text, _long=True, **kw)


def synth(seed, contig_len, n_reads, read_len=150, sub_rate=0.003, indel_frac=0.02, name=b"chr1",
          want_text=False, _long=False, **kw):
    """Seeded synthetic workload.  Returns PackedBatch, or (PackedBatch, sam, fasta)."""
    so = SynthOpts(seed, contig_len, n_reads, read_len, sub_rate, indel_frac, name)
    o = _opts(long_reads=_long, **kw)
    out = ctypes.POINTER(Packed)()
    err = ctypes.create_string_buffer(512)
    sam_p, fa_p = ctypes.c_void_p(), ctypes.c_void_p()
    sam_n, fa_n = ctypes.c_size_t(), ctypes.c_size_t()
    fn = lib().cbc_synth_long if _long else lib().cbc_synth_packed
    rc = fn(ctypes.byref(so), ctypes.byref(o), ctypes.byref(out),
                                ctypes.byref(sam_p) if want_text else None, ctypes.byref(sam_n),
                                ctypes.byref(fa_p) if want_text else None, ctypes.byref(fa_n), err, 512)
    if rc != 0:
        raise CbcInputError("cbc_synth_packed failed (%d): %s" % (rc, err.value.decode(errors="replace")))
    pb = PackedBatch(out)
    if not want_text:
        return pb
    # ctypes.string_at takes a C int: texts of 2 GiB and more (10 M reads) go through a buffer view
    sam = bytes((ctypes.c_char * sam_n.value).from_address(sam_p.value)) if sam_n.value else b""
    fa = bytes((ctypes.c_char * fa_n.value).from_address(fa_p.value)) if fa_n.value else b""
    lib().cbc_free(sam_p)
    lib().cbc_free(fa_p)
    return pb, sam, fa


class UnpackPlan:
    """Container + FASTA -> decode launch plan (cbc_unpack_plan_create).  Keeps the container bytes alive."""

    def __init__(self, container: bytes, fasta: bytes):
        self._blob = np.frombuffer(container, dtype=np.uint8).copy()
        out = ctypes.POINTER(UnpackPlanC)()
        err = ctypes.create_string_buffer(512)
        rc = lib().cbc_unpack_plan_create(self._blob.ctypes.data, self._blob.size, fasta, len(fasta),
                                          ctypes.byref(out), err, 512)
        if rc != 0:
            raise CbcInputError("cbc_unpack_plan_create failed (%d): %s" % (rc, err.value.decode(errors="replace")))
        self._ptr = out
        p = out.contents
        self.n_blocks = int(p.n_blocks)
        self.n_recs = int(p.n_recs)
        self.read_length = int(p.read_length)
        self.seq_stride = int(p.seq_stride)
        self.cap_pos, self.cap_var = int(p.caps.cap_pos), int(p.caps.cap_var)
        self.blocks = _np_view(p.blocks, p.n_blocks, DEC_BLOCK_DTYPE)
        self.payloads = _np_view(p.payloads, p.payload_bytes, np.uint8)
        self.ref = _np_view(p.ref, p.ref_bytes, np.uint8)
        self.window_start = _np_view(p.window_start, p.n_blocks, np.uint64)
        self.long_reads = bool(p.long_reads)
        self.max_read_len = int(p.max_read_len)
        self.seq_total = int(p.seq_total)

    def text(self, recs: np.ndarray, seq: np.ndarray) -> bytes:
        """One reconstructed read per line (what `cbc -d` writes)."""
        cap = (int(self.seq_total) + int(self.n_recs) + 16) if self.long_reads else int(self.n_recs) * (self.seq_stride + 1) + 16
        dst = np.zeros(cap, dtype=np.uint8)
        recs = np.ascontiguousarray(recs)
        seq = np.ascontiguousarray(seq)
        n = lib().cbc_unpack_write_text(self._ptr, recs.ctypes.data, seq.ctypes.data, dst.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("cbc_unpack_write_text failed: %d" % n)
        return dst[:int(n)].tobytes()

    def close(self):
        if self._ptr is not None:
            for k in ("blocks", "payloads", "ref", "window_start"):
                setattr(self, k, None)
            lib().cbc_unpack_plan_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
