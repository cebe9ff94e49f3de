"""ctypes binding of oracle/libcbc_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of oracle/cbc_oracle.c).  The product package cbc_amd never does.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcbc_oracle.so")


class OracleStats(ctypes.Structure):
    _fields_ = [
        ("n_records", ctypes.c_uint64),
        ("n_bases", ctypes.c_uint64),
        ("n_symbols", ctypes.c_uint64),
        ("read_length", ctypes.c_uint32),
        ("err", ctypes.c_int32),
    ]


def build(force=False):
    """Compile the restatement with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("cbc_oracle.c", "cbc_cpu.c", "cbc_long.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libcbc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.cbc_oracle_encode.restype = ctypes.c_int64
        L.cbc_oracle_encode.argtypes = [
            ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(OracleStats)]
        L.cbc_oracle_decode.restype = ctypes.c_int64
        L.cbc_oracle_decode.argtypes = [
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
            ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
        L.cbc_oracle_version.restype = ctypes.c_char_p
        L.cbc_cpu_init.restype = ctypes.c_int
        L.cbc_cpu_init.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.cbc_cpu_shutdown.argtypes = [ctypes.c_void_p]
        L.cbc_cpu_upload_reference.restype = ctypes.c_int
        L.cbc_cpu_upload_reference.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        L.cbc_cpu_encode_blocks.restype = ctypes.c_int
        L.cbc_cpu_encode_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.c_void_p, ctypes.c_void_p]
        L.cbc_cpu_long_encode_blocks.restype = ctypes.c_int
        L.cbc_cpu_long_encode_blocks.argtypes = L.cbc_cpu_encode_blocks.argtypes
        L.cbc_cpu_long_decode_block.restype = ctypes.c_int64
        L.cbc_cpu_long_decode_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                                ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


_ERR = {-2: "reference assert would fire", -3: "output capacity", -4: "malformed input", -5: "out of memory"}


def encode(sam: bytes, fasta: bytes, var_length: bool = False, return_stats: bool = False):
    """Whole-file encode: the bytes `program -c 1 in.sam out ref.fa` (-DDEBUG build) writes."""
    L = lib()
    cap = 4096 + min(len(sam), (64 << 20) + len(sam) // 16)  # the stream is far smaller than the SAM text
    out = ctypes.create_string_buffer(cap)
    st = OracleStats()
    n = L.cbc_oracle_encode(sam, len(sam), fasta, len(fasta), out, cap, int(var_length), ctypes.byref(st))
    if n < 0:
        raise OracleError("oracle encode failed: %s (%d)" % (_ERR.get(n, "?"), n))
    data = out.raw[:n]
    return (data, st) if return_stats else data


def decode(stream: bytes, fasta: bytes, max_out: int = None):
    """Whole-file decode: the text `program -x in.cbc out.txt ref.fa` writes (one read per line)."""
    L = lib()
    cap = max_out if max_out is not None else max(1 << 20, len(stream) * 400)
    out = ctypes.create_string_buffer(cap)
    nr = ctypes.c_uint64(0)
    buf = ctypes.create_string_buffer(stream, len(stream))
    n = L.cbc_oracle_decode(buf, len(stream), fasta, len(fasta), out, cap, ctypes.byref(nr))
    if n < 0:
        raise OracleError("oracle decode failed: %s (%d)" % (_ERR.get(n, "?"), n))
    return out.raw[:n], nr.value


def cpu_encode_blocks(pb, blocks=None, return_payloads=False, long_reads=False, return_flat=False):
    """The cbc_cpu_* entry points (oracle/cbc_cpu.c): the SAME packed batch the HIP library takes, coded block
    by block on one core.  `blocks` = optional list of block indices (default: all).  Returns the total payload
    bytes, or (list of payload bytes, results array) with return_payloads, or (flat payload bytes as a uint8 array,
    offsets[n + 1], results array) with return_flat -- the layout of the HIP library's compacted output."""
    import numpy as np
    from cbc_amd import gpu, host          # struct layouts only (ctypes mirrors of include/cbc_gpu.h)
    L = lib()
    ctx = ctypes.c_void_p()
    if L.cbc_cpu_init(0, ctypes.byref(ctx)) != 0:
        raise OracleError("cbc_cpu_init failed")
    try:
        L.cbc_cpu_upload_reference(ctx, pb.ref.ctypes.data, pb.ref.size)
        bl = pb.blocks.copy() if blocks is None else np.ascontiguousarray(pb.blocks[list(blocks)])
        nb = len(bl)
        hb = gpu.HostBatch(pb.recs.ctypes.data, pb.n_recs, pb.seq.ctypes.data, len(pb.seq), pb.tok.ctypes.data, pb.n_tok,
                           pb.names.ctypes.data, len(pb.names), bl.ctypes.data, nb, host.LdsCaps(pb.cap_pos, pb.cap_var))
        cap = int(4096 * nb + 48 * int(bl["n_reads"].sum()) + 8 * int(bl["n_tok"].sum())) + 4096
        if long_reads:
            cap = int(8192 * nb + 9 * pb.n_bases + 64 * pb.n_recs)
        out = np.zeros(cap, dtype=np.uint8)
        offs = np.zeros(nb + 1, dtype=np.uint64)
        res = np.zeros(nb, dtype=host.RESULT_DTYPE)
        fn = L.cbc_cpu_long_encode_blocks if long_reads else L.cbc_cpu_encode_blocks
        rc = fn(ctx, ctypes.byref(hb), out.ctypes.data, cap, offs.ctypes.data, res.ctypes.data)
        if rc not in (0, -4):
            raise OracleError("cbc_cpu_encode_blocks failed: %d" % rc)
        if return_flat:
            return out[:int(offs[nb])].copy(), offs, res
        if not return_payloads:
            return int(offs[nb])
        return [out[int(offs[b]):int(offs[b + 1])].tobytes() for b in range(nb)], res
    finally:
        L.cbc_cpu_shutdown(ctx)


def cpu_long_decode_block(payload: bytes, ref, ref_off, max_recs, max_bases):
    """Decode one long-read-format block on the CPU (oracle/cbc_long.c).  Returns (recs, flat bases)."""
    import numpy as np
    from cbc_amd import host
    L = lib()
    ctx = ctypes.c_void_p()
    L.cbc_cpu_init(0, ctypes.byref(ctx))
    try:
        L.cbc_cpu_upload_reference(ctx, ref.ctypes.data, ref.size)
        recs = np.zeros(max_recs, dtype=host.REC_DTYPE)
        seq = np.zeros(max_bases + 16, dtype=np.uint8)
        buf = np.frombuffer(payload, dtype=np.uint8)
        n = L.cbc_cpu_long_decode_block(ctx, buf.ctypes.data, buf.size, int(ref_off), recs.ctypes.data, max_recs, seq.ctypes.data, seq.size)
        if n < 0:
            raise OracleError("cbc_cpu_long_decode_block failed: %d" % n)
        return recs[:n], seq
    finally:
        L.cbc_cpu_shutdown(ctx)
