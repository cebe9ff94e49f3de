"""2-bit transport of bases: host pack / unpack exactness, and -- on the GPU -- the device
expansion giving the same reference / the same payloads, the device packing giving back the same reads."""
import os

import numpy as np
import pytest

import synth
from cbc_amd import gpu, host


def _rt(arr, threads):
    codes, runs = host.pack_2bit(arr, threads=threads)
    assert len(codes) == (len(arr) + 15) // 16
    back = host.unpack_2bit(codes, runs, len(arr))
    assert (back == arr).all()
    return codes, runs


def test_pack_unpack_is_exact(built):
    rng = np.random.default_rng(4)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for n in (0, 1, 15, 16, 17, 1000, (1 << 22) + 37):
        arr = acgt[rng.integers(0, 4, size=n)].copy()
        if n > 100:
            arr[10:60] = ord("N")                     # a run
            arr[70] = ord("n"); arr[71] = ord("R"); arr[72] = 0
            arr[n - 5:] = 0                           # the zero pad behind a contig
        codes, runs = _rt(arr, 1)
        c4, r4 = _rt(arr, 4)
        assert (c4 == codes).all() and r4.tobytes() == runs.tobytes()          # thread count does not change the result
        if n > 100:
            assert len(runs) == 5 and int(runs[0]["start"]) == 10 and int(runs[0]["length"]) == 50 and int(runs[0]["byte"]) == ord("N")
    # a long N-run is ONE exception entry, whatever the thread boundaries
    arr = np.full(5_000_000, ord("N"), dtype=np.uint8); arr[:16] = ord("A")
    codes, runs = _rt(arr, 8)
    assert len(runs) == 1 and int(runs[0]["length"]) == 5_000_000 - 16


def test_packed_batch_and_reference_round_trip(built):
    pb, sam, fa = host.synth(5, 400_000, 3000, 150, want_text=True)
    for arr in (pb.seq, pb.ref):
        codes, runs = host.pack_2bit(arr)
        assert (host.unpack_2bit(codes, runs, len(arr)) == arr).all()
        assert codes.nbytes + runs.nbytes < len(arr) * 0.27


@pytest.mark.gpu
def test_gpu_2bit_transport(built):
    """Reference and reads uploaded at 2 bits per base give byte-identical payloads; decoded reads packed on the device
    give back the same bases (N included) with a third of the bytes over PCIe."""
    enc = gpu.Encoder(0)
    fa, sam, _, _ = synth.dataset(5, [300000, 120000], [4000, 1500], 150, sub_rate=0.01, indel_frac=0.2)
    # put some N into the reads and the reference (exceptions on both sides)
    lines = sam.splitlines(keepends=True)
    pb = host.pack_sam(sam, fa, block_reads=1024)
    enc.upload_reference(pb.ref)
    p_ref, r_ref, offs, flat = enc.encode_blocks(pb)
    assert (r_ref["status"] == 0).all()
    rc, rr = host.pack_2bit(pb.ref)
    enc.upload_reference_2bit(rc, rr, len(pb.ref))
    p1, r1, _, _ = enc.encode_blocks(pb)
    assert p1 == p_ref
    sc, sr = host.pack_2bit(pb.seq)
    p2, r2, _, _ = enc.encode_blocks_2bit(pb, sc, sr)
    assert p2 == p_ref and (r2["n_symbols"] == r_ref["n_symbols"]).all()
    plan = host.UnpackPlan(pb.container(flat, offs), fa)
    recs, seq, dres = enc.decode_blocks(plan)
    recs2, bases2, dres2, pcie = enc.decode_blocks_2bit(plan)
    assert (dres2["status"] == 0).all()
    assert all((recs2[k] == recs[k]).all() for k in ("pos", "flag", "rlen"))       # seq_off differs: rows are 160 bytes here
    want = seq[:plan.n_recs * plan.seq_stride].reshape(plan.n_recs, plan.seq_stride)[:, :150]
    assert (bases2[:, :150] == want).all()
    assert pcie < 0.3 * plan.n_recs * plan.seq_stride
    # reads with N: exceptions travel separately and come back exactly
    fa2, sam2, rbc, _ = synth.dataset(6, [100000], [800], 100, sub_rate=0.0, indel_frac=0.0)
    for r in rbc[0][2][::7]:
        s = bytearray(r["seq"]); ref_base = s[40]; s[40] = ord("N"); r["seq"] = bytes(s); r["md"] = "40%s59" % chr(ref_base); r["nm"] = 1
    sam2 = synth.sam_text(rbc)
    pn = host.pack_sam(sam2, fa2, block_reads=256)
    enc.upload_reference(pn.ref)
    pa, ra, offs, flat = enc.encode_blocks(pn)
    sc, sr = host.pack_2bit(pn.seq)
    assert len(sr) > 100
    pb2, rb2, _, _ = enc.encode_blocks_2bit(pn, sc, sr)
    assert (ra["status"] == 0).all() and pb2 == pa
    plan = host.UnpackPlan(pn.container(flat, offs), fa2)
    recs2, bases2, dres2, _ = enc.decode_blocks_2bit(plan)
    want = np.frombuffer(b"".join(ln.split(b"\t")[9] for ln in sam2.splitlines() if not ln.startswith(b"@")), dtype=np.uint8).reshape(-1, 100)
    assert (bases2[:, :100] == want).all() and (want == ord("N")).sum() > 100
    enc.close()


def test_wrapped_exception_run_is_rejected_under_asan(built):
    """cbc_2bit_unpack with a run whose start + length wraps in 64 bits (round-2 advisor finding: that sum passed the old
    range check and memset wrote out of bounds): an input error, checked on the AddressSanitizer build in a child process;
    so are a run past the end, a run count without a run array, and a good run right at the end."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "cbc_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "libcbc_host_asan.so"], stdout=subprocess.DEVNULL)
    code = textwrap.dedent("""
        import sys, ctypes
        sys.path.insert(0, %r)
        import numpy as np
        from cbc_amd import host
        host.HOST_LIB = %r
        n = 1000
        codes = np.zeros((n + 15) // 16, dtype=np.uint32)
        out = np.zeros(n, dtype=np.uint8)
        def unpack(runs, n_runs=None, null_runs=False):
            r = np.array(runs, dtype=host.RUN_DTYPE)
            c = host.TwoBitC(codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n,
                             None if null_runs else ctypes.cast(r.ctypes.data, ctypes.POINTER(host.TwoBitRun)), len(r) if n_runs is None else n_runs)
            return host.lib().cbc_2bit_unpack(ctypes.byref(c), out.ctypes.data)
        rcs = [unpack([(0xffffffffffffff00, 0x140, 78)]), unpack([(990, 11, 78)]), unpack([(2000, 1, 78)]),
               unpack([], n_runs=3, null_runs=True), unpack([(990, 10, 78)])]
        print("RCS", rcs, bytes(out[988:1000]))
    """ % (root, os.path.join(csrc, "libcbc_host_asan.so")))
    env = dict(os.environ, LD_PRELOAD=subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip(),
               ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert "RCS [-5, -5, -5, -1, 0] b'AANNNNNNNNNN'" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
