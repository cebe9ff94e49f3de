"""A/B at cfg3 size: python tools/ab_big.py lib1.so lib2.so"""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from cbc_amd import host, gpu
pb = host.synth(0xCBC00002, 248956422, 49791284, 150, block_reads=4096)
for lib in sys.argv[1:]:
    pass
enc = gpu.Encoder(0); enc.upload_reference(pb.ref)
ts = []
for i in range(4):
    out = enc.encode_blocks(pb); ts.append(enc.last_kernel_ms())
res = out[1]
print("RESULT", min(ts[1:]), sum(int(x) for x in res["nbytes"]), int((res["status"] != 0).sum()), "lds", gpu.lib().cbc_gpu_lds_bytes(__import__("ctypes").byref(host.LdsCaps(pb.cap_pos, pb.cap_var))))
''' % R
for lib in sys.argv[1:]:
    env = dict(os.environ, CBC_GPU_LIB=os.path.join(R, "scratch", "abl", lib))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("%-20s cfg3-sized: kernel ms, payload bytes, failed blocks: %s" % (lib, line[0][7:] if line else r.stderr[-400:]), flush=True)
