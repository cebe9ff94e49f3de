"""A/B kernel timing of encoder builds in one process / on one box: python tools/ab.py lib1.so lib2.so ..."""
import sys, os, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from cbc_amd import host, gpu
pb = host.synth(0xCBC00002, 248956422, int(sys.argv[1]), 150, block_reads=4096)
enc = gpu.Encoder(0); enc.upload_reference(pb.ref)
ts = []
for i in range(6):
    out = enc.encode_blocks(pb); ts.append(enc.last_kernel_ms())
res = out[1]
print("RESULT", min(ts[1:]), sum(int(x) for x in res["nbytes"]), int((res["status"] != 0).sum()))
''' % R
for reads in (10_000_000, 1_048_576):
    for lib in sys.argv[1:]:
        env = dict(os.environ, CBC_GPU_LIB=os.path.join(R, "scratch", "abl", lib))
        r = subprocess.run([sys.executable, "-c", code, str(reads)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("%-28s reads %9d  kernel ms, payload bytes, failed blocks: %s" % (lib, reads, line[0][7:] if line else r.stderr[-300:]), flush=True)
