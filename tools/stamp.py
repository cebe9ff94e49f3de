import sys, os, ctypes
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
os.environ['CBC_GPU_LIB']=R+'/scratch/abl/'+os.environ.get('STAMP_LIB','lib_stamp.so')
import numpy as np, torch
from cbc_amd import host, gpu
NR=int(sys.argv[1]) if len(sys.argv)>1 else 10_000_000
pb = host.synth(0xCBC00002, int(NR*24.9), NR, 150, block_reads=4096)
enc = gpu.Encoder(0); L=gpu.lib(); dev=torch.device('cuda',0)
blocks = pb.blocks.copy()
scratch = int(L.cbc_gpu_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, pb.tok.ctypes.data))
td=lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
d=[td(x) for x in (pb.recs,pb.seq,pb.tok,pb.names,blocks,pb.ref)]
d_out=torch.zeros(scratch,dtype=torch.uint8,device=dev); d_res=torch.zeros(pb.n_blocks*16,dtype=torch.uint8,device=dev)
db=gpu.DeviceBatch(d[0].data_ptr(),d[1].data_ptr(),d[2].data_ptr(),d[3].data_ptr(),d[4].data_ptr(),pb.n_blocks,d[5].data_ptr(),d[5].numel(),d_out.data_ptr(),scratch,d_res.data_ptr(),d[1].numel(),pb.n_tok,pb.n_recs,host.LdsCaps(pb.cap_pos,pb.cap_var))
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(2): enc.encode_device(db, st)
torch.cuda.synchronize(); print('kernel ms', enc.last_kernel_ms())
out=d_out.cpu().numpy()
sums=np.zeros(16)
for b in range(pb.n_blocks):
    o=int(blocks[b]['out_off'])+int(blocks[b]['out_cap'])-128; sums+=out[o:o+128].view(np.uint64).astype(np.float64)
    o=int(blocks[b]['out_off']); sums+=out[o:o+128].view(np.uint64).astype(np.float64)
names=['M: grp loads+match','M: seg_end','-','M: to the first win_first of a record / between SNPs','M: win_first','M: until publish','M: at barrier','M: var_code','M: win_set+chars','C: coding','C: waiting','M: publish, next-record prefetch, window slide, token header','M: edit counts','M: rest of edits()','-','-']
tot=sums.sum()
for n,v in zip(names,sums): print('%-12s %8.0f cycles/read  %5.1f%%'%(n, v/pb.n_recs, 100*v/tot))
print('total stamped cycles/read', tot/pb.n_recs)
