# per-section s_memtime sums of the long-read encoder's model and coder wavefronts from a -DCBC_STAMP build (scratch/abl/lib_stamp.so):
#   hipcc -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -DCBC_STAMP -shared -o scratch/abl/lib_stamp.so cbc_amd/csrc/cbc_gpu.hip
import sys, os, ctypes
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
os.environ['CBC_GPU_LIB']=R+'/scratch/abl/'+os.environ.get('STAMP_LIB','lib_stamp.so')
import numpy as np, torch
from cbc_amd import host, gpu
NR=int(sys.argv[1]) if len(sys.argv)>1 else 200_000
pb = host.synth_long(0xCBC00005, NR*100, NR, 10_000, 0.05, b"chrL", block_reads=64)
enc = gpu.Encoder(0); L=gpu.lib(); dev=torch.device('cuda',0)
blocks = pb.blocks.copy()
scratch = int(L.cbc_gpu_long_plan_output(blocks.ctypes.data, pb.n_blocks, pb.recs.ctypes.data, 8))
td=lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
d=[td(x) for x in (pb.recs,pb.seq,pb.tok,pb.names,blocks,pb.ref)]
d_out=torch.zeros(scratch,dtype=torch.uint8,device=dev); d_res=torch.zeros(pb.n_blocks*16,dtype=torch.uint8,device=dev)
db=gpu.DeviceBatch(d[0].data_ptr(),d[1].data_ptr(),d[2].data_ptr(),d[3].data_ptr(),d[4].data_ptr(),pb.n_blocks,d[5].data_ptr(),d[5].numel(),d_out.data_ptr(),scratch,d_res.data_ptr(),d[1].numel(),max(pb.n_tok,1),pb.n_recs,host.LdsCaps(pb.cap_pos,pb.cap_var))
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(2): enc.encode_long_device(db, st)
torch.cuda.synchronize(); print(os.environ.get('STAMP_LIB','lib_stamp.so'), 'kernel ms', enc.last_kernel_ms(), 'blocks', pb.n_blocks, 'Gbases/s', pb.n_bases/enc.last_kernel_ms()/1e6)
if 'stamp' not in os.environ.get('STAMP_LIB','lib_stamp.so'): sys.exit(0)
out=d_out.cpu().numpy()
cod=np.zeros(16); mod=np.zeros(16)
for b in range(pb.n_blocks):
    o=int(blocks[b]['out_off']); cod+=out[o:o+128].view(np.uint64).astype(np.float64); mod+=out[o+128:o+256].view(np.uint64).astype(np.float64)
nb=pb.n_blocks
mn={0:'reading a batch back',1:'batch model arithmetic',2:'record header',3:'tail',4:'WAITING for the walker',5:'until the hand-off wait',6:'WAITING for a free slot',7:'hand-over of triples'}
cn={9:'coding',10:'WAITING for a batch'}
print('model wavefront, s_memtime ticks per block:')
for k,v in enumerate(mod):
    if v: print('  %-28s %12.0f ticks  %5.1f%%'%(mn.get(k,str(k)), v/nb, 100*v/mod.sum()))
print('coder wavefront:')
for k,v in enumerate(cod):
    if v: print('  %-28s %12.0f ticks  %5.1f%%'%(cn.get(k,str(k)), v/nb, 100*v/cod.sum()))
