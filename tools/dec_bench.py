import sys, os, ctypes, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import numpy as np, torch
from cbc_amd import host, gpu
N=int(sys.argv[1]) if len(sys.argv)>1 else 10_000_000
pb = host.synth(0xCBC00002, 248956422, N, 150, block_reads=4096)
enc = gpu.Encoder(0); L=gpu.lib(); dev=torch.device('cuda',0)
enc.upload_reference(pb.ref)
t=time.time(); payloads, res, offs, flat = enc.encode_blocks(pb); print('host-path encode wall %.3f s'%(time.time()-t), 'kernel ms', enc.last_kernel_ms())
assert (res['status']==0).all()
stride=152; nrec=pb.n_recs
blocks = np.zeros(pb.n_blocks, dtype=host.DEC_BLOCK_DTYPE)
nb0=0
for b in range(pb.n_blocks):
    blocks[b]['in_off']=int(offs[b]); blocks[b]['in_bytes']=int(offs[b+1]-offs[b]); blocks[b]['ref_off']=int(pb.blocks[b]['ref_off'])
    blocks[b]['rec_base']=nb0; blocks[b]['seq_base']=nb0*stride; blocks[b]['n_reads']=int(pb.blocks[b]['n_reads']); blocks[b]['read_length']=150; blocks[b]['seq_stride']=stride
    nb0+=int(pb.blocks[b]['n_reads'])
td=lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
d_in=td(np.concatenate([flat, np.zeros(16,dtype=np.uint8)])); d_blocks=td(blocks); d_ref=td(pb.ref)
d_recs=torch.zeros(nrec*16,dtype=torch.uint8,device=dev); d_seq=torch.zeros(nrec*stride+16,dtype=torch.uint8,device=dev); d_res=torch.zeros(pb.n_blocks*16,dtype=torch.uint8,device=dev)
d_vs=torch.zeros(pb.n_blocks*pb.cap_var, dtype=torch.int32, device=dev)
NB=int(os.environ.get('DEC_NBLK', pb.n_blocks))     # decode only the first NB blocks of the same data (contention curve)
db=gpu.DecDeviceBatch(d_in.data_ptr(), d_in.numel(), d_blocks.data_ptr(), NB, d_ref.data_ptr(), d_ref.numel(), d_recs.data_ptr(), nrec, d_seq.data_ptr(), d_seq.numel(), d_res.data_ptr(), d_vs.data_ptr(), d_vs.numel(), host.LdsCaps(pb.cap_pos,pb.cap_var))
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ms=[]
for i in range(4):
    enc.decode_device(db, st); torch.cuda.synchronize(); ms.append(enc.last_kernel_ms())
r=d_res.cpu().numpy().view(host.RESULT_DTYPE); assert (r['status'][:NB]==0).all()
allseq=d_seq.cpu().numpy()
if os.environ.get('DEC_STAMP'):
    sums=np.zeros(16)
    for b in range(pb.n_blocks):
        o=int(blocks[b]['seq_base']); sums+=allseq[o:o+128].view(np.uint64).astype(np.float64)
    names=['same_ref(+name)','rlength x4','pos','flag','match','edits: after the last SNP / perfect copy','record store','edits: SNP count, to each SNP','win_first','var_dec: events appended','chars+patch','var_dec: class, bucket, events','var_dec: filter, global list','var_dec: target, search','var_dec: step']
    for n_,v in zip(names,sums): print('%-16s %8.0f cycles/read %5.1f%%'%(n_, v/nrec, 100*v/sums.sum()))
    print('total', sums.sum()/nrec)
else:
    nr=int(blocks[NB-1]['rec_base'])+int(blocks[NB-1]['n_reads'])
    got=allseq[:nr*stride].reshape(nr,stride)[:,:150]; assert (got==pb.seq[:nr*150].reshape(nr,150)).all()
caps=host.LdsCaps(pb.cap_pos,pb.cap_var)
print('blocks', NB, 'decode kernel ms', ms, 'Mbases/s', pb.n_bases/ (min(ms)*1e-3)/1e6, 'lds', L.cbc_gpu_decode_lds_bytes(ctypes.byref(caps)))
