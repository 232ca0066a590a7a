import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from me_cases import MeCase
from svt_av1_psyex_amd import api, abi
ctx=api.Context()
ext=torch.cuda.ExternalStream(ctx.stream)
w,h,dist=int(sys.argv[1]),int(sys.argv[2]),2
c=MeCase(w,h,enc_mode=6,cur=8,refs={(0,0):8-dist,(1,0):8+dist},n_frames=17,seed=11,temporal_layer_index=3)
cur=ctx.upload(c.cur); refs={k:ctx.upload(v) for k,v in c.refs.items()}
nb=((w+63)//64)*((h+63)//64); n=abi.n_pu(c.desc.enable_me_16x16,c.desc.enable_me_8x8)
res=abi.MeResults(); keep=[]
for name,dt,cnt in abi.RESULT_FIELDS:
    if name in ('hme_sc','hme_sad','do_ref'): continue
    t_=torch.zeros(nb*cnt(n,c.desc.max_refs,c.desc.max_cand)*np.dtype(dt).itemsize,dtype=torch.uint8,device='cuda'); keep.append(t_); setattr(res,name,t_.data_ptr())
torch.cuda.synchronize()
with torch.cuda.stream(ext):
    for _ in range(5): ctx.me_picture_async(c.cfg,c.desc,cur,refs,res)
ctx.sync(); torch.cuda.synchronize()
