import sys, time, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from me_cases import MeCase, compare
from svt_av1_psyex_amd import api, abi
ctx=api.Context()
ext=torch.cuda.ExternalStream(ctx.stream)

out=[os.path.basename(api.LIB_PATH)]
w,h=3840,2160
for dist in (1,2,8):
    c=MeCase(w,h,enc_mode=6,cur=8,refs={(0,0):8-dist,(1,0):8+dist},n_frames=17,seed=11,temporal_layer_index={1:4,2:3,4:2,8:1}[dist])
    cur=ctx.upload(c.cur); refs={k:ctx.upload(v) for k,v in c.refs.items()}
    if dist==2:
        b=ctx.me_picture(c.cfg,c.desc,cur,refs)
        a=c.run_cpu('oracle'); bad=compare(a,b); out.append('OK' if not bad else 'BAD %s'%bad[:2])
    nb=((w+63)//64)*((h+63)//64); n=abi.n_pu(c.desc.enable_me_16x16,c.desc.enable_me_8x8)
    res=abi.MeResults(); keep=[]
    for name,dt,cnt in abi.RESULT_FIELDS:
        if name in ('hme_sc','hme_sad','do_ref') or (os.environ.get('NO_SB_BEST') and name.startswith('sb_best')): continue
        t_=torch.zeros(nb*cnt(n,c.desc.max_refs,c.desc.max_cand)*np.dtype(dt).itemsize,dtype=torch.uint8,device='cuda'); keep.append(t_); setattr(res,name,t_.data_ptr())
    torch.cuda.synchronize()
    with torch.cuda.stream(ext):
        for _ in range(3): ctx.me_picture_async(c.cfg,c.desc,cur,refs,res)
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ctx.me_picture_async(c.cfg,c.desc,cur,refs,res)
        e1.record()
    ctx.sync(); torch.cuda.synchronize()
    out.append('d%d %.3f ms'%(dist,e0.elapsed_time(e1)/10))
print(*out, flush=True)
