import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, ctypes as C
import me_cases
from svt_av1_psyex_amd import api, abi
api.LIB_PATH=os.path.join(os.path.dirname(os.path.abspath(__file__)),'_libprof.so')
import torch
from me_cases import MeCase
ctx=api.Context()
names=['fetch','setup','-','-','searches-tail','post','pre-zz','pre-prehme','pre-L0','pre-L1','pre-L2','pre-c00','pre-probe','prune','cands','dist','out-tail']
for dist in (1,):
    c=MeCase(3840,2160,enc_mode=6,cur=8,refs={(0,0):8-dist,(1,0):8+dist},n_frames=17,seed=11,temporal_layer_index={1:4,8:1}[dist])
    cur=ctx.upload(c.cur); refs={k:ctx.upload(v) for k,v in c.refs.items()}
    out=(C.c_ulonglong*24)()
    ctx.me_picture(c.cfg,c.desc,cur,refs); api.lib().svt_hip_me_profile_read(ctx._h,out)
    for _ in range(3): ctx.me_picture(c.cfg,c.desc,cur,refs)
    api.lib().svt_hip_me_profile_read(ctx._h,out)
    v=np.array(out[:17],dtype=np.float64)/3/2040
    w=np.array(out[17:24],dtype=np.float64)/3/2040
    print('   run_searches split: plan-bar',int(w[0]),'stage',int(w[1]),'eval-keys',int(w[2]),'eval-items',int(w[4]),'eval-bar',int(w[5]),'plan',int(w[6]),'pre-main',int(w[3]))
    print('dist',dist,'cycles per b64 (100MHz ticks?):', {n:int(x) for n,x in zip(names,v)}, 'total',int(v.sum()))
