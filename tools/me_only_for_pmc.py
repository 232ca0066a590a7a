import sys; sys.path.insert(0,'.')
sys.path.insert(0,'tests')
import me_cases
from me_cases import MeCase
from svt_av1_psyex_amd import api
ctx=api.Context()
c=MeCase(3840,2160,enc_mode=6,cur=8,refs={(0,0):6,(1,0):10},n_frames=17,seed=11,temporal_layer_index=3)
cur=ctx.upload(c.cur); refs={k:ctx.upload(v) for k,v in c.refs.items()}
for _ in range(3): ctx.me_picture(c.cfg,c.desc,cur,refs,search_level=False)
print('done')
