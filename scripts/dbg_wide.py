import sys; sys.path[:0]=['low-cost-mocap_amd','.']
import numpy as np, torch, faulthandler
from mocap_core import capi, synth
core=capi.MocapCore(0)
rig=synth.stress_rig(64)
blobs,counts,_=synth.make_stress_stream(rig,256,256,seed=4242)
core.set_cameras(rig["K"],rig["R"],rig["t"])
res=core.match_triangulate(blobs,counts,gate_px=0.5,K_max=384,G_cap=1<<20)
print('plain ok', np.count_nonzero(res['status']), res['status'][res['status']!=0], flush=True)
bad=np.nonzero(res['status'])[0]
print('bad', bad, flush=True)
r2=core.match_triangulate_auto(blobs[bad],counts[bad],gate_px=0.5,K_max=384,G_cap=1<<20)
print('auto ok', r2['status'], r2['n_out'], r2['n_cand'], flush=True)
