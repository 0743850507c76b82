import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import numpy as np, torch
from robobee3d_amd.batch import BatchUprightMPC
from robobee3d_amd import _lib
import oraclebind
from test_gpu_parity import _load_seq_into
perm=np.array(_lib.lib().umpcKKTPerm().contents)
for fn in sys.argv[1:]:
    seq=np.load(os.path.join(ROOT,'tests/golden',fn)); n=len(seq['p0'])
    mpc=BatchUprightMPC(n, torch.float32, maxIter=int(seq['maxIter']))
    _load_seq_into(mpc, seq, torch); mpc.update(); torch.cuda.synchronize()
    out=mpc.out.cpu().numpy(); ctrl=mpc.ctrl.cpu().numpy(); st=mpc.status.cpu().numpy()
    o=oraclebind.Oracle(np.float32,perm=perm,maxIter=int(seq['maxIter']))
    for k in range(n):
        o.set_canonical(True, seq["pre_E3"][k]); o.set_iterates(seq["pre_x"][k], seq["pre_y"][k], seq["pre_z"][k]); o.set_T0(float(seq["pre_T0"][k]))
        uq,ac=o.update(seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k], seq["sdes"][k], float(seq["actualT0"][k]))
        dt=np.abs(out[1:3,k]-uq[1:]).max(); dx=np.abs(ctrl[:45,k]-o.get('x')).max(); dy=np.abs(ctrl[45:84,k]-o.get('y')).max(); dz=np.abs(ctrl[84:123,k]-o.get('z')).max()
        print(fn,k,'aT0 %.4f'%seq['actualT0'][k],'status gpu %d ora %d ref %d'%(st[k],o.get('status_val')[0],seq['status'][k]),'dtau %.3e dx %.3e dy %.3e dz %.3e'%(dt,dx,dy,dz), 'tau',uq[1:], out[1:3,k])
