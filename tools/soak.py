"""Soak run of the assembly paths (not a test: a longer exercise at several batch sizes, looking for rare faults):
fp32 all-assembly rollout, fp64 assembly-loop rollout (incl. batches beyond one wave per CU), p5f ticks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
from robobee3d_amd.batchqp import PlanarP5fMPC
t0 = time.time()
for dt, ndt, sizes, K in ((torch.float32, np.float32, (1, 63, 4096, 65536, 100000), 300), (torch.float64, np.float64, (1, 100, 4096, 20000, 40000), 100)):
    for B in sizes:
        st, ref = hover_initial_conditions(B, 11, ndt)
        m = BatchUprightMPC(B, dt, plant_mode=1 if dt == torch.float32 else 0)
        m.set_state(st, ref)
        for _ in range(3):
            m.rollout(K)
        torch.cuda.synchronize()
        s = m.state.cpu().numpy()
        stt = m.status.cpu().numpy()
        assert np.isfinite(s).all(), (dt, B)
        assert np.abs(s[0:3]).max() < 50.0, (dt, B, np.abs(s[0:3]).max())
        print("uprightmpc2 %s B=%d 3x%d steps ok: |p|max %.3g mm, solved %.3f  (%.1fs)" % (str(dt)[6:], B, K, np.abs(s[0:3]).max(), (stt > 0).mean(), time.time() - t0), flush=True)
for B in (1, 5, 64, 1000, 16384, 40000):
    mpc = PlanarP5fMPC(B, torch.float32)
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
    for ti in range(2, 152):
        mpc.tick(0.002 * ti)
    torch.cuda.synchronize()
    y = mpc.y.cpu().numpy()
    assert np.isfinite(y).all() and np.isfinite(mpc.qp.sol_x.cpu().numpy()).all(), B
    print("p5f %s B=%d 150 ticks ok: solved %.3f, |y|max %.3g  (%.1fs)" % (mpc.qp.kernel_name, B, (mpc.qp.status.cpu().numpy() > 0).mean(), np.abs(y).max(), time.time() - t0), flush=True)
print("soak ok")
