import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
from robobee3d_amd.batchqp import PlanarP5fMPC
t0 = time.time()
for dt, ndt, B, K, reps in ((torch.float32, np.float32, 65536, 500, 6), (torch.float64, np.float64, 4096, 500, 4), (torch.float64, np.float64, 300, 500, 2)):
    st, ref = hover_initial_conditions(B, 13, ndt)
    m = BatchUprightMPC(B, dt, plant_mode=1 if dt == torch.float32 else 0)
    m.set_state(st, ref)
    for _ in range(reps):
        m.rollout(K)
    torch.cuda.synchronize()
    s = m.state.cpu().numpy(); stt = m.status.cpu().numpy()
    assert np.isfinite(s).all() and np.abs(s[0:3]).max() < 50.0, (dt, B)
    print("uprightmpc2 %s B=%d %dx%d steps ok: |p|max %.3g mm, solved %.3f (%.1fs)" % (str(dt)[6:], B, reps, K, np.abs(s[0:3]).max(), (stt > 0).mean(), time.time() - t0), flush=True)
for B in (16384, 200):
    mpc = PlanarP5fMPC(B, torch.float32)
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
    n, m_ = mpc.qp.n, mpc.qp.m
    for ti in range(2, 1502):
        mpc.tick(0.002 * ti)
        if ti % 500 == 0:
            assert mpc.qp.y[m_ - n:].abs().max().item() == 0.0      # loose rows' multipliers: exactly zero all along (y0 body)
    torch.cuda.synchronize()
    y = mpc.y.cpu().numpy()
    assert np.isfinite(y).all() and np.isfinite(mpc.qp.sol_x.cpu().numpy()).all(), B
    print("p5f %s B=%d 1500 ticks ok: solved %.3f, |y|max %.3g (%.1fs)" % (mpc.qp.kernel_name, B, (mpc.qp.status.cpu().numpy() == 1).mean(), np.abs(y).max(), time.time() - t0), flush=True)
print("soak2 ok")
