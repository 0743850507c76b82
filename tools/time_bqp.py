"""Timing of the table-driven general QP kernel on the uprightmpc2 structure (diagnostic)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from robobee3d_amd.batchqp import UprightMPC2N
from robobee3d_amd.batch import hover_initial_conditions
for N, B in ((3, 65536), (3, 16384), (5, 16384), (10, 16384)):
    st, ref = hover_initial_conditions(B, 20201118, np.float32)
    mpc = UprightMPC2N(B, N, dtype=torch.float32)
    S, R = torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda()
    mpc.update(S, R); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        mpc.update(S, R)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(mpc.qp.kernel_name, "N=%d B=%d nnzL=%d: %.2f ms/step, %.3g robot-steps/s, solved %.3f" % (N, B, mpc.qp.s.nnzL, dt * 1e3, B / dt,
          float((mpc.qp.status > 0).float().mean())), flush=True)
