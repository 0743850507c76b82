"""config 4's shape with FINITE input limits (|u_k| <= 4: the inequality rows of the inputs are ordinary rows, so the workgroups
take the GENERAL variant of the loop block instead of the loose one): ms per tick next to the reference's problem (infinite
bounds). usage: python tools/p5f_finite_limits.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from robobee3d_amd.batchqp import PlanarP5fMPC
B = 16384
for ulim in (None, 4.0):
    mpc = PlanarP5fMPC(B, torch.float32)
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
    if ulim is not None:
        rows = [77 + t for t, j in enumerate(mpc.st["var_order"]) if j >= 77]
        mpc.l[rows] = -ulim
        mpc.u[rows] = ulim
    for ti in range(2, 40):
        mpc.tick(0.002 * ti)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for ti in range(40 + 20 * rep, 60 + 20 * rep):
            mpc.tick(0.002 * ti)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print("input limits %s: %.4f ms per tick (%s), solved %.3f" % ("+-%g" % ulim if ulim else "infinite (the reference's problem)", best,
          mpc.qp.kernel_name, float((mpc.qp.status == 1).float().mean())), flush=True)
