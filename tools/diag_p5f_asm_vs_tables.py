"""GPU diagnostic: the p5f assembly kernel against the table kernel after ONE cold-started solve of k iterations, variable by
variable (which unknowns differ first, and by how much) -- used when a generator change passes the interpreter but not the GPU."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from robobee3d_amd.batchqp import PlanarP5fMPC

B = 64
for iters in (1, 2, 3, 10, 50):
    res = {}
    for mode in ("lane", "tables"):
        mpc = PlanarP5fMPC(B, torch.float32, max_iter=iters)
        mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
        mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
        mpc.qp.set_kernel(mode)
        mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * 0.002 * 2))
        mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
        torch.cuda.synchronize()
        res[mode] = [t.cpu().numpy().astype(np.float64) for t in (mpc.qp.x, mpc.qp.y, mpc.qp.z, mpc.qp.Eprev)]
        name = mpc.qp.kernel_name
    print("iters", iters, "kernel", name)
    for nm, a, b in zip(("x", "y", "z", "E"), res["lane"], res["tables"]):
        d = np.abs(a - b).max(1) / np.maximum(1.0, np.abs(b).max(1))
        bad = np.nonzero(d > 1e-4)[0]
        print("  %s: worst %.3e, %d of %d entries off:" % (nm, d.max(), len(bad), len(d)), bad[:40].tolist())
