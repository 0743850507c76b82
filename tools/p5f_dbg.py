"""debug driver: planar p5f ticks at several batch sizes, prints after every size"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robobee3d_amd.batchqp import PlanarP5fMPC
for B in [int(x) for x in sys.argv[1].split(",")]:
    mpc = PlanarP5fMPC(B, torch.float32)
    print("B", B, mpc.qp.kernel_name, flush=True)
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    for ti in range(2, 2 + int(sys.argv[2])):
        mpc.tick(0.002 * ti)
        torch.cuda.synchronize()
    x = mpc.qp.sol_x.cpu().numpy()
    print("  ok finite", np.isfinite(x).all(), "status>0", (mpc.qp.status.cpu().numpy() > 0).mean(), flush=True)
