"""config 4 (p5f fp32, B = 16 384): per-phase time of one tick's QP step. Needs a library GENERATED with UMPC_QP_TIMING=1
(the assembly specialisation then writes six intervals, 100 MHz ticks, over the info rows):
  UMPC_QP_TIMING=1 python -c "from robobee3d_amd import _lib; _lib.build()"  ... run ...  then rebuild without it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robobee3d_amd.batchqp import PlanarP5fMPC
B = 16384
mpc = PlanarP5fMPC(B, torch.float32)
mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
for ti in range(2, 8):
    mpc.tick(0.002 * ti)
torch.cuda.synchronize()
t = mpc.qp.info.cpu().numpy().astype(np.float64)
names = ("load + classify + Ruiz", "factor", "first iteration (C++)", "hand-off stores", "assembly (48 + 1 iterations)", "reload + residuals + stores")
print(mpc.qp.kernel_name, "us per phase (mean over robots):", ", ".join("%s %.0f" % (n, r.mean() / 100.0) for n, r in zip(names, t)), "| total %.0f" % (t.sum(0).mean() / 100.0))
