"""fp64 assembly ADMM block, inside view (library built with UMPC_ASM64_TIMING=1 and -DUMPC_PHASE_TIMING -DUMPC_ASM64_TIMING):
out rows 0..2 = the block's prologue, iterations, epilogue in 100 MHz ticks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
for iters in (50, 1):
    B = 4096
    st, ref = hover_initial_conditions(B, 5, np.float64)
    m = BatchUprightMPC(B, torch.float64, plant_mode=0, maxIter=iters)
    m.set_state(st, ref)
    m.rollout(5)
    m.rollout(1)
    torch.cuda.synchronize()
    t = m.out.cpu().numpy().astype(np.float64)
    print("maxIter %d: block prologue %.1f us, iterations %.1f us, epilogue %.1f us | phases: %s" %
          (iters, t[0].mean() / 100, t[1].mean() / 100, t[2].mean() / 100, ", ".join("%.1f" % (r.mean() / 100) for r in t[3:9])))
