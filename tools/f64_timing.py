"""config 2 (fp64, B = 4 096): per-phase time of one step, from a library built with -DUMPC_PHASE_TIMING
(UMPC_VARIANT_DEFS="-DUMPC_PHASE_TIMING" tools/build_variant.py timing; run with UMPC_LIB=.../libumpc_timing.so).
The kernel writes the six intervals (100 MHz ticks) over accdes in out rows 3..8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
for dt, name, B in ((torch.float64, "fp64", 4096), (torch.float32, "fp32 C++ kernel", 4096)):
    st, ref = hover_initial_conditions(B, 5, np.float64 if dt == torch.float64 else np.float32)
    m = BatchUprightMPC(B, dt, plant_mode=0)
    if dt == torch.float32:
        m.set_step_kernel("cpp")
    m.set_state(st, ref)
    m.rollout(5)
    m.rollout(1)
    torch.cuda.synchronize()
    ticks = m.out.cpu().numpy()[3:9].astype(np.float64)
    names = ("assemble", "Ruiz", "D/E + factor", "ADMM", "residuals/extraction", "plant")
    print(name, "us per phase (mean over robots):", ", ".join("%s %.1f" % (n, t.mean() / 100.0) for n, t in zip(names, ticks)),
          "| total %.1f" % (ticks.sum(0).mean() / 100.0))
