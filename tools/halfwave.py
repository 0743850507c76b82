"""Does a wave with only its lower 32 (or 16) lanes active run the fp64 step faster? (one wave, pure latency; diagnostic)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
    for B in (8, 16, 32, 48, 64):
        st, ref = hover_initial_conditions(B, 1, np.float64 if dt == torch.float64 else np.float32)
        m = BatchUprightMPC(B, dt, plant_mode=0)
        m.set_state(st, ref)
        m.rollout(20); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m.rollout(40); b.record(); torch.cuda.synchronize()
        print("%s B=%2d (one wave, %2d active lanes): %.4f ms per step  [%s]" % (name, B, B, a.elapsed_time(b) / 40, m.kernel_name))
