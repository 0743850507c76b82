"""Where does the per-launch cost of the step kernel go? Times K-step launches back to back (no idle gap) and after a
host synchronisation + idle gap, B = 65536 fp32 RK4 (diagnostic)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
B = 65536
st, ref = hover_initial_conditions(B, 20201118, np.float32)
m = BatchUprightMPC(B, torch.float32, plant_mode=1)
m.set_state(st, ref)
m.rollout(200); torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
for K in (1, 2, 5, 10, 20, 50, 100):
    # back to back: 6 launches, time the last 4
    es = [ev() for _ in range(7)]
    for i in range(6):
        es[i].record(); m.rollout(K)
    es[6].record(); torch.cuda.synchronize()
    b2b = np.mean([es[i].elapsed_time(es[i + 1]) for i in range(2, 6)])
    # after an idle gap
    gaps = []
    for _ in range(4):
        torch.cuda.synchronize(); time.sleep(0.05)
        a, b = ev(), ev(); a.record(); m.rollout(K); b.record(); torch.cuda.synchronize()
        gaps.append(a.elapsed_time(b))
    print("K=%3d back-to-back %.4f ms/launch = %.4f ms/step | after 50 ms idle %.4f ms/launch = %.4f ms/step" %
          (K, b2b, b2b / K, np.mean(gaps), np.mean(gaps) / K))
