"""Is the short-launch penalty of the step kernel the launch SHAPE or the chip's clocks? A K-step launch is timed right
behind a 300-step launch (clocks at their loaded steady state, no idle gap), and again after 50 ms of idle.
B = 65536 fp32 RK4 (diagnostic)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
B = 65536
st, ref = hover_initial_conditions(B, 20201118, np.float32)
m = BatchUprightMPC(B, torch.float32, plant_mode=1)
m.set_state(st, ref)
m.rollout(300); torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
for K in (1, 2, 5, 10, 20, 50, 100):
    hot, hot2 = [], []
    for _ in range(3):
        a, b, c = ev(), ev(), ev()
        m.rollout(300); a.record(); m.rollout(K); b.record(); m.rollout(K); c.record(); torch.cuda.synchronize()
        hot.append(a.elapsed_time(b)); hot2.append(b.elapsed_time(c))
    cold = []
    for _ in range(3):
        torch.cuda.synchronize(); time.sleep(0.05)
        a, b = ev(), ev(); a.record(); m.rollout(K); b.record(); torch.cuda.synchronize()
        cold.append(a.elapsed_time(b))
    print("K=%3d behind a 300-step launch %.4f ms/step, the next one %.4f | after 50 ms idle %.4f ms/step" %
          (K, np.mean(hot) / K, np.mean(hot2) / K, np.mean(cold) / K))
# how long does the ramp take? one long launch after idle, split into consecutive 10-step launches
torch.cuda.synchronize(); time.sleep(0.1)
es = [ev() for _ in range(41)]
for i in range(40):
    es[i].record(); m.rollout(10)
es[40].record(); torch.cuda.synchronize()
print("after 100 ms idle, consecutive 10-step launches (ms/step):", " ".join("%.4f" % (es[i].elapsed_time(es[i + 1]) / 10) for i in range(40)))
