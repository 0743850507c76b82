"""Where do the 0.23 ms of a B = 1 umpcUpdate go? (diagnostic)  (a) the drop-in call as is; (b) the B = 1 step kernel alone,
back-to-back launches timed with events (no host in the loop); (c) the drop-in call while ANOTHER stream keeps the chip
under load (a 65 536-robot rollout): if (c) is faster than (a), the lone wave of (a) runs at the idle chip's low clock."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from robobee3d_amd.uprightmpc2py import createMPC
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
_, c = createMPC()
p, R, dq = np.zeros(3), np.eye(3), np.zeros(6)
pdes, dpdes, sdes = np.zeros(3), np.zeros(3), np.array([0, 0, 1.0])


def dropin(n=400):
    for _ in range(20):
        c.update(p, R, dq, pdes, dpdes, sdes)
    t0 = time.perf_counter()
    for _ in range(n):
        c.update(p, R, dq, pdes, dpdes, sdes)
    return (time.perf_counter() - t0) / n * 1e6
print("(a) umpcUpdate drop-in, idle chip: %.1f us per call" % dropin())
for B in (1, 64):
    st, ref = hover_initial_conditions(B, 1, np.float32)
    m = BatchUprightMPC(B, torch.float32, nsub=0)
    m.set_state(st, ref)
    for _ in range(20):
        m.update()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        m.update()
    b.record(); torch.cuda.synchronize()
    print("(b) B = %d step kernel alone, 200 back-to-back launches: %.1f us per launch" % (B, a.elapsed_time(b) * 1e3 / 200))
big = BatchUprightMPC(65536, torch.float32, plant_mode=1)
st, ref = hover_initial_conditions(65536, 1, np.float32)
big.set_state(st, ref)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    big.rollout(3000)          # ~0.37 s of load on the other stream
time.sleep(0.05)
print("(c) umpcUpdate drop-in while a 65 536-robot rollout loads the chip: %.1f us per call" % dropin(300))
torch.cuda.synchronize()
