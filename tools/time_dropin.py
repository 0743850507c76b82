"""Latency of the B = 1 drop-in (umpcInit / umpcUpdate through robobee3d_amd.uprightmpc2py), diagnostic."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from robobee3d_amd.uprightmpc2py import createMPC
_, c = createMPC()  # (pyver, cver)
p, R, dq = np.zeros(3), np.eye(3), np.zeros(6)
pdes, dpdes, sdes = np.zeros(3), np.zeros(3), np.array([0, 0, 1.0])
for _ in range(20):
    c.update(p, R, dq, pdes, dpdes, sdes)
t0 = time.perf_counter()
n = 500
for _ in range(n):
    u, acc = c.update(p, R, dq, pdes, dpdes, sdes)
print("umpcUpdate drop-in: %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6), u)
