"""Latency of the B = 1 drop-in (umpcInit / umpcUpdate through robobee3d_amd.uprightmpc2py), diagnostic."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from robobee3d_amd.uprightmpc2py import createMPC
_, c = createMPC()  # (pyver, cver)
p, R, dq = np.zeros(3), np.eye(3), np.zeros(6)
pdes, dpdes, sdes = np.zeros(3), np.zeros(3), np.array([0, 0, 1.0])
# (a process's first ~100 calls contain a one-off stall of ~40 ms -- runtime / power-state first use; round 2's 0.25 ms per call
# was 20 warm-up calls + that stall averaged over 500 calls: time the steady state)
t0 = time.perf_counter()
for _ in range(400):
    c.update(p, R, dq, pdes, dpdes, sdes)
print("first 400 calls of the process: %.1f us per call" % ((time.perf_counter() - t0) / 400 * 1e6))
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    u, acc = c.update(p, R, dq, pdes, dpdes, sdes)
print("umpcUpdate drop-in: %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6), u)
