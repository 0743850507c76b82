"""Latency of the B = 1 drop-in through the PYTHON bindings of umpcInit / umpcUpdate (robobee3d_amd.uprightmpc2py): the compiled
module (csrc/uprightmpc2py_ext.cpp, pybind11 -- what `from uprightmpc2py import UprightMPC2C` names) and the ctypes classes,
one process, float64 numpy arguments as the reference's harness passes them (template/uprightmpc2.py:139). tools/time_dropin_c.py
times the C boundary itself (arguments marshalled once)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from robobee3d_amd import uprightmpc2py as w
prm = (5, 9.81e-3, 2, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2)
Ib = np.array([3333.0, 3333.0, 1000.0])
p, R, dq = np.zeros(3), np.eye(3), np.zeros(6)
pdes, dpdes, sdes = np.zeros(3), np.zeros(3), np.array([0, 0, 1.0])
print("active binding:", w.binding())
for name, cls in (("compiled (pybind11)", w.UprightMPC2C), ("ctypes", w.UprightMPC2C_ctypes), ("compiled (pybind11)", w.UprightMPC2C),
                  ("ctypes", w.UprightMPC2C_ctypes)):
    c = cls(*prm, Ib, 50)
    # (a process's first ~100 calls contain a one-off stall of ~40 ms -- runtime / power-state first use: time the steady state)
    for _ in range(600):
        c.update(p, R, dq, pdes, dpdes, sdes)
    n = 3000
    t0 = time.perf_counter()
    for _ in range(n):
        u, acc = c.update(p, R, dq, pdes, dpdes, sdes)
    print("umpcUpdate through the %-20s binding: %.1f us per call" % (name, (time.perf_counter() - t0) / n * 1e6), u)
    del c
