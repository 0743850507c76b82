"""Latency of umpcUpdate at the C boundary: arguments marshalled ONCE (as bench.py's reference_anchor times the reference's
own umpcUpdate), so the figure is the library call, not the Python wrapper's numpy conversions (tools/time_dropin.py times
the wrapper). Also splits the call: UMPC_DROPIN_TWO_LAUNCHES=1 in a second process gives launch + synchronise."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from robobee3d_amd import _lib

L = _lib.lib()
up = _lib.UprightMPC_t()
f32 = lambda a: np.ascontiguousarray(a, np.float32)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
Ib = f32([3333.0, 3333.0, 1000.0])
L.umpcInit(C.byref(up), *[C.c_float(v) for v in (5.0, 9.81e-3, 2.0, 1e1, 1e3, 1.0, 5.0, 1e3, 2e3, 1e-1, 1e-2)], fp(Ib), C.c_int(50))
uq, ac = np.zeros(3, np.float32), np.zeros(6, np.float32)
rng = np.random.default_rng(1)
args = []
for a, b in rng.uniform(-0.5, 0.5, (64, 2)):
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    R = f32([cb, sa * sb, -ca * sb, 0, ca, sa, sb, -sa * cb, ca * cb])
    dq = f32([0.1, 0, 0, 0, 0, 0])
    keep = (R, dq, f32(np.zeros(3)), f32([0, 0, 1]))
    args.append((keep, (C.byref(up), fp(uq), fp(ac), fp(keep[2]), fp(R), fp(dq), fp(keep[2]), fp(keep[2]), fp(keep[3]), C.c_float(-1.0))))
f = L.umpcUpdate
for k in range(600):
    assert f(*args[k % 64][1]) == 0
n = 4000
t0 = time.perf_counter()
for k in range(n):
    f(*args[k % 64][1])
dt = (time.perf_counter() - t0) / n
print("umpcUpdate at the C boundary (arguments marshalled once): %.1f us per call; kernel %s" % (dt * 1e6, L.umpcKernelName(0, 1).decode()), uq)
