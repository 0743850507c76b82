"""The shared-workgroup p5f kernel beyond one workgroup per CU (B = 40 000 and 65 553: several rounds of 256-thread workgroups,
a ragged last one): ten warm-started ticks next to the fp64 general kernel on the same states. usage: python tools/p5f_big_batch.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from robobee3d_amd.batchqp import PlanarP5fMPC
for B in (40000, 65536 + 17):
    a, b = PlanarP5fMPC(B, torch.float32), PlanarP5fMPC(B, torch.float64)
    y0 = np.random.default_rng(B).normal(size=(7, B)) * 0.05
    a.y.copy_(torch.as_tensor(y0).to(a.y)); b.y.copy_(torch.as_tensor(y0).to(b.y))
    for ti in range(2, 12):
        a.tick(0.002 * ti); b.tick(0.002 * ti)
        b.y.copy_(a.y.double())
    xa, xb = a.solution().double(), b.solution()
    d = float((xa - xb).abs().max() / max(1.0, float(xb.abs().max())))
    print(B, a.qp.kernel_name, "solved", float((a.qp.status == 1).float().mean()), "max rel dx", d, flush=True)
    assert d < 2e-4 and (a.qp.status == 1).all()
print("big ok")
