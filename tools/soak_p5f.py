"""Longer exercise of config 4's shared-workgroup kernel (four wavefronts per group of 64 robots, asmqp.*_group_program): 1 000
warm-started ticks at the benchmarked size and on a ragged batch, the fp32 assembly route next to the fp64 general kernel on
the same states every 100 ticks. usage: python tools/soak_p5f.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from robobee3d_amd.batchqp import PlanarP5fMPC
t0 = time.time()
for B in (16384, 77, 1):
    a, b = PlanarP5fMPC(B, torch.float32), PlanarP5fMPC(B, torch.float64)
    y0 = np.random.default_rng(B).normal(size=(7, B)) * 0.05
    a.y.copy_(torch.as_tensor(y0).to(a.y))
    b.y.copy_(torch.as_tensor(y0).to(b.y))
    n, m_ = a.qp.n, a.qp.m
    worst = 0.0
    for ti in range(2, 1002):
        a.tick(0.002 * ti)
        b.tick(0.002 * ti)
        if ti % 100 == 1:
            assert a.qp.kernel_name == "p5f10+asm"
            assert a.qp.y[m_ - n:].abs().max().item() == 0.0          # loose rows' multipliers: exactly zero all along (y0 body)
            xa, xb = a.solution().double(), b.solution()
            assert torch.isfinite(xa).all() and (a.qp.status == 1).all() and (b.qp.status == 1).all(), (B, ti)
            d = float((xa - xb).abs().max() / max(1.0, float(xb.abs().max())))
            worst = max(worst, d)
            assert d < 2e-4, (B, ti, d)
            b.y.copy_(a.y.double())                                      # (keep the plants on the same fp32 states)
    torch.cuda.synchronize()
    print("p5f B=%d: 1000 ticks on %s next to the fp64 %s kernel, every robot solved at every check, max |dx| / max(1, |x|) %.2e (%.1fs)"
          % (B, a.qp.kernel_name, b.qp.kernel_name, worst, time.time() - t0), flush=True)
print("soak p5f ok")
