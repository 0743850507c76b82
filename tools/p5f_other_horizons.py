"""PlanarP5fMPC at horizons without a build-time specialisation (N = 5: no cut; N = 20: both chains cut by qpstruct.bisect_ordering):
the table kernel in fp32 next to the wave kernel in fp64 on the same states, 20 warm-started ticks. usage: python tools/p5f_other_horizons.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from robobee3d_amd.batchqp import PlanarP5fMPC
for N in (5, 20):
    B = 96
    a, b = PlanarP5fMPC(B, torch.float32, N=N), PlanarP5fMPC(B, torch.float64, N=N)
    y0 = np.random.default_rng(N).normal(size=(7, B)) * 0.05
    a.y.copy_(torch.as_tensor(y0).to(a.y)); b.y.copy_(torch.as_tensor(y0).to(b.y))
    worst = 0.0
    for ti in range(2, 22):
        a.tick(0.002 * ti); b.tick(0.002 * ti)
        xa, xb = a.solution().double(), b.solution()
        worst = max(worst, float((xa - xb).abs().max() / max(1.0, float(xb.abs().max()))))
        b.y.copy_(a.y.double())
    torch.cuda.synchronize()
    print("N = %d: %s (fp32) next to %s (fp64), 20 ticks, solved %.3f / %.3f, max |dx| / max(1, |x|) %.2e" % (
        N, a.qp.kernel_name, b.qp.kernel_name, float((a.qp.status == 1).float().mean()), float((b.qp.status == 1).float().mean()), worst))
    assert worst < 1e-3
