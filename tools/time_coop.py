"""how long does ONE robot's controller step take on the wave-cooperative kernel (bqp_wave_kernel: one
wavefront per robot, the working set in LDS, element-wise phases 64 unknowns per instruction, LDL' and both triangular solves
level-scheduled over the elimination tree) against the lane-per-robot all-assembly step (one lane of one wave)? Kernel time
per launch from events, back-to-back launches, N = 3 uprightmpc2 structure, 50 iterations, fp32. (diagnostic)"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
from robobee3d_amd.batchqp import UprightMPC2N


def ev_time(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for B in (1, 4):
    st, ref = hover_initial_conditions(B, 1, np.float32)
    m = BatchUprightMPC(B, torch.float32, nsub=0)
    m.set_state(st, ref)
    print("B = %d lane-per-robot all-assembly step (%s): %.1f us per launch" % (B, "umpc_rollout_asm_kernel", ev_time(m.update)))
    S, R = torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda()
    for mode in ("wave", "tables"):
        g = UprightMPC2N(B, 3, dtype=torch.float32)
        g.qp.set_kernel(mode)
        g.assemble(S, R)
        solve = lambda: g.qp.solve(g.Pv, g.Av, g.q, g.l, g.u)
        print("B = %d general solver, kernel %-7s (QP solve only: Ruiz, LDL', 50 iterations, residuals): %.1f us per launch"
              % (B, g.qp.kernel_name, ev_time(solve)))
