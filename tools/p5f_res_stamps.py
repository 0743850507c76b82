"""config 4 (p5f fp32, B = 16 384): where the time of the residual block / of the Ruiz + glue blocks goes. Needs a library
GENERATED with UMPC_QP_RES_STAMPS=1 (UMPC_QP_RES_STAMPS=1 tools/build_qp_variant.sh res_stamps) or UMPC_QP_RUIZ_STAMPS=1
(... ruiz_stamps); run with UMPC_LIB=robobee3d_amd/variants/libumpc_<name>.so: six internal intervals (100 MHz ticks) then
replace the info rows. usage: python tools/p5f_res_stamps.py [B] [res|ruiz|loop] (loop: UMPC_QP_LOOP_STAMPS=1 ... loop_stamps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robobee3d_amd.batchqp import PlanarP5fMPC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mpc = PlanarP5fMPC(B, torch.float32)
mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
for ti in range(2, 8):
    mpc.tick(0.002 * ti)
torch.cuda.synchronize()
t = mpc.qp.info.cpu().numpy().astype(np.float64)
which = sys.argv[2] if len(sys.argv) > 2 else "res"
names = ("A (own entries) stream -> LDS", "KKT fill + LDL'", "barrier (the other wavefronts' factorisations)", "warm start x, y, z -> LDS",
         "y0 homes + preloads", "49 + 1 iterations") if which == "loop" else ("prologue: P, q -> AGPRs, A -> LDS (430 loads)", "norm phases, 10 passes", "apply phases, 10 passes", "P, q, c -> LDS",
         "residual stream stores + drain", "glue block") if which == "ruiz" else ("A loads (256) + wait", "pass 1: A x (256 fmac)", "rows (164: norms, 656 stores)", "y -> accumulators (164 moves)",
         "columns (87: A'y, P x, q, 174 stores)", "termination test + drain")
print(mpc.qp.kernel_name, "B = %d, %s, us per part (mean over robots | max):" % (B, "Ruiz + glue blocks" if which == "ruiz" else "loop block, wavefront 0" if which == "loop" else "residual block"))
for n, r in zip(names, t):
    print("  %-48s %7.1f | %7.1f" % (n, r.mean() / 100.0, r.max() / 100.0))
print("  total %.1f" % (t[:6].sum(0).mean() / 100.0))
