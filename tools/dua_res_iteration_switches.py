#!/usr/bin/env python3
"""Round 5, CPU only, part 2: tools/dua_res_decomposition.py shows that the 2.9x dual-residual ratio of the
product's fp32 stream against the reference fixture is a property of the ITERATE it ends on after 50 iterations, not of how
phase C evaluates residuals. This script switches the arithmetic deviations of the stream's ITERATION (DESIGN.md section 3,
items 2 and 3) on ONE AT A TIME in a float32 numpy statement of the reference's loop (oracle/osqp_table.py's, bit-identical
to the reference's C on these fixtures when every switch is off) and records, for the 256 calls of
tests/golden/seq_iter50.npz: the TRUE (float64) dual residual of the final iterate relative to the fixture's dua_res, and
the status words that then differ from the reference's.

    fma_rhs    rhs = fma(sigma, x, -q), fma(-1/rho, y, z)               (v_pk_fma_f32; the reference rounds the product)
    fma_solve  W[d] = fma(-L, W[s], W[d]) in both triangular solves    (v_fmac_f32 / v_pk_fma_f32)
    fma_x      x = fma(alpha, x~, (1 - alpha) x)
    dy_short   dynamics rows (z == l == u): delta_y = alpha (nu - y) instead of rho (alpha z~ + (1 - alpha) z - z) with
               z~ = (z - y / rho) + nu / rho                              
    all        the four together (the stream's iteration up to the order of additions inside one unknown)

usage: python tools/dua_res_iteration_switches.py >> profiles/r05_dua_res_decomposition.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import osqp_table                                   # noqa: E402
from test_bqp import raw_uprightmpc2_qp             # noqa: E402

f32 = np.float32
seq = np.load(os.path.join(ROOT, "tests", "golden", "seq_iter50.npz"))
structure = np.load(os.path.join(ROOT, "tests", "golden", "structure.npz"))
from robobee3d_amd import symbolic                  # noqa: E402
perm = symbolic.analyse(3).perm                     # the product's own elimination order
B = len(seq["p0"])
idx = np.arange(B)
n, m, nk, neq = 45, 39, 84, 36
A_p, A_i, Pv, Av, q, l, u = raw_uprightmpc2_qp(seq, idx, f32)
Eprev = np.ones((m, B), f32)
Eprev[36:] = seq["pre_E3"][idx].T
x0, y0, z0 = seq["pre_x"][idx].T.astype(f32), seq["pre_y"][idx].T.astype(f32), seq["pre_z"][idx].T.astype(f32)
base = osqp_table.solve(n, m, A_p, A_i, list(range(n)), perm, Pv, Av, q, l, u, x0, y0, z0, Eprev, osqp_table.Settings(max_iter=0), dtype=f32)
full = osqp_table.solve(n, m, A_p, A_i, list(range(n)), perm, Pv, Av, q, l, u, x0, y0, z0, Eprev, osqp_table.Settings(max_iter=50), dtype=f32)
Lx, Ddinv, L_i, L_p, qs, E, D, c, rho = (base[k] for k in ("L", "Dinv", "L_i", "L_p", "qs", "E", "D", "c", "rho"))
rinv = np.where(rho == f32(1e-6), f32(1. / 1e-6), np.where(rho == f32(0.1), f32(1. / 0.1), f32(1. / float(f32(1e3 * 0.1))))).astype(f32)
lsc, usc = (l * E).astype(f32), (u * E).astype(f32)
sigma, alpha = f32(1e-6), f32(1.6)
oma = f32(1.0) - alpha


def fma(a, b, c_):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c_, np.float64)).astype(f32)


def run(sw, iters=50):
    x, y, z = x0.copy(), y0.copy(), z0.copy()
    for it in range(iters):
        xp, zp = x, z
        if "fma_rhs" in sw:
            rhs = np.concatenate((fma(sigma, xp, -qs), fma(-rinv, y, zp)), 0)
        else:
            rhs = np.concatenate((sigma * xp - qs, zp - rinv * y), 0).astype(f32)
        bp = rhs[perm].copy()
        for i in range(nk):
            for j in range(L_p[i], L_p[i + 1]):
                bp[L_i[j]] = fma(-Lx[j], bp[i], bp[L_i[j]]) if "fma_solve" in sw else bp[L_i[j]] - Lx[j] * bp[i]
        bp = bp * Ddinv
        for i in range(nk - 1, -1, -1):
            for j in range(L_p[i], L_p[i + 1]):
                bp[i] = fma(-Lx[j], bp[L_i[j]], bp[i]) if "fma_solve" in sw else bp[i] - Lx[j] * bp[L_i[j]]
        sol = np.empty_like(bp)
        sol[perm] = bp
        xt, nu = sol[:n], sol[n:]
        x = fma(alpha, xt, oma * xp) if "fma_x" in sw else alpha * xt + oma * xp
        zt = rhs[n:] + rinv * nu
        zz = alpha * zt + oma * zp + rinv * y
        znew = np.minimum(np.maximum(zz, lsc), usc)
        dy = rho * (alpha * zt + oma * zp - znew)
        if "dy_short" in sw and it > 0:
            # dynamics rows after the first iteration: z stays l, delta_y = alpha (nu - y)
            dy = dy.copy()
            dy[:neq] = alpha * (nu[:neq] - y[:neq])
        y = (y + dy).astype(f32)
        z = znew.astype(f32)
    return x.astype(f32), y, z


def true_dua(x, y):
    f = lambda a: np.asarray(a, np.float64)
    xu, yu = f(D) * f(x), f(E) * f(y) / f(c)
    Aty = np.zeros((n, B))
    for j in range(n):
        for p in range(A_p[j], A_p[j + 1]):
            Aty[j] += f(Av[p]) * yu[A_i[p]]
    r = f(q) + f(Pv) * xu + Aty
    return np.abs(r).max(0), (np.abs(f(q)).max(0), np.abs(Aty).max(0), np.abs(f(Pv) * xu).max(0))


def true_pri(x, z):
    f = lambda a: np.asarray(a, np.float64)
    xu, zu = f(D) * f(x), f(z) / f(E)
    Ax = np.zeros((m, B))
    for j in range(n):
        for p in range(A_p[j], A_p[j + 1]):
            Ax[A_i[p]] += f(Av[p]) * xu[j]
    return np.abs(Ax - zu).max(0), (np.abs(Ax).max(0), np.abs(zu).max(0))


def status(x, y, z):
    dua, (nq, nAty, nPx) = true_dua(x, y)
    pri, (nAx, nz) = true_pri(x, z)
    ep = 1e-4 + 1e-4 * np.maximum(nAx, nz)
    ed = 1e-4 + 1e-4 * np.maximum(np.maximum(nq, nAty), nPx)
    return np.where((pri < ep) & (dua < ed), 1, np.where((pri < 10 * ep) & (dua < 10 * ed), 2, -2)), dua


xb, yb, zb = run(())
assert np.array_equal(xb, full["x"]) and np.array_equal(yb, full["y"]) and np.array_equal(zb, full["z"]), \
    "the switch-free loop must be the table oracle's (bit-identical to the reference's C)"
ref_dua, ref_status = seq["dua_res"][idx].astype(np.float64), seq["status"][idx]
print("\n" + __doc__.split("usage:")[0])
print("%-12s  %-44s  %-34s  %s" % ("switch", "median true dua_res / fixture dua_res  [r, max(r, 1/r)]", "median true dua_res / switch-free",
                                   "status (from true residuals) != reference"))
sb, db = status(xb, yb, zb)
for name, sw in (("none", ()), ("fma_rhs", ("fma_rhs",)), ("fma_solve", ("fma_solve",)), ("fma_x", ("fma_x",)), ("dy_short", ("dy_short",)),
                 ("all", ("fma_rhs", "fma_solve", "fma_x", "dy_short"))):
    x, y, z = (xb, yb, zb) if not sw else run(sw)
    st, dua = status(x, y, z)
    r = dua / ref_dua
    print("%-12s  %-8.3f %-35.3f  %-34.3f  %d of %d" % (name, float(np.median(r)), float(np.median(np.maximum(r, 1 / r))),
                                                       float(np.median(dua / db)), int((st != ref_status).sum()), B))
