"""How far a call of the REFERENCE sits from the decision boundary of its own status word.

check_termination (template/uprightmpc2/auxil.c:684-789) decides OSQP_SOLVED (1) when
    pri_res < eps_prim  and  dua_res < eps_dual,
with eps_prim = eps_abs + eps_rel max(|z|, |A x|) (compute_pri_tol, auxil.c:262-288) and
eps_dual = eps_abs + eps_rel max(|q|, |A'y|, |P x|) (compute_dua_tol, auxil.c:309-349), all in UNSCALED variables
(scaled_termination = 0), eps_abs = eps_rel = 1e-4; osqp.c:555-573 repeats the test with every eps x 10 for
OSQP_SOLVED_INACCURATE (2) and leaves OSQP_MAX_ITER_REACHED (-2) otherwise.

From one fixture record (the reference's own sol_x, sol_y, z, E, pri_res, dua_res and its raw P, q, A) this module
restates those two thresholds in float64 and returns, for a pair (reference status, other status) that differ, the
factor by which the residual that decides between them is away from its threshold: max(r / eps, eps / r) of the
comparison that must have come out the other way. A status flip is legitimate only when that factor is small.
"""
import numpy as np


def raw_A(structure, Ax):
    """dense raw constraint matrix of one call: the constant entries of the generated pattern (symbolic.A_tag) and the
    48 per-call entries at Ax_idx (uprightmpc2.c:65-113)"""
    from robobee3d_amd import symbolic
    s = symbolic.analyse(3, perm=structure["perm"])
    vals = np.zeros(len(s.A_i))
    for p, tag in enumerate(s.A_tag):
        if tag[0] == "c":
            vals[p] = tag[1]
    vals[np.asarray(structure["Ax_idx"])] = Ax
    A = np.zeros((s.nc, s.nx))
    for j in range(s.nx):
        for p in range(s.A_p[j], s.A_p[j + 1]):
            A[s.A_i[p], j] = vals[p]
    return A


def thresholds(structure, seq, k):
    """(eps_prim, eps_dual) of call k at the strict tolerances, from the reference's own final iterate"""
    f = lambda a: np.asarray(a, np.float64)
    A = raw_A(structure, f(seq["Ax"][k]))
    x, y = f(seq["sol_x"][k]), f(seq["sol_y"][k])
    z = f(seq["z"][k]) / f(seq["E"][k])
    ninf = lambda v: float(np.abs(v).max())
    eps_prim = 1e-4 + 1e-4 * max(ninf(z), ninf(A @ x))
    eps_dual = 1e-4 + 1e-4 * max(ninf(f(seq["q"][k])), ninf(A.T @ y), ninf(f(seq["Px"][k]) * x))
    return eps_prim, eps_dual


def flip_distance(structure, seq, k, other_status):
    """factor (>= 1) by which the reference's deciding residual is away from its threshold, for a robot whose status
    in some other evaluation of the same algorithm is `other_status` != seq["status"][k]. inf when no single
    comparison of check_termination can explain the pair."""
    ref = int(seq["status"][k])
    pri, dua = float(seq["pri_res"][k]), float(seq["dua_res"][k])
    ep, ed = thresholds(structure, seq, k)
    fac = lambda r, e: max(r / e, e / r) if r > 0 and e > 0 else np.inf
    # the eps x 10 test scales eps_abs and eps_rel alike, so its thresholds are 10 x the strict ones
    strict = (fac(pri, ep), fac(dua, ed))
    approx = (fac(pri, 10 * ep), fac(dua, 10 * ed))
    pair = {ref, int(other_status)}

    def deciding(facs, pri_ok, dua_ok):
        # the comparison(s) that would have to flip: a failing test flips when a failed comparison passes (all failed
        # ones must), a passing test flips when any one comparison fails
        if pri_ok and dua_ok:
            return min(facs)
        return max(f_ for f_, ok in zip(facs, (pri_ok, dua_ok)) if not ok)
    if pair == {1, 2}:
        return deciding(strict, pri < ep, dua < ed)
    if pair == {2, -2}:
        return deciding(approx, pri < 10 * ep, dua < 10 * ed)
    if pair == {1, -2}:      # both tests come out differently: both must be at their boundary
        return max(deciding(strict, pri < ep, dua < ed), deciding(approx, pri < 10 * ep, dua < 10 * ed))
    return np.inf


def worst_flip(structure, seq, status):
    """(number of flips, worst flip_distance, index of the worst) over a batch of status words"""
    status = np.asarray(status)
    idx = np.nonzero(status != seq["status"][:len(status)])[0]
    worst, arg = 1.0, -1
    for k in idx:
        d = flip_distance(structure, seq, int(k), int(status[k]))
        if d > worst:
            worst, arg = d, int(k)
    return len(idx), worst, arg
