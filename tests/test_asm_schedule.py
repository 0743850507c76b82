"""CPU check of the generated gfx950 ADMM assembly (robobee3d_amd/asmgen.py): the emitted
instruction list is interpreted on numpy float32 (one lane) and compared with the oracle's
iterates after the same number of iterations from the same factorisation. Catches register
reuse / fetch-distance / loop-control mistakes before the code ever reaches a GPU."""
import numpy as np
import pytest

from conftest import golden


@pytest.fixture(scope="module")
def prog():
    from robobee3d_amd import asmgen
    ins, s = asmgen.program()
    return asmgen, ins, s


def _case(oracle_built, asmgen, ins, s, seq, k, iters):
    perm = np.array(s.perm, np.int32)
    args = (seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k], seq["sdes"][k],
            float(seq["actualT0"][k]))

    def fresh(mi):
        o = oracle_built.Oracle(np.float32, perm=perm, maxIter=mi)
        o.set_canonical(True, seq["pre_E3"][k])
        o.set_iterates(seq["pre_x"][k], seq["pre_y"][k], seq["pre_z"][k])
        o.set_T0(float(seq["pre_T0"][k]))
        o.update(*args)
        return o

    o = fresh(0)  # factorisation only
    ws = np.zeros(asmgen.WS_ROWS, np.float32)
    ctrl = np.zeros(127, np.float32)
    # L entry j lives at storage position pos[j] (asmgen.solve_plan): LDS word if < NLDS, else workspace row
    pos = np.array(asmgen.l_positions())
    Lstore = np.zeros(213, np.float32)
    Lstore[pos] = -o.get("L_x")          # the loop keeps the factor negated (v_fmac_f32)
    ws[asmgen.FAC_L:asmgen.FAC_L + 213] = Lstore
    ws[asmgen.FAC_DI:asmgen.FAC_DI + 84] = o.get("Ddinv")
    ws[asmgen.FAC_Q:asmgen.FAC_Q + 45] = o.get("q")
    l, u = o.get("l"), o.get("u")
    ws[asmgen.FAC_LOEQ:asmgen.FAC_LOEQ + 36] = l[:36]
    ws[asmgen.FAC_M:asmgen.FAC_M + 3] = l[36:]
    ws[asmgen.FAC_M + 3:asmgen.FAC_M + 6] = u[36:]
    ws[asmgen.FAC_M + 6:asmgen.FAC_M + 9] = o.get("rho_vec")[36:]
    ws[asmgen.FAC_M + 9:asmgen.FAC_M + 12] = o.get("rho_inv_vec")[36:]
    ctrl[:45], ctrl[45:84], ctrl[84:123] = seq["pre_x"][k], seq["pre_y"][k], seq["pre_z"][k]
    x_before_last = None
    if iters >= 1:
        ob = fresh(iters - 1)
        x_before_last = ob.get("x") if iters > 1 else seq["pre_x"][k]
    lds = np.zeros(160, np.float32)
    lds[:asmgen.NLDS] = Lstore[:asmgen.NLDS]   # phase A leaves storage positions 0..159 in LDS
    ws[asmgen.FAC_L:asmgen.FAC_L + asmgen.NLDS] = np.nan  # ... and the program must not read those rows
    asmgen.simulate(ins, ws, ctrl, iters, lds)
    o2 = fresh(iters)
    for name, sl in (("x", slice(0, 45)), ("y", slice(45, 84)), ("z", slice(84, 123))):
        ref = o2.get(name)
        assert np.abs(lds[sl] - ref).max() <= 2e-5 * max(1e-6, np.abs(ref).max()), (name, k, iters)
    if iters >= 1:  # captured x_prev of the last iteration
        xp = ws[asmgen.WS_XPREV:asmgen.WS_XPREV + 45]
        assert np.abs(xp - x_before_last).max() <= 2e-5 * np.abs(x_before_last).max()
        dy = ws[asmgen.WS_DY:asmgen.WS_DY + 39]
        assert np.all(np.isfinite(dy)) and np.abs(dy).max() > 0


@pytest.mark.parametrize("iters", [0, 1, 2, 3, 7, 50])
def test_generated_admm_program_matches_oracle(oracle_built, prog, iters):
    asmgen, ins, s = prog
    seq = golden("seq_iter50.npz")
    for k in (0, 3, 11):
        _case(oracle_built, asmgen, ins, s, seq, k, iters)


def test_register_map_is_disjoint():
    from robobee3d_amd import asmgen as g
    regs = [(g.V_W, 45), (g.V_WZ, 39), (g.V_X, 45), (g.V_Y, 39), (g.V_Z, 39), (g.V_RING, 16),
            (g.V_AT, 2 * g.N_ATP), (g.V_TT, 8)]
    used = set()
    for lo, n in regs:
        r = set(range(lo, lo + n))
        assert lo % 2 == 0, "packed operands need even-aligned bases"
        assert not (used & r) and min(r) >= 2 and max(r) < g.V_END
        used |= r
    assert g.A_M + 12 <= 256 and g.NLDS * 4 * 64 <= 40960
    # every register the program writes is inside the declared clobber range
    ins, _ = g.program()
    import re
    for t in ins:
        if t[0].startswith(("v_", "ds_read", "global_load")) and isinstance(t[1], str) and t[1].startswith("v"):
            hi = max(int(n) for n in re.findall(r"\d+", t[1]))
            assert 2 <= hi < g.V_END, t
