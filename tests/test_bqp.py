"""General-structure batch QP solver: csrc/umpc_bqp.hip walks symbolic tables
built by robobee3d_amd/qpstruct.py. The oracle is oracle/osqp_table.py (numpy, vectorised over robots only);
it is PINNED by reproducing, bit for bit in fp32 with the reference's KKT permutation, the C restatement
(oracle/umpc_oracle.c, itself bit-identical to the compiled reference on tests/golden/seq_iter*.npz) on the
uprightmpc2 N = 3 problem."""
import os
import sys

import numpy as np
import pytest

from conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def raw_uprightmpc2_qp(g, idx, dtype):
    """Raw (Pv, Av, q, l, u) [rows][B] of the uprightmpc2 N = 3 QP from a seq_iter fixture: constants of A as
    assembled (+-1, template_controllers.py:28-63), the 48 state-dependent entries from the fixture's Ax."""
    from robobee3d_amd import symbolic
    A_p, A_i, A_tag = symbolic.build_A(3)
    B = len(idx)
    Av = np.zeros((len(A_i), B), dtype)
    for p, tag in enumerate(A_tag):
        if tag[0] == 'c':
            Av[p] = tag[1]
    axidx = symbolic.ax_idx(3)
    Av[axidx] = g["Ax"][idx].T
    return A_p, A_i, g["Px"][idx].T.astype(dtype), Av, g["q"][idx].T.astype(dtype), g["l"][idx].T.astype(dtype), \
        g["u"][idx].T.astype(dtype)


def test_qpstruct_matches_specialised_symbolic(structure):
    """qpstruct.analyse_qp on the uprightmpc2 pattern = symbolic.analyse = the reference's generated tables."""
    from robobee3d_amd import qpstruct, symbolic
    A_p, A_i, _ = symbolic.build_A(3)
    s = qpstruct.analyse_qp(45, 39, A_p, A_i, list(range(45)), perm=structure["perm"])
    assert s.K_p == [int(v) for v in structure["K_p"]] and s.K_i == [int(v) for v in structure["K_i"]]
    assert s.L_p == [int(v) for v in structure["L_p"]] and s.L_i == [int(v) for v in structure["L_i"]]
    assert s.etree == [int(v) for v in structure["etree"]]
    own = qpstruct.analyse_qp(45, 39, A_p, A_i, list(range(45)))
    assert own.nnzL == 213 and own.perm == symbolic.analyse(3).perm
    # CSR views are permutations of the CSC ones
    assert sorted(own.tables["Ar_k"]) == list(range(own.nnzA)) and sorted(own.tables["Lr_k"]) == list(range(own.nnzL))
    assert int(own.blob[0]) == 45 and int(own.blob[6]) == own.nrows


@pytest.mark.parametrize("fixture", ["seq_iter2", "seq_iter50"])
def test_table_oracle_is_bitwise_the_c_oracle(oracle_built, structure, fixture):
    import osqp_table
    g = golden(fixture + ".npz")
    n = min(len(g["p0"]), 48)
    idx = np.arange(n)
    iters = int(g["maxIter"])
    perm = structure["perm"]
    A_p, A_i, Pv, Av, q, l, u = raw_uprightmpc2_qp(g, idx, np.float32)
    Eprev = np.ones((39, n), np.float32)
    Eprev[36:] = g["pre_E3"][idx].T
    r = osqp_table.solve(45, 39, A_p, A_i, list(range(45)), perm, Pv, Av, q, l, u, g["pre_x"][idx].T,
                         g["pre_y"][idx].T, g["pre_z"][idx].T, Eprev, osqp_table.Settings(max_iter=iters),
                         dtype=np.float32)
    for k in idx:
        o = oracle_built.Oracle(np.float32, perm=perm, maxIter=iters)
        o.set_canonical(True, g["pre_E3"][k])
        o.set_iterates(g["pre_x"][k], g["pre_y"][k], g["pre_z"][k])
        o.set_T0(g["pre_T0"][k])
        o.update(g["p0"][k], g["R0"][k], g["dq0"][k], g["pdes"][k], g["dpdes"][k], g["sdes"][k], g["actualT0"][k])
        for name, mine in (("D", r["D"]), ("E", r["E"]), ("L_x", r["L"]), ("Ddinv", r["Dinv"]), ("x", r["x"]),
                           ("y", r["y"]), ("z", r["z"]), ("sol_x", r["sol_x"])):
            assert np.array_equal(o.get(name), mine[:, k], equal_nan=True), (fixture, k, name)
        assert o.get("c")[0] == r["c"][k] and o.get("pri_res")[0] == r["pri_res"][k]
        assert o.get("dua_res")[0] == r["dua_res"][k]
        assert int(o.get("status_val")[0]) == int(r["status"][k])
