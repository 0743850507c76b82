"""Build-time symbolic analysis of the uprightmpc2 QP (pure Python, no numpy
needed at import). Everything the HIP kernels need to know about sparsity is
decided HERE, once, and baked into straight-line code by codegen.py -- the
device never loads an index.

What is restated (reference file:line):
  * the constraint-matrix pattern A (39x45, 111 nnz at N=3) and the positions of
    the 48 state-dependent entries: template/template_controllers.py:28-63
    (initConstraint), template/uprightmpc2/uprightmpc2.c:65-113 (Ax_idx);
  * the quasi-definite KKT matrix [[P+sigma I, A'],[A, -diag(1/rho)]] in upper
    triangular CSC form and its symmetric permutation:
    template/uprightmpc2/kkt.c:6-177, workspace.c:2004-2165;
  * the elimination tree / column counts and the up-looking LDL' schedule:
    template/uprightmpc2/qdldl.c:34-247.

The fill-reducing ordering is OUR OWN (greedy minimum fill, deterministic); it
reaches the same fill as the reference's AMD ordering (nnz(L) = 213 at N = 3).
`analyse(N, perm=...)` accepts an explicit permutation so the CPU tests can
check this module against the reference's tables (tests/golden/structure.npz).
"""
from collections import namedtuple

NY, NU = 6, 3


def dims(N):
    nx = N * (2 * NY + NU)
    nc = 2 * N * NY + N
    return nx, nc


def build_A(N):
    """Returns (A_p, A_i, A_tag) in CSC order. A_tag[p] describes the value of
    entry p as assembled by umpcUpdateConstraint (uprightmpc2.c:161-179):
      ('c', v)      constant v (+1 / -1)
      ('dt',)       dt
      ('T0dt',)     dt*T0
      ('s0', i)     dt*s0[i]
      ('Btau', i)   dt*Btau[i]   (i = 0..5, column-major 3x2)
    """
    nx, nc = dims(N)
    dense = {}
    n1, n2, nc1, nc2 = N * NY, 2 * N * NY, N * NY, 2 * N * NY
    for k in range(N):
        for i in range(NY):
            dense[(k * NY + i, k * NY + i)] = ('c', -1.0)
            dense[(k * NY + i, n1 + k * NY + i)] = ('dt',)
            if k > 0:
                dense[(k * NY + i, (k - 1) * NY + i)] = ('c', 1.0)
            dense[(nc1 + k * NY + i, n1 + k * NY + i)] = ('c', -1.0)
            if k > 0:
                dense[(nc1 + k * NY + i, n1 + (k - 1) * NY + i)] = ('c', 1.0)
        for i in range(3):
            dense[(nc1 + k * NY + i, n2 + k * NU)] = ('s0', i)
            dense[(nc1 + k * NY + 3 + i, n2 + k * NU + 1)] = ('Btau', i)
            dense[(nc1 + k * NY + 3 + i, n2 + k * NU + 2)] = ('Btau', 3 + i)
            if k > 1:
                dense[(nc1 + k * NY + i, (k - 2) * NY + 3 + i)] = ('T0dt',)
        dense[(nc2 + k, n2 + 3 * k)] = ('c', 1.0)
    A_p, A_i, A_tag = [0], [], []
    for j in range(nx):
        for i in range(nc):
            if (i, j) in dense:
                A_i.append(i)
                A_tag.append(dense[(i, j)])
        A_p.append(len(A_i))
    return A_p, A_i, A_tag


def ax_idx(N):
    """Positions (into A's CSC value array) of the state-dependent entries, in the
    reference's order [T0dt | dt | s0 | Btau]: uprightmpc2.c:65-113."""
    idx = []
    n2 = 2 * NY + 3
    for k in range(N - 2):
        idx += [n2 * k + 8, n2 * k + 11, n2 * k + 14]
    n1 = (2 * N - 1) * NY + (N - 2) * 3
    n2 = 3 * NY
    for k in range(N):
        idx += [n1 + n2 * k + (3 * i if k < N - 1 else 2 * i) for i in range(6)]
    n1 += 3 * NY * (N - 1) + 2 * NY
    n2 = 10
    for k in range(N):
        idx += [n1 + n2 * k + i for i in range(3)]
    for k in range(N):
        idx += [n1 + n2 * k + 4 + i for i in range(6)]
    return idx


def kkt_graph(N, A_p, A_i):
    nx, nc = dims(N)
    adj = [set() for _ in range(nx + nc)]
    for j in range(nx):
        for p in range(A_p[j], A_p[j + 1]):
            r = nx + A_i[p]
            adj[j].add(r)
            adj[r].add(j)
    return adj


def min_fill_ordering(adj0, last=(), hold=None):
    """Greedy minimum-fill elimination ordering, ties -> (held back, degree, index). last: vertices that are eliminated only
    after every other one (a separator held back: qpstruct.bisect_ordering). hold: {vertex: set of vertices} -- among
    candidates of EQUAL fill, a vertex is passed over while none of hold[vertex] has been eliminated (a numerical hint, no
    effect on the fill: a variable without a cost term eliminated before any constraint row that carries it with a unit
    coefficient gets the pivot sigma -- 1e-6 -- and multipliers of 5e5; batchqp.p5f_analysis)."""
    n = len(adj0)
    adj = [set(s) for s in adj0]
    alive = set(range(n))
    last = set(last)
    hold = hold or {}
    perm = []
    while alive:
        best, bkey = None, None
        for v in sorted(alive - last or alive):
            nb = [u for u in adj[v] if u in alive]
            fill = 0
            for a_i, a in enumerate(nb):
                for b in nb[a_i + 1:]:
                    if b not in adj[a]:
                        fill += 1
            held = 1 if (hold.get(v) and all(u in alive for u in hold[v])) else 0
            key = (fill, held, len(nb), v)
            if bkey is None or key < bkey:
                best, bkey = v, key
        nb = [u for u in adj[best] if u in alive]
        for a in nb:
            adj[a].update(nb)
            adj[a].discard(a)
        alive.discard(best)
        perm.append(best)
    return perm


Structure = namedtuple("Structure", "N nx nc nk A_p A_i A_tag Ax_idx perm pinv K_p K_i K_src "
                                    "PtoKKT AtoKKT rhotoKKT etree Lnz L_p L_i factor_ops")


def analyse(N=3, perm=None):
    nx, nc = dims(N)
    nk = nx + nc
    A_p, A_i, A_tag = build_A(N)
    if perm is None:
        perm = min_fill_ordering(kkt_graph(N, A_p, A_i))
    perm = [int(v) for v in perm]
    pinv = [0] * nk
    for k, v in enumerate(perm):
        pinv[v] = k
    # upper-triangular KKT in form_KKT's triplet order, bucketed by column (stable)
    trip = []  # (row, col, src)
    for j in range(nx):
        trip.append((j, j, ('P', j)))
    for j in range(nx):
        for p in range(A_p[j], A_p[j + 1]):
            trip.append((j, nx + A_i[p], ('A', p)))
    for i in range(nc):
        trip.append((nx + i, nx + i, ('R', i)))
    cols = [[] for _ in range(nk)]
    for (i, j, src) in trip:
        cols[j].append((i, src))
    # symmetric permutation, upper part (cs_symperm traversal order)
    pcols = [[] for _ in range(nk)]
    for j in range(nk):
        j2 = pinv[j]
        for (i, src) in cols[j]:
            i2 = pinv[i]
            pcols[max(i2, j2)].append((min(i2, j2), src))
    K_p, K_i, K_src = [0], [], []
    PtoKKT, AtoKKT, rhotoKKT = [0] * nx, [0] * len(A_i), [0] * nc
    for j in range(nk):
        for (i, src) in pcols[j]:
            {'P': PtoKKT, 'A': AtoKKT, 'R': rhotoKKT}[src[0]][src[1]] = len(K_i)
            K_i.append(i)
            K_src.append(src)
        K_p.append(len(K_i))
    # elimination tree + column counts (QDLDL_etree)
    work = [0] * nk
    Lnz = [0] * nk
    etree = [-1] * nk
    for j in range(nk):
        work[j] = j
        for p in range(K_p[j], K_p[j + 1]):
            i = K_i[p]
            while work[i] != j:
                if etree[i] == -1:
                    etree[i] = j
                Lnz[i] += 1
                work[i] = j
                i = etree[i]
    L_p = [0]
    for i in range(nk):
        L_p.append(L_p[-1] + Lnz[i])
    # symbolic run of the up-looking factorisation (QDLDL_factor, qdldl.c:86-247):
    # records, for every row k, the elimination sequence and where each L entry lands.
    L_i = [-1] * L_p[-1]
    Lnext = list(L_p[:-1])
    ops = []  # per k: dict(diag=src_index, init=[(row, kkt_index)], elim=[(cidx, [(Lidx,row)...], newLidx)])
    ops.append(dict(k=0, diag=K_p[0], init=[], elim=[]))
    assert K_p[1] - K_p[0] == 1 and K_i[0] == 0
    for k in range(1, nk):
        marked = [False] * nk
        yIdx = []
        init = []
        diag = None
        for p in range(K_p[k], K_p[k + 1]):
            b = K_i[p]
            if b == k:
                diag = p
                continue
            init.append((b, p))
            nxt = b
            if not marked[nxt]:
                marked[nxt] = True
                buf = [nxt]
                nxt = etree[b]
                while nxt != -1 and nxt < k:
                    if marked[nxt]:
                        break
                    marked[nxt] = True
                    buf.append(nxt)
                    nxt = etree[nxt]
                while buf:
                    yIdx.append(buf.pop())
        elim = []
        for c in reversed(yIdx):
            upd = [(j, L_i[j]) for j in range(L_p[c], Lnext[c])]
            new = Lnext[c]
            L_i[new] = k
            Lnext[c] += 1
            elim.append((c, upd, new))
        assert diag is not None
        ops.append(dict(k=k, diag=diag, init=init, elim=elim))
    assert all(v >= 0 for v in L_i)
    return Structure(N, nx, nc, nk, A_p, A_i, A_tag, ax_idx(N), perm, pinv, K_p, K_i, K_src,
                     PtoKKT, AtoKKT, rhotoKKT, etree, Lnz, L_p, L_i, ops)


if __name__ == "__main__":
    s = analyse(3)
    print("nnzA", len(s.A_i), "nnzKKT", len(s.K_i), "nnzL", len(s.L_i), "perm", s.perm)
