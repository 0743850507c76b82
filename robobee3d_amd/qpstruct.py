"""Build-time symbolic analysis of an ARBITRARY sparse QP structure for the table-driven batch solver
(csrc/umpc_bqp.hip; SURVEY 8 rows a21 / a22 / f-4: planar p5f MPC, the v1 template QP, general-N
uprightmpc2). Same mathematics as symbolic.py (which bakes the N = 3 structure into straight-line code);
here the result is a flat int32 table blob that the generic kernel walks with scalar loads.

  min 1/2 x'Px + q'x  s.t.  l <= Ax <= u,   P diagonal on a subset of the columns, A in CSC.

What is restated (reference file:line):
  * KKT [[P + sigma I, A'], [A, -diag(1/rho)]] in upper-triangular CSC form and its symmetric
    permutation: template/uprightmpc2/kkt.c:6-177 (form_KKT), workspace.c:2004-2165;
  * elimination tree / column counts / up-looking LDL' schedule: template/uprightmpc2/qdldl.c:34-247.
The fill-reducing ordering is symbolic.min_fill_ordering (our own); an explicit `perm` can be passed so
the tests can replay the reference's tables.
"""
from collections import namedtuple

import numpy as np

from . import symbolic

QPStructure = namedtuple(
    "QPStructure", "n m nk nnzP nnzA nnzL P_cols A_p A_i perm pinv K_p K_i K_src etree L_p L_i factor_ops "
                   "tables blob rows nrows")

# workspace rows, in order; sizes in terms of (n, m, nk, nnzP, nnzA, nnzL)
_ROWS = [("PS", "nnzP"), ("AS", "nnzA"), ("QS", "n"), ("LS", "m"), ("US", "m"), ("D", "n"), ("E", "m"),
         ("DT", "n"), ("ET", "m"), ("RHO", "m"), ("RINV", "m"), ("KD", "nk"), ("LX", "nnzL"), ("DI", "nk"),
         ("YV", "nk"), ("WV", "nk"), ("XP", "n"), ("DY", "m"), ("T1", "n"), ("T2", "n"), ("T3", "m"), ("SC", 4)]
_TABLES = ["pinv", "pidx", "A_p", "A_i", "Ar_p", "Ar_j", "Ar_k", "fi_p", "fi_b", "fi_src", "fe_p", "fe_c",
           "fe_new", "L_p", "L_i", "Lr_p", "Lr_j", "Lr_k",
           # wave-per-robot kernel (level schedules, csrc/umpc_bqp.hip bqp_wave_kernel)
           "perm", "A_j", "lev_p", "lev_nodes", "elev_p", "elev_ent", "l_ksrc", "l_col", "ft_p", "ft_a", "ft_b", "ft_j",
           "fd_p", "fd_a", "fd_j"]
HEADER_WORDS = 64


def kkt_adjacency(n, m, A_p, A_i):
    adj = [set() for _ in range(n + m)]
    for j in range(n):
        for q in range(A_p[j], A_p[j + 1]):
            adj[j].add(n + A_i[q])
            adj[n + A_i[q]].add(j)
    return adj


def analyse_pattern(nx, nc, A_p, A_i, perm=None):
    """The structure-only part shared with symbolic.analyse: permuted upper-triangular KKT, etree, L pattern
    and the recorded up-looking factorisation schedule. Every column has a diagonal entry ('P', j) (= P_jj +
    sigma, or sigma alone) / ('R', i)."""
    nk = nx + nc
    if perm is None:
        adj = [set() for _ in range(nk)]
        for j in range(nx):
            for p in range(A_p[j], A_p[j + 1]):
                r = nx + A_i[p]
                adj[j].add(r)
                adj[r].add(j)
        perm = symbolic.min_fill_ordering(adj)
    perm = [int(v) for v in perm]
    pinv = [0] * nk
    for k, v in enumerate(perm):
        pinv[v] = k
    trip = [(j, j, ('P', j)) for j in range(nx)]
    for j in range(nx):
        for p in range(A_p[j], A_p[j + 1]):
            trip.append((j, nx + A_i[p], ('A', p)))
    for i in range(nc):
        trip.append((nx + i, nx + i, ('R', i)))
    cols = [[] for _ in range(nk)]
    for (i, j, src) in trip:
        cols[j].append((i, src))
    pcols = [[] for _ in range(nk)]
    for j in range(nk):
        j2 = pinv[j]
        for (i, src) in cols[j]:
            i2 = pinv[i]
            pcols[max(i2, j2)].append((min(i2, j2), src))
    K_p, K_i, K_src = [0], [], []
    for j in range(nk):
        for (i, src) in pcols[j]:
            K_i.append(i)
            K_src.append(src)
        K_p.append(len(K_i))
    work, Lnz, etree = [0] * nk, [0] * nk, [-1] * nk
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
    L_i = [-1] * L_p[-1]
    Lnext = list(L_p[:-1])
    ops = []
    for k in range(nk):
        marked = [False] * nk
        yIdx, init, diag = [], [], None
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
    return perm, pinv, K_p, K_i, K_src, etree, L_p, L_i, ops


def analyse_qp(n, m, A_p, A_i, P_cols, perm=None, parts=None):
    """n, m: sizes; (A_p, A_i): CSC pattern of A (row indices ascending within a column); P_cols: sorted list of
    the columns j that carry a diagonal P entry (value index k = position in this list).
    parts: bisect_parts' labels that `perm` (bisect_ordering) was made from; kept as tables["part"] (not part of the blob:
    a generator's hint -- asmqp.LoopSplit deals the halves of a cut component to two wavefronts)."""
    A_p = [int(v) for v in A_p]
    A_i = [int(v) for v in A_i]
    P_cols = [int(v) for v in P_cols]
    nk, nnzA, nnzP = n + m, len(A_i), len(P_cols)
    perm, pinv, K_p, K_i, K_src, etree, L_p, L_i, ops = analyse_pattern(n, m, A_p, A_i, perm)
    nnzL = len(L_i)
    sizes = dict(n=n, m=m, nk=nk, nnzP=nnzP, nnzA=nnzA, nnzL=nnzL)
    rows, off = {}, 0
    for name, sz in _ROWS:
        rows[name] = off
        off += sizes[sz] if isinstance(sz, str) else sz
    nrows = off
    pidx = [-1] * n
    for k, j in enumerate(P_cols):
        pidx[j] = k
    # CSR view of A: entries of row i in ascending column order (= the order in which CSC traversal meets them)
    rlist = [[] for _ in range(m)]
    for j in range(n):
        for p in range(A_p[j], A_p[j + 1]):
            rlist[A_i[p]].append((j, p))
    Ar_p, Ar_j, Ar_k = [0], [], []
    for i in range(m):
        for (j, p) in rlist[i]:
            Ar_j.append(j)
            Ar_k.append(p)
        Ar_p.append(len(Ar_j))
    # factor schedule; init sources are workspace rows (off-diagonal KKT entries are A values; P is diagonal)
    fi_p, fi_b, fi_src, fe_p, fe_c, fe_new = [0], [], [], [0], [], []
    for op in ops:
        for (b, p) in op["init"]:
            src = K_src[p]
            assert src[0] == 'A', "off-diagonal KKT entries come from A (P is diagonal)"
            fi_b.append(b)
            fi_src.append(rows["AS"] + src[1])
        fi_p.append(len(fi_b))
        for (c, upd, new) in op["elim"]:
            assert [j for j, _ in upd] == list(range(L_p[c], new))
            fe_c.append(c)
            fe_new.append(new)
        fe_p.append(len(fe_c))
    # CSR view of L (row r: entries L[r, c] in ascending c = the order the column-oriented forward solve applies them)
    lrows = [[] for _ in range(nk)]
    for c in range(nk):
        for j in range(L_p[c], L_p[c + 1]):
            lrows[L_i[j]].append((c, j))
    Lr_p, Lr_j, Lr_k = [0], [], []
    for r in range(nk):
        for (c, j) in lrows[r]:
            Lr_j.append(c)
            Lr_k.append(j)
        Lr_p.append(len(Lr_j))
    # ---- level schedules for the wave-per-robot kernel: level(c) = 1 + max level over the row pattern of c, so the
    # entries of row r (forward solve) live in lower levels and the entries of column r (backward solve) in higher ones
    lev = [0] * nk
    for c in range(nk):
        lev[c] = 1 + max([lev[j] for (j, _) in lrows[c]], default=-1)
    nlev = max(lev) + 1
    lev_nodes = sorted(range(nk), key=lambda c: (lev[c], c))
    lev_p = [0]
    for l in range(nlev):
        lev_p.append(lev_p[-1] + sum(1 for c in range(nk) if lev[c] == l))
    ent_col = [0] * nnzL
    for c in range(nk):
        for j in range(L_p[c], L_p[c + 1]):
            ent_col[j] = c
    elev_ent = sorted(range(nnzL), key=lambda e: (lev[ent_col[e]], e))
    elev_p = [0]
    for l in range(nlev):
        elev_p.append(elev_p[-1] + sum(1 for e in range(nnzL) if lev[ent_col[e]] == l))
    # K source of every L position: the A entry of the permuted KKT at (row, col) = (L row, L col), or -1 (fill-in)
    lidx = {}
    for c in range(nk):
        for j in range(L_p[c], L_p[c + 1]):
            lidx[(L_i[j], c)] = j
    l_ksrc = [-1] * nnzL
    for jj in range(nk):
        for p in range(K_p[jj], K_p[jj + 1]):
            ii = K_i[p]
            if ii != jj:
                l_ksrc[lidx[(jj, ii)]] = K_src[p][1]       # ('A', index)
    # right-looking dot products: L[i,c] = (K[i,c] - sum_j L[i,j] L[c,j] D[j]) / D[c], D[c] = K[c,c] - sum_j L[c,j]^2 D[j]
    rowmap = [dict((c, k) for (c, k) in lrows[r]) for r in range(nk)]
    ft_p, ft_a, ft_b, ft_j = [0], [], [], []
    for e in range(nnzL):
        i, c = L_i[e], ent_col[e]
        for (j, kcj) in lrows[c]:
            if j in rowmap[i]:
                ft_a.append(rowmap[i][j]); ft_b.append(kcj); ft_j.append(j)
        ft_p.append(len(ft_a))
    fd_p, fd_a, fd_j = [0], [], []
    for c in range(nk):
        for (j, kcj) in lrows[c]:
            fd_a.append(kcj); fd_j.append(j)
        fd_p.append(len(fd_a))
    A_j = [0] * nnzA
    for j in range(n):
        for p in range(A_p[j], A_p[j + 1]):
            A_j[p] = j
    tables = dict(pinv=pinv, pidx=pidx, A_p=A_p, A_i=A_i, Ar_p=Ar_p, Ar_j=Ar_j, Ar_k=Ar_k, fi_p=fi_p, fi_b=fi_b,
                  fi_src=fi_src, fe_p=fe_p, fe_c=fe_c, fe_new=fe_new, L_p=L_p, L_i=L_i, Lr_p=Lr_p, Lr_j=Lr_j,
                  Lr_k=Lr_k, perm=perm, A_j=A_j, lev_p=lev_p, lev_nodes=lev_nodes, elev_p=elev_p, elev_ent=elev_ent,
                  l_ksrc=l_ksrc, l_col=ent_col, ft_p=ft_p, ft_a=ft_a, ft_b=ft_b, ft_j=ft_j, fd_p=fd_p, fd_a=fd_a, fd_j=fd_j)
    # blob: [HEADER_WORDS header | tables]; header = sizes, nrows, then (table offset) x len(_TABLES), then row offsets
    hdr = [n, m, nk, nnzP, nnzA, nnzL, nrows, nlev]
    if parts is not None:
        tables["part"] = [int(v) for v in parts]
    body, offs = [], []
    pos = HEADER_WORDS
    for name in _TABLES:
        offs.append(pos)
        t = tables[name]
        body += t
        pos += len(t)
    hdr += offs
    hdr += [rows[name] for name, _ in _ROWS]
    assert len(hdr) <= HEADER_WORDS
    hdr += [0] * (HEADER_WORDS - len(hdr))
    blob = np.asarray(hdr + body, dtype=np.int32)
    return QPStructure(n, m, nk, nnzP, nnzA, nnzL, P_cols, A_p, A_i, perm, pinv, K_p, K_i, K_src, etree, L_p, L_i,
                       ops, tables, blob, rows, nrows)


def csc_pattern(dense_mask):
    """CSC (A_p, A_i) of a boolean [m][n] mask."""
    dense_mask = np.asarray(dense_mask, bool)
    A_p, A_i = [0], []
    for j in range(dense_mask.shape[1]):
        A_i += [int(i) for i in np.nonzero(dense_mask[:, j])[0]]
        A_p.append(len(A_i))
    return A_p, A_i


_PARTS_CACHE = {}


def bisect_parts(n, m, A_p, A_i, min_size=64, max_sep=2, hold=None):
    key = (n, m, tuple(A_p), tuple(A_i), min_size, max_sep, tuple(sorted((k, tuple(sorted(v))) for k, v in (hold or {}).items())))
    if key not in _PARTS_CACHE:
        _PARTS_CACHE[key] = _bisect_parts(n, m, A_p, A_i, min_size, max_sep, hold)
    return list(_PARTS_CACHE[key])


def _bisect_parts(n, m, A_p, A_i, min_size, max_sep, hold=None):
    """For every KKT vertex (variables, then rows): 0 / 1 = half A / half B of its component, 2 = the separator, for the
    components of >= min_size unknowns that a separator of <= max_sep vertices cuts into two balanced halves (all such
    separators are tried; the one with the lightest heavier side wins); 0 everywhere else. An unknown that hangs off one
    other unknown (a box row on its variable) travels with it, the ones of a separator vertex with half A."""
    import itertools
    nk = n + m
    adj = [set() for _ in range(nk)]
    for j in range(n):
        for q in range(A_p[j], A_p[j + 1]):
            adj[j].add(n + A_i[q])
            adj[n + A_i[q]].add(j)
    vc, rc = qp_components(n, m, A_p, A_i)
    comp = list(vc) + list(rc)
    part = [0] * nk
    for c in range(max(comp) + 1):
        V = [v for v in range(nk) if comp[v] == c]
        core = [v for v in V if len(adj[v]) > 1]
        wt = {v: 1 + sum(1 for u in adj[v] if len(adj[u]) == 1) for v in core}
        if len(V) < min_size or len(core) < 4:
            continue

        def pieces(S):
            seen, out = set(S), []
            for v in core:
                if v in seen:
                    continue
                st, piece = [v], []
                seen.add(v)
                while st:
                    x = st.pop()
                    piece.append(x)
                    for u in adj[x]:
                        if u in wt and u not in seen:
                            seen.add(u)
                            st.append(u)
                out.append(piece)
            return sorted(out, key=lambda pc: (-sum(wt[v] for v in pc), pc[0]))
        cands = []
        for size in range(1, max_sep + 1):
            for S in itertools.combinations(core, size):
                pcs = pieces(S)
                if len(pcs) < 2:
                    continue
                bins, load = [[], []], [sum(wt[v] for v in S), 0]          # (the separator is half A's work)
                for pc in pcs:
                    b_ = 0 if load[0] <= load[1] else 1
                    bins[b_].append(pc)
                    load[b_] += sum(wt[v] for v in pc)
                cands.append((max(load), size, S, bins))
        if not cands:
            continue
        # among the (nearly) best balanced: the one whose half B reaches into the separator through the fewest entries of L
        # (each is an exchange between the two wavefronts), then the least fill
        lightest = min(c_[0] for c_ in cands)
        best = None
        loc = {v: q for q, v in enumerate(V)}
        sub_c = [{loc[u] for u in adj[v]} for v in V]         # (the component alone, renumbered: min-fill is quadratic)
        hold_c = {loc[v]: {loc[u] for u in us} for v, us in (hold or {}).items() if v in loc}
        short = sorted((c_ for c_ in cands if c_[0] <= lightest + 2), key=lambda c_: c_[:3])[:max(32, 24000 // len(V))]     # (each costs a min-fill run: quadratic in the component)
        for (ld, size, S, bins) in short:
            inB = {v for pc in bins[1] for v in pc}
            order = [V[q] for q in symbolic.min_fill_ordering(sub_c, last=[loc[v] for v in S], hold=hold_c)]
            g = {v: set(adj[v]) for v in V}
            done, nnz, cross = set(), 0, 0
            for v in order:
                nb = [u for u in g[v] if u not in done]
                nnz += len(nb)
                if v in inB:
                    cross += sum(1 for u in nb if u in S)
                for a_ in nb:
                    g[a_].update(x for x in nb if x != a_)
                done.add(v)
            key = (cross, nnz, ld, size, S)
            if best is None or key < best[0]:
                best = (key, S, bins)
        if best is None:
            continue
        _, S, bins = best
        for b_, bin_ in enumerate(bins):
            for pc in bin_:
                for v in pc:
                    part[v] = b_
                    for u in adj[v]:
                        if len(adj[u]) == 1:
                            part[u] = b_
        for v in S:
            part[v] = 2
    return part


def bisect_ordering(n, m, A_p, A_i, min_size=64, max_sep=2, parts=None, hold=None):
    """Elimination ordering for a QP whose large components are CHAINS (an MPC horizon): min-fill walks a chain from one end
    to the other -- an elimination tree that is one long spine, nothing for two wavefronts to share. Here every large
    component is cut in two by a small vertex separator S (bisect_parts), both halves are ordered by min-fill with S held
    back, and the result is laid out half A, half B, S: the fill of a chain eliminated from both ends, an elimination tree of
    two subtrees under S. (asmqp.LoopSplit gives the subtrees to two wavefronts; they meet twice per solve.)
    hold: symbolic.min_fill_ordering's numerical tie-break.
    Returns perm (position -> KKT index: variables, then rows)."""
    nk = n + m
    adj = [set() for _ in range(nk)]
    for j in range(n):
        for q in range(A_p[j], A_p[j + 1]):
            adj[j].add(n + A_i[q])
            adj[n + A_i[q]].add(j)
    vc, rc = qp_components(n, m, A_p, A_i)
    comp = list(vc) + list(rc)
    part = bisect_parts(n, m, A_p, A_i, min_size, max_sep, hold) if parts is None else list(parts)
    perm = []
    for c in range(max(comp) + 1):
        S = [v for v in range(nk) if comp[v] == c and part[v] == 2]
        sub = [v for v in symbolic.min_fill_ordering([adj[v] if comp[v] == c else set() for v in range(nk)], last=S, hold=hold)
               if comp[v] == c]
        perm += sorted(sub, key=lambda v: part[v])             # (stable: min-fill order inside each part)
    assert sorted(perm) == list(range(nk))
    return perm


def qp_components(n, m, A_p, A_i):
    """Connected components of a QP with DIAGONAL P: variables that share a constraint row belong together. Returns
    (component of each variable, component of each row), components numbered by decreasing number of variables (ties: by
    first variable). A QP with several components is several independent QPs solved side by side -- ADMM, the Ruiz
    equilibration's row / column norms and the LDL' factorisation never couple them."""
    par = list(range(n + m))

    def find(a):
        while par[a] != a:
            par[a] = par[par[a]]
            a = par[a]
        return a
    for j in range(n):
        for q in range(A_p[j], A_p[j + 1]):
            ra, rb = find(j), find(n + A_i[q])
            if ra != rb:
                par[ra] = rb
    members = {}
    for j in range(n):
        members.setdefault(find(j), []).append(j)
    order = sorted(members, key=lambda r: (-len(members[r]), members[r][0]))
    num = {r: c for c, r in enumerate(order)}
    rows = []
    for i in range(m):
        r = find(n + i)
        assert r in num, "a constraint row without a variable"
        rows.append(num[r])
    return [num[find(j)] for j in range(n)], rows
