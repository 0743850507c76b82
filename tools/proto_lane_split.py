"""Prototype for DESIGN.md 9.1 (two lanes per robot): how many instructions the triangular solves need when an op of\nlane 0 and an op of lane 1 share an instruction (same destination / source register, source read own-lane, swapped or\nbroadcast), and how many consistent packed pairs a better slot matching gives the CURRENT one-lane design."""
import sys, random
sys.path.insert(0,'/root/repo')
from robobee3d_amd import symbolic, asmgen
s = symbolic.analyse(3)
nk = s.nk
ents = [(s.L_i[j], c, j) for c in range(nk) for j in range(s.L_p[c], s.L_p[c+1])]   # (row, col)
xs, zs, xinv, zinv = asmgen.slot_maps(s)
# initial slots from the (x,y) symmetric slot maps: unknown k -> original index -> register slot; pair = aligned regs
def w_slot(k):
    o = s.perm[k]
    return ('x', xs[o]) if o < s.nx else ('z', zs[o - s.nx])
byslot = {}
for k in range(nk):
    byslot[w_slot(k)] = k
slots = []
for part, n in (('x', s.nx + 1), ('z', s.nc + 1)):
    for b in range(0, n, 2):
        a_, b_ = byslot.get((part, b)), byslot.get((part, b + 1))
        slots.append((a_, b_))
singles = [p for p in slots if p[0] is None or p[1] is None]
print("slots", len(slots), "with a hole:", singles)
# fix holes: pair the two leftover singles together
left = [k for p in singles for k in p if k is not None]
slots = [p for p in slots if p[0] is not None and p[1] is not None] + [tuple(left)] if len(left) == 2 else slots
assert sorted(k for p in slots for k in p) == list(range(nk))

def cost(slots, detail=False):
    slot_of, lane_of = {}, {}
    for si, (a, b) in enumerate(slots):
        slot_of[a] = slot_of[b] = si
        lane_of[a], lane_of[b] = 0, 1
    fwd, bwd = {}, {}
    cross = 0
    for (r, c, j) in ents:
        e = fwd.setdefault((slot_of[r], slot_of[c]), [0, 0]); e[lane_of[r]] += 1
        e = bwd.setdefault((slot_of[c], slot_of[r]), [0, 0]); e[lane_of[c]] += 1
        cross += lane_of[r] != lane_of[c]
    f = sum(max(v) for v in fwd.values()); b = sum(max(v) for v in bwd.values())
    if detail:
        return f, b, cross
    return f + b

print("initial (x,y)-symmetric slots: fwd instr, bwd instr, cross-lane entries:", cost(slots, True))
random.seed(1)
best = list(slots); bc = cost(best)
for it in range(60000):
    i, j = random.sample(range(len(best)), 2)
    cand = list(best)
    a, b = cand[i]; c, d = cand[j]
    mode = random.randrange(4)
    if mode == 0: cand[i], cand[j] = (a, c), (b, d)
    elif mode == 1: cand[i], cand[j] = (a, d), (c, b)
    elif mode == 2: cand[i] = (b, a)
    else: cand[i], cand[j] = (c, b), (a, d)
    cc = cost(cand)
    if cc <= bc:
        best, bc = cand, cc
print("after local search:", cost(best, True), "total", bc)

# ---- consistent pairs for the CURRENT packed design: rows are slot partners AND columns are slot partners
def npairs(slots):
    slot_of, lane_of = {}, {}
    for si, (a, b) in enumerate(slots):
        slot_of[a] = slot_of[b] = si
        lane_of[a], lane_of[b] = 0, 1
    groups = {}
    for (r, c, j) in ents:
        if slot_of[r] == slot_of[c]:
            continue
        groups.setdefault((slot_of[r], slot_of[c]), []).append((lane_of[r], lane_of[c]))
    n = 0
    for g in groups.values():
        # entries on this slot-edge: (row member, col member); two pair up if both members differ
        s_ = set(g)
        for (a, b) in [((0, 0), (1, 1)), ((0, 1), (1, 0))]:
            if a in s_ and b in s_:
                n += 1
    return n
print("consistent pairs, symmetric slots:", npairs(slots))
random.seed(2)
best = list(slots); bn = npairs(best)
for it in range(80000):
    i, j = random.sample(range(len(best)), 2)
    cand = list(best)
    a, b = cand[i]; c, d = cand[j]
    mode = random.randrange(3)
    if mode == 0: cand[i], cand[j] = (a, c), (b, d)
    elif mode == 1: cand[i], cand[j] = (a, d), (c, b)
    else: cand[i], cand[j] = (c, b), (a, d)
    cn = npairs(cand)
    if cn >= bn:
        best, bn = cand, cn
print("after local search:", bn, "pairs ->", 213 - bn, "instructions per direction (now", 213 - 63, ")")
