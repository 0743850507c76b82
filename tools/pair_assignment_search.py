import sys, random, numpy as np
sys.path.insert(0,'/root/repo')
from robobee3d_amd import asmx, asmgen, symbolic
plan = asmx.Plan()
s = plan.s
print("asmx fwd", len(plan.fwd), "bwd", len(plan.bwd), "coef words", len(plan.coef), "nboth", plan.nboth)
ents = [(s.L_i[j], c, j) for c in range(s.nk) for j in range(s.L_p[c], s.L_p[c+1])]
def cost(home):
    # home[k] = (lane, reg)
    tot = 0
    for ops in ([(r,c) for r,c,j in ents], [(c,r) for r,c,j in ents]):
        g = {}
        for d,sr in ops:
            (ld,rd),(ls,rs) = home[d], home[sr]
            g.setdefault((rd,rs),[0,0])[ld]+=1
        tot += sum(max(a,b) for a,b in g.values())
    return tot
home = list(plan.home)
print("static cost of asmx homes:", cost(home))
# which unknowns are x/y members (paired) vs singles
xs, zs = plan.xs, plan.zs
import math, time
nx, nc, nk = s.nx, s.nc, s.nk
# unknown k (permuted) -> original index o = perm[k]; x-part if o < nx
isx = [s.perm[k] < nx for k in range(nk)]
NXR, NZR = 23, 20
def anneal(seed, iters=400000, T0=2.0, T1=0.02):
    rnd = random.Random(seed)
    # slots: x-part: (lane, reg) reg in 0..22 ; z-part: reg in 100..119 (distinct index spaces)
    xslots = [(l, r) for r in range(NXR) for l in (0,1)]
    zslots = [(l, 100 + r) for r in range(NZR) for l in (0,1)]
    # start from asmx homes mapped
    home = {}
    occ = {}
    for k in range(nk):
        l, r = plan.home[k]
        h = (l, r - asmx.VWX) if isx[k] else (l, 100 + r - asmx.VWZ)
        home[k] = h; occ[h] = k
    cur = cost(home); best = cur; besth = dict(home)
    ks = list(range(nk))
    for it in range(iters):
        T = T0 * (T1/T0) ** (it/iters)
        k = rnd.choice(ks)
        slots = xslots if isx[k] else zslots
        h2 = rnd.choice(slots)
        h1 = home[k]
        if h1 == h2: continue
        k2 = occ.get(h2)
        home[k] = h2; occ[h2] = k
        if k2 is not None:
            home[k2] = h1; occ[h1] = k2
        else:
            del occ[h1]
        c = cost(home)
        if c <= cur or rnd.random() < math.exp((cur - c)/T):
            cur = c
            if c < best: best = c; besth = dict(home)
        else:
            home[k] = h1; occ[h1] = k
            if k2 is not None:
                home[k2] = h2; occ[h2] = k2
            else:
                del occ[h2]
    return best, besth
t=time.time()
for seed in [int(sys.argv[1])]:
    b, h = anneal(seed, iters=600000, T0=1.5, T1=0.03)
    print(seed, b, time.time()-t)
