#!/usr/bin/env python3
"""A/B of the two forms of the fp64 assembly ADMM phase on ONE box (BASELINE config 2's shape and around it): one lane per
robot against one lane quad per robot (robobee3d_amd/asmquad64.py), ms per closed-loop step. usage: python tools/ab_quad64.py [K]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions_device

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
print("batch   lane form ms/step   quad form ms/step   quad/lane   (fp64, Euler + expm plant, K = %d steps per launch)" % K)
for B in (1, 256, 1024, 4096, 8192):
    row = []
    for form in ("lane", "quad"):
        m = BatchUprightMPC(B, torch.float64, plant_mode=0)
        m.set_step_kernel(form)
        st, ref, _ = hover_initial_conditions_device(B, 20201117, torch.float64)
        m.set_state(st, ref)
        m.rollout(K)
        m.rollout(K)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            m.rollout(K)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
        row.append(best)
        del m
    print("%6d   %14.4f      %14.4f      %6.3f" % (B, row[0], row[1], row[1] / row[0]), flush=True)
