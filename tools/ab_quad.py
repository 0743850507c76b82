#!/usr/bin/env python3
"""A/B of the two forms of the all-assembly fp32 step stream on ONE box: one lane per robot (umpc_rollout_asm_kernel) against
one lane quad per robot (umpc_rollout_asm_quad_kernel, robobee3d_amd/asmquad.py), ms per closed-loop step by batch size.
usage: python tools/ab_quad.py [K]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions_device

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
print("batch   lane form ms/step   quad form ms/step   quad/lane   (K = %d steps per launch, RK4 plant, 50 iterations)" % K)
for B in (1, 64, 1024, 4096, 8192, 16384, 32768, 65536):
    row = []
    for form in ("lane", "quad"):
        m = BatchUprightMPC(B, torch.float32, plant_mode=1)
        m.set_step_kernel(form)
        st, ref, _ = hover_initial_conditions_device(B, 20201118, torch.float32)
        m.set_state(st, ref)
        m.rollout(K)                      # warm: code object, clocks
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
