"""Longer exercise of the quad forms (round 4): fp32 B = 16 384 / 1 000 (ragged: 8 live quads in the last wave) / 1, fp64
B = 4 096 / 37; 2 000 closed-loop steps each, every robot must stay finite, converge to hover and keep its R orthonormal;
the lane form of the same batch must end within the closed-loop band of it."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
t0 = time.time()
for dt, ndt, B, K, reps in ((torch.float32, np.float32, 16384, 500, 4), (torch.float32, np.float32, 1000, 500, 4), (torch.float32, np.float32, 1, 500, 4),
                            (torch.float64, np.float64, 4096, 500, 4), (torch.float64, np.float64, 37, 500, 4)):
    st, ref = hover_initial_conditions(B, 13, ndt)
    end = {}
    for form in ("quad", "lane"):
        m = BatchUprightMPC(B, dt, plant_mode=1 if dt == torch.float32 else 0)
        m.set_step_kernel(form)
        m.set_state(st, ref)
        for _ in range(reps):
            m.rollout(K)
        torch.cuda.synchronize()
        s = m.state.cpu().numpy().astype(np.float64); stt = m.status.cpu().numpy()
        assert np.isfinite(s).all() and np.linalg.norm(s[0:3], axis=0).max() < 0.05, (dt, B, form)
        R = s[3:12].reshape(3, 3, B)
        assert np.abs(np.einsum("crb,drb->cdb", R, R) - np.eye(3)[:, :, None]).max() < (1e-3 if dt == torch.float32 else 1e-9)
        end[form] = s
        kn = m.kernel_name
        if form == "quad":
            assert "quad" in kn.lower(), kn
    d = np.abs(end["quad"] - end["lane"]).max()
    assert d < (2e-3 if dt == torch.float32 else 1e-7), d
    print("%s B=%d %dx%d steps: quad and lane form converge, |state diff| %.2e (%.1fs)" % (str(dt)[6:], B, reps, K, d, time.time() - t0), flush=True)
print("soak quad ok")
