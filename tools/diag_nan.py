import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _load_seq_into
from robobee3d_amd.batch import BatchUprightMPC
seq = np.load(os.path.join(ROOT, "tests/golden/nan_branch.npz"))
n = len(seq["p0"])
for mi in (50, 1, 2, 10):
    mpc = BatchUprightMPC(n, torch.float32, maxIter=mi)
    _load_seq_into(mpc, seq, torch)
    mpc.update()
    info = mpc.info.cpu().numpy()
    print("maxIter", mi, "status", mpc.status.cpu().numpy()[[4, 9]], "pri", info[0, [4, 9]], "dua", info[1, [4, 9]])
    c = mpc.ctrl.cpu().numpy()
    print("   |x| max", np.abs(c[:45, [4, 9]]).max(0), "|y| max", np.abs(c[45:84, [4, 9]]).max(0), "nan?", np.isnan(c[:, [4, 9]]).any(0))
print("ref dua", seq["dua_res"][[4, 9]], "pri", seq["pri_res"][[4, 9]])
