"""Wrench-linearisation step (template/uprightmpc2/funapprox.c): oracle pinned
bit-for-bit against the compiled reference (CPU); HIP path and the WLCon drop-in class checked
against the golden sequence on the GPU."""
import numpy as np
import pytest

from conftest import golden


def _args(g):
    return (g["u0"], g["umin"], g["umax"], g["dumax"], g["Qw"], float(g["controlRate"]), g["popts"])


def test_wl_oracle_bit_identical_to_reference(oracle_built):
    g = golden("wl_step.npz")
    o = oracle_built.WLOracle(*_args(g), dtype=np.float32)
    for k in range(len(g["h0"])):
        u1, w0 = o.update(g["h0"][k], g["pdotdes"][k])
        assert np.array_equal(u1, g["u1"][k]) and np.array_equal(w0, g["w0"][k]), k
    # the sequence hits both the rate limit and the frozen-at-the-box branch
    du = np.abs(g["u1"] - g["pre_u0"])
    assert np.any(np.isclose(du[:, 0], 5.0)) and np.any(du[:, 0] == 0)


def test_wl_fp64_oracle_tracks_fp32(oracle_built):
    g = golden("wl_step.npz")
    o = oracle_built.WLOracle(*_args(g), dtype=np.float64)
    for k in range(len(g["h0"])):
        o.set_u0(g["pre_u0"][k])
        u1, w0 = o.update(g["h0"][k], g["pdotdes"][k])
        np.testing.assert_allclose(w0, g["w0"][k], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(u1, g["u1"][k], rtol=1e-5, atol=2e-3)  # gradient step x1e3 amplifies fp32 rounding


@pytest.mark.gpu
def test_wl_batched_kernel_matches_reference_golden():
    import torch
    from robobee3d_amd.batch import BatchWLCon
    g = golden("wl_step.npz")
    n = len(g["h0"])
    for dtype, tol in ((torch.float32, 2e-3), (torch.float64, 2e-3)):
        wl = BatchWLCon(n, *_args(g), dtype=dtype)
        wl.u.copy_(torch.as_tensor(g["pre_u0"].T.copy()))
        u, w0 = wl.update(g["h0"].T.copy(), g["pdotdes"].T.copy())
        np.testing.assert_allclose(w0.cpu().numpy().T, g["w0"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(u.cpu().numpy().T, g["u1"], rtol=1e-5, atol=tol)
    # fp64 kernel vs fp64 oracle: tight
    import oraclebind
    o = oraclebind.WLOracle(*_args(g), dtype=np.float64)
    wl = BatchWLCon(n, *_args(g), dtype=torch.float64)
    wl.u.copy_(torch.as_tensor(g["pre_u0"].T.astype(np.float64)))
    u, w0 = wl.update(g["h0"].T.astype(np.float64), g["pdotdes"].T.astype(np.float64))
    for k in range(n):
        o.set_u0(g["pre_u0"][k])
        u1, w = o.update(g["h0"][k], g["pdotdes"][k])
        # the C ABI carries limits / coefficients as float (WLCon_t, funapprox.h:37-41): dumax/controlRate is
        # rounded to fp32 there and kept in fp64 by the oracle -> 2e-10 on a rate-limited step
        np.testing.assert_allclose(u.cpu().numpy()[:, k], u1, rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(w0.cpu().numpy()[:, k], w, rtol=1e-12, atol=1e-13)


@pytest.mark.gpu
def test_wlcon_dropin_sequence():
    """uprightmpc2py.WLCon through wlConInit/wlConUpdate: state carried in the caller's struct."""
    from robobee3d_amd.uprightmpc2py import WLCon
    g = golden("wl_step.npz")
    wl = WLCon(*_args(g))
    for k in range(48):
        u1, w0 = wl.update(g["h0"][k], g["pdotdes"][k])
        assert u1.shape == (4,) and w0.shape == (6,)
        np.testing.assert_allclose(w0, g["w0"][k], rtol=5e-5, atol=5e-5)
        np.testing.assert_allclose(u1, g["u1"][k], rtol=1e-5, atol=5e-3)
