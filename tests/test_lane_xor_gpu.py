"""ccm_slam_amd/csrc/lane_xor.h (round 6): every xor butterfly of the solvers moves its values by DPP (lane ^ 1, 2, 4, 8) and by gfx950's v_permlane16_swap /
v_permlane32_swap (lane ^ 16, 32) instead of ds_bpermute.  The sums must be the SAME sums: the moved value is the partner's value bit for bit, a butterfly step gives
v + partner with the bits of the __shfl_xor form, and the wave sum equals the __shfl_xor butterfly 32 ... 1 — on values chosen so that a wrong pairing or a wrong order of
the steps shows (magnitudes spread over 30 decades, signed zeros, a denormal)."""
import ctypes as C
import os

import numpy as np
import pytest

from ccm_slam_amd import optimizer
from ccm_slam_amd._lib import Context

pytestmark = pytest.mark.gpu


def _hooks():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(optimizer.__file__)), "libccm_testhooks.so"))
    lib.ccm_debug_lane_xor.restype = C.c_int
    lib.ccm_debug_lane_xor.argtypes = [C.c_void_p] * 4
    return lib


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lane_moves_equal_shfl_xor(seed):
    ctx = Context(0)
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(64) * 10.0 ** rng.integers(-15, 15, 64)
    v[3] = 0.0; v[35] = -0.0; v[17] = 5e-324; v[60] = -v[28]
    out = np.zeros((6, 3, 64)); sums = np.zeros((3, 64))
    rc = _hooks().ccm_debug_lane_xor(ctx.handle, v.ctypes.data, out.ctypes.data, sums.ctypes.data)
    assert rc == 0
    lanes = np.arange(64)
    for k, mask in enumerate((1, 2, 4, 8, 16, 32)):
        partner = v[lanes ^ mask]
        assert np.array_equal(out[k, 2].view(np.uint64), partner.view(np.uint64)), f"__shfl_xor {mask}"
        assert np.array_equal(out[k, 0].view(np.uint64), partner.view(np.uint64)), f"from_partner<{mask}>"
        assert np.array_equal(out[k, 1].view(np.uint64), (v + partner).view(np.uint64)), f"add_partner<{mask}>"
    assert np.array_equal(sums[0].view(np.uint64), sums[1].view(np.uint64))
    assert len(set(sums[0].view(np.uint64).tolist())) == 1   # every lane ends with the same bits
    assert np.array_equal(sums[2], np.cumsum((np.arange(64) * 37 % 101) - 20).astype(np.float64))   # the DPP prefix sum (ORB's compactions)
