"""Oracle vs the committed golden fixtures, its own execution modes, and the cloud-slot
state machine of the reference (cvo.cpp:345-386, 461-618)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import make_tf, rot_trans_err


def load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("name", ["small_pair_11.npz", "small_pair_12.npz", "small_pair_13.npz"])
def test_oracle_reproduces_golden_trace(oracle, name):
    g = load(name)
    o = oracle.OracleCvo()
    o.set_pcd(g["fixed_xyz"], g["fixed_feat"]); o.set_pcd(g["moving_xyz"], g["moving_feat"])
    rc, tr = o.align(trace_cap=400)
    assert rc == 0 and len(tr) == len(g["trace_nnz"])
    np.testing.assert_array_equal([r["nnz"] for r in tr], g["trace_nnz"])
    np.testing.assert_array_equal(np.array([r["omega"] for r in tr]), g["trace_omega"])
    np.testing.assert_array_equal(np.array([r["step"] for r in tr], np.float32), g["trace_step"])
    st = o.get_state()
    np.testing.assert_array_equal(st["transform"], g["final_transform"])
    assert st["iter"] == int(g["iter"]) and st["A_nonzero"] == int(g["A_nonzero"])


def test_golden_ell_schedule_and_stop(oracle):
    # cvo.cpp:810-812: ell used by iteration k is 0.15 (k<=3), 0.10 (4..10), 0.06 (11..20), 0.03 (21+)
    g = load("tum_pair_0.npz")
    ell = g["trace_ell"]
    for k, l in enumerate(ell):
        want = 0.15 if k <= 3 else 0.10 if k <= 10 else 0.06 if k <= 20 else 0.03
        assert l == np.float32(want)
    assert int(g["iter"]) == len(ell) - 1                                  # iter = k at the break
    assert g["trace_dist"][-1] < 1e-5 or g["trace_dist"][-1] == -1         # stop B (or stop A)


def test_oracle_full_size_modes_agree(oracle):
    """KD-tree + OpenMP mode (the timed CPU baseline) gives the same pose as the
    single-thread brute-force mode that wrote the fixture."""
    g = load("tum_pair_0.npz")
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=4)
    o.set_pcd(g["fixed_xyz"], g["fixed_feat"]); o.set_pcd(g["moving_xyz"], g["moving_feat"])
    rc, _ = o.align()
    st = o.get_state()
    re, te = rot_trans_err(st["transform"], g["final_transform"])
    assert re <= 1e-6 and te <= 1e-6
    assert st["iter"] == int(g["iter"])
    rc, sc = o.compute_innerproduct(st["transform"])
    assert sc["inn_post"][1] == int(g["inn_post"][1])
    assert sc["inn_post"][0] == pytest.approx(float(g["inn_post"][0]), rel=1e-6)
    np.testing.assert_allclose(sc["post_hessian"], g["post_hessian"], rtol=1e-3, atol=1e-3)


def test_not_initialized_and_empty(oracle):
    o = oracle.OracleCvo()
    rc, _ = o.match(np.zeros((4, 3), np.float32), np.zeros((5, 4), np.float32))
    assert rc == 1                                                         # "cvo not initialized !", cvo.cpp:463-466
    g = load("small_pair_11.npz")
    o.set_pcd(g["fixed_xyz"], g["fixed_feat"])
    rc, _ = o.align()
    assert rc == 2                                                         # no moving cloud yet


def test_no_overlap_stops_at_iteration_zero(oracle):
    # clouds 10 m apart: A is empty, omega = v = 0 -> stop A at k=0 (cvo.cpp:782), B=C=D=E=0 -> step = min_step
    g = load("small_pair_11.npz")
    o = oracle.OracleCvo()
    o.set_pcd(g["fixed_xyz"], g["fixed_feat"])
    far = g["moving_xyz"] + np.array([10, 0, 0], np.float32)
    o.set_pcd(far, g["moving_feat"])
    rc, tr = o.align(trace_cap=8)
    assert rc == 0 and len(tr) == 1 and tr[0]["nnz"] == 0 and tr[0]["step"] == pytest.approx(0.2)
    st = o.get_state()
    assert st["iter"] == 0
    np.testing.assert_array_equal(st["transform"], np.eye(3, 4, dtype=np.float32))


def test_ell_carries_over_between_calls(oracle):
    # Q1: ell is never reset; a second alignment on the same object starts with the ell the first one left
    g = load("small_pair_12.npz")
    o = oracle.OracleCvo()
    o.set_pcd(g["fixed_xyz"], g["fixed_feat"])
    o.match(g["moving_xyz"], g["moving_feat"])
    left = o.get_state()["ell"]
    assert left in (np.float32(0.10), np.float32(0.06), np.float32(0.03))  # the first call ran past k=2
    o.set_pcd(g["moving_xyz"], g["moving_feat"])
    rc, tr = o.align(trace_cap=50)
    ells = [r["ell"] for r in tr]
    assert ells[0] == left                                                 # not reset to 0.15
    if len(ells) > 4:
        assert ells[4] == np.float32(0.10)                                 # the schedule re-raises ell at k>2


def test_slot_state_machine(oracle):
    """update_fixed_pcd / update_previous_pcd / reset_keyframe / reset_initial move clouds
    exactly as cvo.cpp:578-618."""
    g = load("small_pair_13.npz")
    A = (g["fixed_xyz"], g["fixed_feat"]); B = (g["moving_xyz"], g["moving_feat"])
    o = oracle.OracleCvo()
    o.set_pcd(*A); o.set_pcd(*B)
    rc, r1 = o.function_inner_product(oracle.SLOT_MOVING, None, oracle.SLOT_FIXED)
    o.update_fixed_pcd()                                                   # MOVING -> FIXED
    assert o.function_inner_product(oracle.SLOT_MOVING, None, oracle.SLOT_FIXED)[0] == 2
    o.set_pcd(*A)                                                          # new MOVING = A, FIXED = B
    rc, r2 = o.function_inner_product(oracle.SLOT_FIXED, None, oracle.SLOT_MOVING)
    assert r1[1] == r2[1] and r1[0] == pytest.approx(r2[0], rel=1e-6)      # <B,A> pairs, same set
    o.update_previous_pcd()                                                # MOVING -> PREVIOUS
    odom = make_tf([0, 0, 1], 0.01, [0.01, 0, 0])
    o.set_pcd(*B)
    o.reset_keyframe(odom)                                                 # FIXED <- PREVIOUS (=A), MOVING (=B) -> PREVIOUS, transform <- odom
    np.testing.assert_array_equal(o.get_state()["transform"], odom)
    assert o.function_inner_product(oracle.SLOT_FIXED, None, oracle.SLOT_PREVIOUS)[0] == 0
    back = o.reset_initial(make_tf([1, 0, 0], 0.02, [0, 0.01, 0]))         # R,T <- (transform*odom)^-1
    st = o.get_state()
    M = np.eye(4); M[:3] = odom; N = np.eye(4); N[:3] = make_tf([1, 0, 0], 0.02, [0, 0.01, 0])
    inv = np.linalg.inv(M @ N)
    np.testing.assert_allclose(st["R"], inv[:3, :3], atol=2e-6); np.testing.assert_allclose(st["T"], inv[:3, 3], atol=2e-6)
    np.testing.assert_allclose(back, (M @ N)[:3], atol=2e-6)
