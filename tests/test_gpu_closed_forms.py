"""The scalar closed forms of the align kernel's epilogue, evaluated ON THE DEVICE (C ABI cvo_selftest_*), against
known answers that do not come from the oracle: numpy polynomial roots, scipy expm / logm
(tests/golden/closed_forms.json, made by tests/golden/make_golden.py) and hand-written limits.  Device and oracle share the
source text of these functions, so oracle-vs-device agreement alone would be the same code compiled twice."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "closed_forms.json")) as f:
        return json.load(f)


def test_device_cubic_step_matches_numpy_roots(hiplib, kat):
    # cvo.cpp:76-92,317-333: smallest strictly positive real root, else min_step 0.2, clamp 0.8
    from cvo_slam_amd import api
    rows = [c["coef"] + [0.2] for c in kat["cubic"]]
    got = api.selftest_cubic_step(rows)
    for case, g in zip(kat["cubic"], got):
        root = case["smallest_positive_real_root"]
        want = 0.2 if root is None else min(root, 0.8)
        assert g == pytest.approx(want, rel=2e-5, abs=1e-7), (case["coef"], g, want)


def test_device_cubic_step_fallbacks(hiplib):
    """E = 0 (division by zero in the companion matrix, cvo.cpp:86) with and without nonzero lower coefficients, no positive
    root, roots beyond the 0.8 clamp: the branches the alignment itself rarely takes."""
    from cvo_slam_amd import api
    rows = [[0.0, 1.0, -1.0, 0.5, 0.2],        # leading coefficient 0 although the rest has roots: min_step
            [0.0, 0.0, 0.0, 0.0, 0.2],
            [1.0, 6.0, 11.0, 6.0, 0.2],        # roots -1, -2, -3: none positive -> min_step
            [1.0, -6.0, 11.0, -6.0, 0.2],      # roots 1, 2, 3 -> clamp 0.8
            [1.0, 0.0, 1.0, 0.0, 0.3],         # roots 0, +-i: zero is not > 0 -> min_step (0.3 here)
            [2.0, -1.0, 0.0, 0.0, 0.2]]        # roots 0, 0, 0.5 -> 0.5
    got = api.selftest_cubic_step(rows)
    np.testing.assert_allclose(got, [0.2, 0.2, 0.2, 0.8, 0.3, 0.5], rtol=1e-6)


def test_device_exp_sek3_matches_scipy_expm(hiplib, kat):
    from cvo_slam_amd import api
    rows = [c["omega"] + c["v"] + [c["dt"]] for c in kat["exp"]]
    dR, dT = api.selftest_exp_sek3(rows)
    for case, R, t in zip(kat["exp"], dR, dT):
        # the reference evaluates (1-cos)/theta^2 and (dt*theta-sin)/theta^3 in f32 (LieGroup.cpp:176-179): ~1e-5 on dT
        np.testing.assert_allclose(R.ravel(), case["R"], atol=1e-5)
        np.testing.assert_allclose(t, case["t"], atol=5e-5)


def test_device_exp_sek3_small_angle_branch(hiplib):
    """LieGroup.cpp:168-170 (Q3): theta < 1e-6 -> R = I and Jl = I, so dT = v, NOT dt*v -- on the device; just above the
    tolerance the regular branch gives dT ~ dt*v."""
    from cvo_slam_amd import api
    v = [0.3, -0.2, 0.1]
    dR, dT = api.selftest_exp_sek3([[1e-7, 0, 0] + v + [0.25], [0, 0, 0] + v + [0.7], [2e-6, 0, 0] + v + [0.25]])
    np.testing.assert_array_equal(dR[0], np.eye(3, dtype=np.float32)); np.testing.assert_array_equal(dT[0], np.float32(v))
    np.testing.assert_array_equal(dR[1], np.eye(3, dtype=np.float32)); np.testing.assert_array_equal(dT[1], np.float32(v))
    np.testing.assert_allclose(dT[2], 0.25 * np.float32(v), rtol=1e-3)
    np.testing.assert_allclose(dR[2], np.eye(3), atol=1e-6)


def test_device_dist_se3_matches_scipy_logm(hiplib, kat):
    from cvo_slam_amd import api
    rows = [c["dR"] + c["dT"] for c in kat["dist"]]
    got = api.selftest_dist_se3(rows)
    for case, g in zip(kat["dist"], got):
        assert g == pytest.approx(case["frob_log"], rel=2e-3, abs=2e-7), case       # f32 dR carries ~1e-7 absolute noise
    ident = api.selftest_dist_se3([[1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]])
    assert ident[0] == 0.0
