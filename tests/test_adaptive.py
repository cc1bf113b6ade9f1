"""Adaptive-ell variant (acvo::align, thirdparty/cvo/src/adaptive_cvo.cpp:490-555): the oracle's restatement
on the CPU, and the HIP path (cvo_adaptive_align) against it on the GPU."""
import numpy as np
import pytest

from helpers import rot_trans_err


def _normalised(pair):
    """The variant's constants (c_ell = 0.5) presume colour features scaled to [0, 1] (adaptive_cvo.cpp:41; Q7)."""
    return pair.fixed.xyz, pair.fixed.feat / np.float32(255.0), pair.moving.xyz, pair.moving.feat / np.float32(255.0)


def test_oracle_adaptive_align_converges_and_moves_ell(oracle):
    from cvo_slam_amd import synth
    p = synth.make_small_pair(11, n=600)
    rc, r = oracle.adaptive_align(*_normalised(p), trace_cap=200)
    assert rc == 0 and 3 <= r["iter"] < 200
    re0, te0 = rot_trans_err(np.eye(3, 4), p.true_transform)
    re, te = rot_trans_err(r["transform"], p.true_transform)
    assert re < re0 / 3 and te < te0 / 2                               # the known camera motion is recovered
    ells = [t["ell"] for t in r["trace"]]
    assert ells[0] == pytest.approx(0.1) and min(ells) >= np.float32(0.0391) and max(ells) <= 0.15   # adaptive_cvo.cpp:27-30, 541-545
    assert len(set(ells)) > 2                                          # the length scale really moves
    # raw 0..255 features: nothing passes the colour gate (d2_c_thres = 2.39), omega = v = 0, stop A at k = 0 (why the reference never ran it on its own clouds)
    rc, r0 = oracle.adaptive_align(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat, trace_cap=4)
    assert rc == 0 and r0["iter"] == 0 and r0["trace"][0]["nnz_xy"] == 0 and r0["trace"][0]["nnz_xx"] == p.fixed.n


def test_oracle_adaptive_reproduces_the_unfilled_ayy_rows(oracle):
    """adaptive_cvo.cpp:218-226 never fills sum_diff_yy_2: Ayy adds to dl only through its rows from num_fixed on (:243-266).
    Dropping the LAST rows of the fixed cloud turns Ayy rows into contributors: dl of the first iteration changes by exactly
    what those rows add, and with num_moving <= num_fixed the moving cloud's own spread never enters."""
    from cvo_slam_amd import synth
    p = synth.make_small_pair(12, n=400)
    fx, ff, mx, mf = _normalised(p)
    rc, a = oracle.adaptive_align(fx, ff, mx, mf, trace_cap=1)
    keep = fx.shape[0] - 60
    rc2, b = oracle.adaptive_align(fx[:keep], ff[:, :keep], mx, mf, trace_cap=1)
    assert rc == 0 and rc2 == 0
    assert a["trace"][0]["nnz_yy"] == b["trace"][0]["nnz_yy"]          # Ayy itself does not depend on the fixed cloud
    assert a["trace"][0]["dl"] != b["trace"][0]["dl"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,drop", [(11, 600, 0), (12, 450, 70), (13, 300, 0)])
def test_hip_adaptive_align_matches_the_oracle(hiplib, oracle, seed, n, drop):
    from cvo_slam_amd import synth, api
    p = synth.make_small_pair(seed, n=n)
    fx, ff, mx, mf = _normalised(p)
    if drop:                                                           # num_moving > num_fixed: the rows of Ayy that do count (adaptive_cvo.cpp:243-266)
        fx, ff = fx[:-drop], np.ascontiguousarray(ff[:, :-drop])
    rc, want = oracle.adaptive_align(fx, ff, mx, mf, trace_cap=400)
    got = api.adaptive_align(fx, ff, mx, mf, trace_cap=400)
    assert rc == 0 and got["iter"] == want["iter"] and len(got["trace"]) == len(want["trace"])
    for k, (g, w) in enumerate(zip(got["trace"], want["trace"])):
        assert (g["nnz_xy"], g["nnz_xx"], g["nnz_yy"]) == (w["nnz_xy"], w["nnz_xx"], w["nnz_yy"]), k
        np.testing.assert_allclose(g["omega"], w["omega"], rtol=1e-5, atol=1e-9); np.testing.assert_allclose(g["v"], w["v"], rtol=1e-5, atol=1e-9)
        assert g["ell"] == pytest.approx(w["ell"], rel=1e-6) and g["step"] == pytest.approx(w["step"], rel=1e-5)
        if np.isfinite(w["dl"]):
            assert g["dl"] == pytest.approx(w["dl"], rel=1e-5, abs=1e-9), k
        else:
            assert not np.isfinite(g["dl"]) or np.isnan(w["dl"])
    rot, tr = rot_trans_err(got["transform"], want["transform"])
    assert rot <= 1e-4 and tr <= 1e-4
    assert got["ell"] == pytest.approx(want["ell"], rel=1e-6)


@pytest.mark.gpu
def test_hip_adaptive_align_full_size_pair(hiplib, oracle):
    """BASELINE-size clouds (640x480 TUM shape, ~3 k points, colours scaled to [0, 1]): the rows of every sweep are spread over the
    device (one-wave workgroups, grid over rows), the iteration trace is the oracle's."""
    import time
    from cvo_slam_amd import synth, api
    p = synth.make_pair(0)
    assert p.fixed.n > 2500 and p.moving.n > 2500
    fx, ff, mx, mf = _normalised(p)
    rc, want = oracle.adaptive_align(fx, ff, mx, mf, trace_cap=400)
    api.adaptive_align(fx, ff, mx, mf, trace_cap=4)                     # (first call: module load)
    t0 = time.perf_counter(); got = api.adaptive_align(fx, ff, mx, mf, trace_cap=400); dt = time.perf_counter() - t0
    assert rc == 0 and got["iter"] == want["iter"] and len(got["trace"]) == len(want["trace"]) and len(want["trace"]) > 10
    for k, (g, w) in enumerate(zip(got["trace"], want["trace"])):
        assert (g["nnz_xy"], g["nnz_xx"], g["nnz_yy"]) == (w["nnz_xy"], w["nnz_xx"], w["nnz_yy"]), k
        np.testing.assert_allclose(g["omega"], w["omega"], rtol=1e-5, atol=1e-9); np.testing.assert_allclose(g["v"], w["v"], rtol=1e-5, atol=1e-9)
        assert g["ell"] == pytest.approx(w["ell"], rel=1e-6) and g["step"] == pytest.approx(w["step"], rel=1e-5)
        assert g["dl"] == pytest.approx(w["dl"], rel=1e-5, abs=1e-9), k
    rot, tr = rot_trans_err(got["transform"], want["transform"])
    assert rot <= 1e-4 and tr <= 1e-4
    print(f"adaptive align, {p.fixed.n} x {p.moving.n} points, {len(got['trace'])} iterations: {1e3 * dt:.1f} ms")
    assert dt < 1.0                                                    # (one workgroup took ~3 ms per iteration)


@pytest.mark.gpu
def test_hip_adaptive_raw_features_stop_at_once(hiplib):
    from cvo_slam_amd import synth, api
    p = synth.make_small_pair(14, n=300)
    got = api.adaptive_align(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat, trace_cap=4)
    assert got["iter"] == 0 and got["trace"][0]["nnz_xy"] == 0 and got["trace"][0]["nnz_xx"] == 300
    np.testing.assert_array_equal(got["transform"], np.eye(3, 4, dtype=np.float32))
