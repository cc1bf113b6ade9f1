"""Parity of the HIP path (through the C ABI, include/cvo_hip.h) against the oracle and
the committed golden fixtures.  Tolerance from BASELINE.json's north_star: rotation
<= 1e-4 rad, translation <= 1e-4 m on the final SE(3); per-iteration quantities are
compared much tighter because the survivor arithmetic follows the oracle's float
sequence."""
import contextlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import make_tf, rot_trans_err

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-4     # rad
TRANS_TOL = 1e-4   # m


def load(name):
    return np.load(os.path.join(GOLDEN, name))


@contextlib.contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def gpu_align(hiplib, fixed, moving, wgs=0, trace_cap=0, params=None):
    g = hiplib.Cvo(params)
    g.set_workgroups(wgs)
    g.set_pcd(*fixed); g.set_pcd(*moving)
    tr = g.align(trace_cap=trace_cap)
    return g, tr


def oracle_align(oracle, fixed, moving, trace_cap=0, params=None):
    o = oracle.OracleCvo(params)
    o.set_pcd(*fixed); o.set_pcd(*moving)
    rc, tr = o.align(trace_cap=trace_cap)
    assert rc == 0
    return o, tr


def assert_pose_close(a, b):
    re, te = rot_trans_err(a, b)
    assert re <= ROT_TOL and te <= TRANS_TOL, (re, te)


# ----------------------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("name", ["small_pair_11.npz", "small_pair_12.npz", "small_pair_13.npz"])
@pytest.mark.parametrize("wgs", [1, 3])
def test_golden_small_pairs_trace_and_pose(hiplib, name, wgs):
    gd = load(name)
    g, tr = gpu_align(hiplib, (gd["fixed_xyz"], gd["fixed_feat"]), (gd["moving_xyz"], gd["moving_feat"]), wgs=wgs, trace_cap=400)
    assert_pose_close(g.transform, gd["final_transform"])
    assert g.get_iteration_number() == int(gd["iter"])
    assert g.get_A_nonzero() == int(gd["A_nonzero"])
    n = len(gd["trace_nnz"])
    assert len(tr) == n
    np.testing.assert_array_equal([r["nnz"] for r in tr], gd["trace_nnz"])                       # the same sparse set every iteration
    np.testing.assert_allclose(np.array([r["omega"] for r in tr]), gd["trace_omega"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(np.array([r["v"] for r in tr]), gd["trace_v"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(np.array([r["BCDE"] for r in tr]), gd["trace_BCDE"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose([r["step"] for r in tr], gd["trace_step"], rtol=1e-5)
    np.testing.assert_array_equal(np.array([r["ell"] for r in tr], np.float32), gd["trace_ell"])
    for r in tr:
        assert r["candidates"] >= r["nnz"]
    g.close()


def test_golden_full_size_pair_pose_and_scores(hiplib):
    gd = load("tum_pair_0.npz")
    g, _ = gpu_align(hiplib, (gd["fixed_xyz"], gd["fixed_feat"]), (gd["moving_xyz"], gd["moving_feat"]))
    assert_pose_close(g.transform, gd["final_transform"])
    assert g.get_iteration_number() == int(gd["iter"])
    assert g.get_fixed_and_moving_number() == (gd["fixed_xyz"].shape[0], gd["moving_xyz"].shape[0])
    sc = g.compute_innerproduct(gd["final_transform"])
    for key, gk in (("inn_pre", "inn_pre"), ("inn_post", "inn_post"), ("inn_fixed_pcd", "inn_fixed"), ("inn_moving_pcd", "inn_moving")):
        assert sc[key][1] == int(gd[gk][1]), key                                                  # pair counts are integers: exact
        assert sc[key][0] == pytest.approx(float(gd[gk][0]), rel=1e-6), key
    assert sc["inliers"] == int(gd["inliers"])
    assert sc["cos_angle"] == pytest.approx(float(gd["cos_angle"]), rel=1e-6)
    # the reference keeps an f32 Hessian summed in arbitrary order (cvo.cpp:622,707): relative 1e-3
    np.testing.assert_allclose(sc["post_hessian"], gd["post_hessian"], rtol=1e-3, atol=1e-3 * np.abs(gd["post_hessian"]).max())
    g.close()


# ----------------------------------------------------------------------------- live oracle
@pytest.mark.parametrize("seed,n", [(101, 64), (102, 257), (103, 700), (104, 1500)])
def test_seeded_pairs_vs_oracle(hiplib, oracle, seed, n):
    from cvo_slam_amd import synth
    p = synth.make_small_pair(seed, n=n)
    fixed = (p.fixed.xyz, p.fixed.feat)
    moving = (p.moving.xyz[: max(8, n - 13)], p.moving.feat[:, : max(8, n - 13)])               # ragged: N != M
    o, otr = oracle_align(oracle, fixed, moving, trace_cap=2000)
    g, gtr = gpu_align(hiplib, fixed, moving, trace_cap=2000)
    ost = o.get_state()
    assert_pose_close(g.transform, ost["transform"])
    assert g.get_iteration_number() == ost["iter"] and g.get_A_nonzero() == ost["A_nonzero"]
    assert [r["nnz"] for r in gtr] == [r["nnz"] for r in otr]
    st = g.get_state()
    assert st["ell"] == np.float32(ost["ell"])
    np.testing.assert_allclose(st["R"], ost["R"], atol=1e-6); np.testing.assert_allclose(st["T"], ost["T"], atol=1e-6)
    g.close()


def test_workgroup_counts_tiles_and_overflow_fallback_agree(hiplib, oracle):
    """The same alignment under every decomposition knob: workgroups per pair (incl. a
    non-power-of-two and more workgroups than 64-row blocks), LDS tile smaller than the
    cloud (multi-tile streaming), flat candidate capacities so small that the dense
    per-row fallback runs, and candidate-list skins from 0 (cull every iteration)
    to 150 % (one cull per ell), rows re-sorted at every list refinement / never.  All must land on the
    oracle's pose and sparse-set sizes."""
    from cvo_slam_amd import synth
    p = synth.make_small_pair(55, n=900)
    fixed, moving = (p.fixed.xyz, p.fixed.feat), (p.moving.xyz, p.moving.feat)
    o, otr = oracle_align(oracle, fixed, moving, trace_cap=2000)
    want_nnz = [r["nnz"] for r in otr]
    ost = o.get_state()
    configs = [dict(wgs=1), dict(wgs=2), dict(wgs=7), dict(wgs=32), dict(wgs=1, CVO_HIP_TILE=128), dict(wgs=4, CVO_HIP_TILE=252),
               dict(wgs=1, CVO_HIP_FLAT_CAP=4), dict(wgs=5, CVO_HIP_FLAT_CAP=1, CVO_HIP_TILE=64), dict(wgs=3, CVO_HIP_FLAT_CAP=1),
               dict(wgs=2, CVO_HIP_SKIN=0.0), dict(wgs=2, CVO_HIP_SKIN=0.02), dict(wgs=4, CVO_HIP_SKIN=1.5),
               # depth-proportional list margin (DevParams::skin_alpha): alone, with the constant part, large, under the plane layout and with forced re-sorts
               dict(wgs=1, CVO_HIP_SKIN=0.0, CVO_HIP_SKIN_ALPHA=0.02), dict(wgs=2, CVO_HIP_SKIN=0.05, CVO_HIP_SKIN_ALPHA=0.0125), dict(wgs=3, CVO_HIP_SKIN=0.35, CVO_HIP_SKIN_ALPHA=0.15),
               dict(wgs=1, CVO_HIP_SKIN_ALPHA=0.01, CVO_HIP_Y_MODE=2), dict(wgs=2, CVO_HIP_SKIN=0.1, CVO_HIP_SKIN_ALPHA=0.03, CVO_HIP_RESORT=2), dict(wgs=1, CVO_HIP_SKIN_ALPHA=0.05, CVO_HIP_NO_YLDS=1),
               # lists built / filtered around extrapolated positions (DevParams::predict): off, far ahead, far ahead with wide margins, under the plane layout
               dict(wgs=1, CVO_HIP_PREDICT=0.0), dict(wgs=2, CVO_HIP_PREDICT=0.95, CVO_HIP_PREDICT_STEPS=50), dict(wgs=1, CVO_HIP_PREDICT=0.9, CVO_HIP_SKIN=0.3, CVO_HIP_SKIN_ALPHA=0.05, CVO_HIP_PREDICT_STEPS=50),
               dict(wgs=2, CVO_HIP_PREDICT=0.9, CVO_HIP_Y_MODE=2), dict(wgs=3, CVO_HIP_PREDICT=0.9, CVO_HIP_RESORT=2),
               dict(wgs=1, CVO_HIP_NO_YLDS=1), dict(wgs=3, CVO_HIP_NO_YLDS=1, CVO_HIP_TILE=128),          # transformed cloud in HBM/L2, not LDS
               dict(wgs=1, CVO_HIP_Y_MODE=2), dict(wgs=4, CVO_HIP_Y_MODE=2, CVO_HIP_TILE=256), dict(wgs=2, CVO_HIP_Y_MODE=0),   # 12-byte LDS layout / forced HBM
               dict(wgs=1, CVO_HIP_ROW_CAP=8), dict(wgs=2, CVO_HIP_ROW_CAP=16, CVO_HIP_SKIN=0.6),         # rows longer than the lists hold
               dict(wgs=1, CVO_HIP_WGS_PER_CU=2),                                                          # two 256-thread workgroups per CU
               dict(wgs=1, CVO_HIP_RESORT=2), dict(wgs=3, CVO_HIP_RESORT=2), dict(wgs=2, CVO_HIP_RESORT=2, CVO_HIP_Y_MODE=2),   # rows re-sorted at EVERY list refinement
               dict(wgs=2, CVO_HIP_RESORT=2, CVO_HIP_NO_YLDS=1, CVO_HIP_TILE=128), dict(wgs=1, CVO_HIP_RESORT=0)]
    for cfg in configs:
        envs = {k: v for k, v in cfg.items() if k.startswith("CVO_")}
        with env(**envs):
            g, gtr = gpu_align(hiplib, fixed, moving, wgs=cfg["wgs"], trace_cap=2000)
        assert [r["nnz"] for r in gtr] == want_nnz, cfg
        re, te = rot_trans_err(g.transform, ost["transform"])
        assert re <= 1e-6 and te <= 1e-6, (cfg, re, te)
        assert g.get_iteration_number() == ost["iter"], cfg
        g.close()


def test_full_size_pair_fresh_vs_oracle(hiplib, oracle):
    from cvo_slam_amd import synth
    p = synth.make_pair(3)
    fixed, moving = (p.fixed.xyz, p.fixed.feat), (p.moving.xyz, p.moving.feat)
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
    o.set_pcd(*fixed); o.set_pcd(*moving); rc, _ = o.align(); assert rc == 0
    g, _ = gpu_align(hiplib, fixed, moving)
    assert_pose_close(g.transform, o.get_state()["transform"])
    assert g.get_iteration_number() == o.get_state()["iter"]
    # size-independent properties at full size: <a,b> is symmetric in its pair set, self inner product counts >= n
    ab = g.function_inner_product(hiplib.api.SLOT_MOVING, None, hiplib.api.SLOT_FIXED)
    ba = g.function_inner_product(hiplib.api.SLOT_FIXED, None, hiplib.api.SLOT_MOVING)
    assert ab[1] == ba[1] and ab[0] == pytest.approx(ba[0], rel=1e-6)
    aa = g.function_inner_product(hiplib.api.SLOT_FIXED, None, hiplib.api.SLOT_FIXED)
    assert aa[1] >= p.fixed.n and aa[0] >= p.fixed.n * 0.01 * (1 - 1e-6)                         # every point pairs with itself: k*ck = sigma^2
    g.close()


def test_full_size_batch_properties_without_the_oracle(hiplib):
    """BASELINE-size pairs (640x480, ~3 k points) through size-independent properties of the path: the known camera motion
    is recovered, the result is frame-covariant (both clouds moved by S => S T S^-1), repeated launches and different
    workgroup counts give the same bits."""
    from cvo_slam_amd import synth
    n = 8
    pairs = [synth.make_pair(100 + i) for i in range(n)]
    B = hiplib.CvoBatch(n)
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    res = B.align(n)
    for p, r in zip(pairs, res):
        assert r["status"] == 0
        re, te = rot_trans_err(r["transform"], p.true_transform)
        re0, te0 = rot_trans_err(np.eye(3, 4), p.true_transform)
        assert re < 2e-3 and te < 3e-3, (re, te)                                # ~0.05 deg, ~1 mm on these scenes
        assert re < re0 / 5 and te < te0 / 5
    # determinism: the same launch again, and with 2 and 4 cooperating workgroups per pair
    for wgs in (1, 2, 4):
        B.set_workgroups(wgs); B.reset_states()
        again = B.align(n)
        for r, r2 in zip(res, again):
            np.testing.assert_array_equal(r["transform"], r2["transform"])
            assert (r["iterations_run"], r["A_nonzero"]) == (r2["iterations_run"], r2["A_nonzero"])
    B.close()
    # covariance: a rigid change of the common frame
    S = np.eye(4); S[:3] = make_tf([0.3, 1, 0.2], 0.2, [0.1, -0.05, 0.2])
    B2 = hiplib.CvoBatch(n)
    for i, p in enumerate(pairs):
        fx = (p.fixed.xyz.astype(np.float64) @ S[:3, :3].T + S[:3, 3]).astype(np.float32)
        mx = (p.moving.xyz.astype(np.float64) @ S[:3, :3].T + S[:3, 3]).astype(np.float32)
        B2.set_pair(i, fx, p.fixed.feat, mx, p.moving.feat)
    res2 = B2.align(n)
    for r, r2 in zip(res, res2):
        T = np.eye(4); T[:3] = r["transform"]
        re, te = rot_trans_err(r2["transform"], (S @ T @ np.linalg.inv(S))[:3])
        assert re < 5e-4 and te < 5e-4, (re, te)
    B2.close()


@pytest.mark.parametrize("changes", [
    dict(sp_thres=1.0e-3),                                   # wide radius: exponents beyond the range-reduction-free polynomial (general exp path)
    dict(sp_thres=9.0e-3, sigma=0.12),
    dict(c_ell=40.0, c_sigma=1.0),                           # the colour gate binds: most pairs fail it (NaN-marked entries)
    dict(c_ell=25.0, c_sigma=0.8, sp_thres=4.0e-3),
    dict(c=3.0, d=11.0, min_step=0.35),
    dict(ell=0.10, eps=2e-4, eps_2=5e-5),
    dict(max_iter=7)])
def test_non_default_parameters_vs_oracle(hiplib, oracle, changes):
    """Every hyper-parameter of cvo.cpp:35-51 away from its default: the kernel's fast paths are selected from the parameters
    (polynomial exp only when the exponent range allows it, colour factors kept per list entry), the results must not depend
    on which path runs."""
    from cvo_slam_amd import synth
    p = synth.make_small_pair(91, n=700)
    fixed, moving = (p.fixed.xyz, p.fixed.feat), (p.moving.xyz, p.moving.feat)
    gp, op = hiplib.default_params(), oracle.default_params()
    for k, v in changes.items():
        setattr(gp, k, v); setattr(op, k, v)
    o, otr = oracle_align(oracle, fixed, moving, trace_cap=2000, params=op)
    ost = o.get_state()
    for wgs in (1, 3):
        g, gtr = gpu_align(hiplib, fixed, moving, wgs=wgs, trace_cap=2000, params=gp)
        assert [r["nnz"] for r in gtr] == [r["nnz"] for r in otr], (changes, wgs)
        re, te = rot_trans_err(g.transform, ost["transform"])
        assert re <= 1e-6 and te <= 1e-6, (changes, wgs, re, te)
        if "max_iter" not in changes:
            assert g.get_iteration_number() == ost["iter"]
        assert g.get_A_nonzero() == ost["A_nonzero"]
        sg = g.compute_innerproduct(ost["transform"]); rc, so = o.compute_innerproduct(ost["transform"]); assert rc == 0
        assert sg["inn_post"][1] == so["inn_post"][1] and sg["inn_post"][0] == pytest.approx(so["inn_post"][0], rel=1e-5)
        g.close()


def test_dense_near_surface_keeps_the_list_path(hiplib, oracle):
    """A wall 0.6 m from the camera sampled every ~8 mm: ~450 neighbours per point inside the first radius.  The lists must
    hold that (no dense per-row fallback) and the result must still be the oracle's."""
    rng = np.random.default_rng(5)
    n = 3000
    uv = rng.uniform(-1, 1, size=(n, 2)) * np.array([0.28, 0.21])
    fixed_xyz = np.stack([uv[:, 0], uv[:, 1], 0.6 + 0.02 * np.sin(9 * uv[:, 0]) * np.cos(7 * uv[:, 1])], axis=1).astype(np.float32)
    feat = np.stack([128 + 80 * np.sin(11 * uv[:, 0]), 128 + 70 * np.cos(13 * uv[:, 1]), 128 + 60 * np.sin(9 * (uv[:, 0] + uv[:, 1])),
                     15 * np.cos(17 * uv[:, 0]), 15 * np.sin(19 * uv[:, 1])], axis=0).astype(np.float32)
    tf = make_tf([0.2, 1.0, 0.1], 0.012, [0.006, -0.004, 0.003])
    moving_xyz = ((fixed_xyz - tf[:, 3]) @ tf[:, :3] + rng.normal(0, 5e-4, size=fixed_xyz.shape)).astype(np.float32)
    fixed, moving = (fixed_xyz, feat), (moving_xyz, feat + rng.normal(0, 1.0, size=feat.shape).astype(np.float32))
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
    o.set_pcd(*fixed); o.set_pcd(*moving); rc, otr = o.align(trace_cap=400); assert rc == 0
    assert otr[0]["nnz"] > 150 * n                                         # really dense: > 150 nonzeros per row in the first iteration
    ost = o.get_state()
    B = hiplib.CvoBatch(1)
    B.set_pair(0, fixed[0], fixed[1], moving[0], moving[1])
    for wgs in (1, 4):
        B.set_workgroups(wgs); B.reset_states()
        r = B.align(1)[0]
        assert r["status"] == 0 and r["dense_fallbacks"] == 0, r
        re, te = rot_trans_err(r["transform"], ost["transform"])
        assert re <= 1e-6 and te <= 1e-6 and r["iter"] == ost["iter"] and r["A_nonzero"] == ost["A_nonzero"]
    B.close()


def test_clouds_far_from_the_origin_fall_back_to_the_constant_list_margin(hiplib, oracle):
    """The depth-proportional part of the list margin (DevParams::skin_alpha) is metres for clouds given in a frame whose origin is far away (a world frame,
    say): the lists of that margin do not fit, the pair takes the narrow constant margin instead -- no dense per-row sweeps, and the oracle's result."""
    from cvo_slam_amd import synth
    p = synth.make_small_pair(91, n=1400)
    shift = np.array([40.0, -25.0, 12.0], np.float32)                      # both clouds 48 m from the origin; the alignment itself is translation-invariant up to rounding
    fixed, moving = (p.fixed.xyz + shift, p.fixed.feat), (p.moving.xyz + shift, p.moving.feat)
    o = oracle.OracleCvo(); o.set_pcd(*fixed); o.set_pcd(*moving); rc, _ = o.align(); assert rc == 0
    ost = o.get_state()
    B = hiplib.CvoBatch(1)
    B.set_pair(0, fixed[0], fixed[1], moving[0], moving[1])
    for wgs in (1, 3):
        B.set_workgroups(wgs); B.reset_states()
        r = B.align(1)[0]
        assert r["status"] == 0 and r["dense_fallbacks"] == 0, r
        re, te = rot_trans_err(r["transform"], ost["transform"])
        assert re <= 1e-6 and te <= 1e-5 and r["iter"] == ost["iter"] and r["A_nonzero"] == ost["A_nonzero"], (re, te, r["iter"], ost["iter"])
    B.close()


def test_eth3d_shape_pair_tile_sweep(hiplib, oracle):
    """BASELINE config 5: ETH3D-shape 736x456 pair, ~9 k points per cloud (dense sampling).  The transformed cloud no longer
    fits in LDS (HBM/L2 path), a workgroup owns at most 4096 rows (G >= 3), and the cull tile is swept over
    512 ... 4096 columns: every setting must land on the oracle's pose, iteration count and final nnz."""
    from cvo_slam_amd import synth
    p = synth.make_pair(2, cam=synth.ETH3D)
    assert p.fixed.n > 8000 and p.moving.n > 8000
    fixed, moving = (p.fixed.xyz, p.fixed.feat), (p.moving.xyz, p.moving.feat)
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
    o.set_pcd(*fixed); o.set_pcd(*moving); rc, _ = o.align(); assert rc == 0
    ost = o.get_state()
    for cfg in [dict(wgs=0), dict(wgs=3, CVO_HIP_TILE=512), dict(wgs=4, CVO_HIP_TILE=1024), dict(wgs=8, CVO_HIP_TILE=2048), dict(wgs=32, CVO_HIP_TILE=4096)]:
        envs = {k: v for k, v in cfg.items() if k.startswith("CVO_")}
        with env(**envs):
            g, _ = gpu_align(hiplib, fixed, moving, wgs=cfg["wgs"])
        re, te = rot_trans_err(g.transform, ost["transform"])
        assert re <= 1e-6 and te <= 1e-6, (cfg, re, te)
        assert g.get_iteration_number() == ost["iter"] and g.get_A_nonzero() == ost["A_nonzero"], cfg
        g.close()


@pytest.mark.parametrize("points", [3400, 3700, 4000, 5200])
def test_clouds_between_the_layouts_keep_their_lists(hiplib, oracle, points):
    """Clouds a little larger than the 3 k-point shape, one workgroup per pair: more points per thread than the epilogue pre-loads for the next
    transform, and the sizes at which the resident layout changes.  Besides the oracle's pose, iteration count and nnz: the candidate lists are
    rebuilt a few times per pair, not at every iteration (a staleness bound fed with a wrong list position for the points behind the pre-loaded ones
    leaves every result right and takes three times as long)."""
    from cvo_slam_amd import synth
    p = synth.make_pair(3, cam=synth.ETH3D)
    def thin(c):
        idx = np.unique(np.linspace(0, c.n - 1, points).astype(np.int64))
        return np.ascontiguousarray(c.xyz[idx]), np.ascontiguousarray(c.feat[:, idx])   # xyz (n, 3), features (5, n) as cvo::set_pcd takes them
    fixed, moving = thin(p.fixed), thin(p.moving)
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
    o.set_pcd(*fixed); o.set_pcd(*moving); rc, _ = o.align(); assert rc == 0
    ost = o.get_state()
    for wgs in (1, 2):
        b = hiplib.CvoBatch(1)
        b.set_workgroups(wgs)
        b.set_pairs([(fixed[0], fixed[1], moving[0], moving[1])])
        b.reset_states(); b.align_async(1)
        r = b.wait(1)[0]
        assert r["status"] == 0, (points, wgs, r["status"])
        re, te = rot_trans_err(r["transform"], ost["transform"])
        assert re <= 1e-6 and te <= 1e-6, (points, wgs, re, te)
        assert r["iter"] == ost["iter"] and r["A_nonzero"] == ost["A_nonzero"], (points, wgs)
        masks, _ = b.last_cull_masks(1)
        assert bin(masks[0]).count("1") <= 8, (points, wgs, bin(masks[0]))
        b.close()


# ----------------------------------------------------------------------------- scores
def test_score_block_vs_oracle(hiplib, oracle):
    from cvo_slam_amd import synth
    p = synth.make_small_pair(77, n=800)
    fixed, moving = (p.fixed.xyz, p.fixed.feat), (p.moving.xyz, p.moving.feat)
    o, _ = oracle_align(oracle, fixed, moving)
    g, _ = gpu_align(hiplib, fixed, moving)
    tf = o.get_state()["transform"]
    rc, so = o.compute_innerproduct(tf); assert rc == 0
    sg = g.compute_innerproduct(tf)
    for key in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd"):
        assert sg[key][1] == so[key][1] and sg[key][2] == 0
        assert sg[key][0] == pytest.approx(so[key][0], rel=1e-6)
    assert sg["inliers"] == so["inliers"]
    assert sg["cos_angle"] == pytest.approx(so["cos_angle"], rel=1e-6)
    np.testing.assert_allclose(sg["post_hessian"], so["post_hessian"], rtol=1e-3, atol=1e-3 * np.abs(so["post_hessian"]).max())
    # un-regularised Hessian against the oracle's f64 accumulation of the same terms
    rc, Ho, inl_o, Hraw = o.se3_hessian(oracle.SLOT_MOVING, tf, oracle.SLOT_FIXED)
    Hg, inl_g = g.se3_hessian(hiplib.api.SLOT_MOVING, tf, hiplib.api.SLOT_FIXED)
    assert inl_g == inl_o
    # loop-closure score block (cvo.cpp:505-561)
    t1, t2, t3 = make_tf([0, 1, 0], 0.01, [0.01, 0, 0]), make_tf([1, 0, 0], 0.02, [0, 0.01, 0]), make_tf([0, 0, 1], 0.015, [0, 0, 0.01])
    rc, lo = o.compute_innerproduct_lc(t1, t2, t3, tf); assert rc == 0
    lg = g.compute_innerproduct_lc(t1, t2, t3, tf)
    for key in ("inn_prior", "inn_lc_prior", "inn_lc_pre", "inn_lc_post", "inn_fixed_pcd", "inn_moving_pcd"):
        assert lg[key][1] == lo[key][1]
        assert lg[key][0] == pytest.approx(lo[key][0], rel=1e-6)
    assert (lg["inliers_svd"], lg["inliers_pnpransac"]) == (lo["inliers_svd"], lo["inliers_pnpransac"])
    assert lg["cos_angle"] == pytest.approx(lo["cos_angle"], rel=1e-6)
    g.close()


@pytest.mark.parametrize("case", ["shuffled", "behind_camera", "tiny", "ragged_33", "wide_ell"])
def test_score_box_cull_edge_cases(hiplib, oracle, case):
    """The score kernels skip 32-point groups by bounding box (x, y, z, ray slope).  Point order, points at or behind the camera
    plane (no slope bound), clouds smaller than a group, a ragged last group and the widest radius (fresh object, ell = 0.15)
    must not change a single pair: counts equal the oracle's, sums agree to f32 rounding."""
    from cvo_slam_amd import synth
    rng = np.random.default_rng(5)
    p = synth.make_small_pair(31, n=700)
    fx, ff, mx, mf = p.fixed.xyz.copy(), p.fixed.feat.copy(), p.moving.xyz.copy(), p.moving.feat.copy()
    ell = 0.03
    if case == "shuffled":                                           # no scan order: boxes are loose, nothing else changes
        a, b = rng.permutation(fx.shape[0]), rng.permutation(mx.shape[0])
        fx, ff, mx, mf = fx[a], ff[:, a], mx[b], mf[:, b]
    elif case == "behind_camera":                                    # the whole scene pushed through z = 0
        fx[:, 2] -= 1.2; mx[:, 2] -= 1.2
        assert (fx[:, 2] < 0).any() and (fx[:, 2] > 0).any()
    elif case == "tiny":
        fx, ff, mx, mf = fx[:7], ff[:, :7], mx[:5], mf[:, :5]
        mx[:] = fx[:5] + 0.002
    elif case == "ragged_33":
        fx, ff, mx, mf = fx[:97], ff[:, :97], mx[:33], mf[:, :33]
        mx[:] = fx[:33] + np.float32(0.003)
    elif case == "wide_ell":
        ell = 0.15
    tf = make_tf([0.2, 1, 0.1], 0.01, [0.004, -0.002, 0.003])
    g = hiplib.Cvo(); g.set_pcd(fx, ff); g.set_pcd(mx, mf); g.set_state(np.eye(3), np.zeros(3), ell)
    o = oracle.OracleCvo(); o.set_pcd(fx, ff); o.set_pcd(mx, mf); o.set_state(np.eye(3), np.zeros(3), ell)
    for (sa, t, sb) in ((1, None, 0), (1, tf, 0), (0, None, 0), (1, None, 1), (0, tf, 1)):
        rc, want = o.function_inner_product(sa, t, sb); assert rc == 0
        got = g.function_inner_product(sa, t, sb)
        assert got[1] == want[1], (case, sa, sb)
        assert got[0] == pytest.approx(want[0], rel=2e-6), (case, sa, sb)
        rc, Ho, inl_o, _ = o.se3_hessian(sa, t, sb); assert rc == 0
        Hg, inl_g = g.se3_hessian(sa, t, sb)
        assert inl_g == inl_o, (case, sa, sb)
        np.testing.assert_allclose(Hg, Ho, rtol=1e-3, atol=1e-3 * max(np.abs(Ho).max(), 1e-30))
    rc, so = o.compute_innerproduct(tf); assert rc == 0
    sg = g.compute_innerproduct(tf)
    assert [sg[k][1] for k in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd")] == [so[k][1] for k in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd")]
    assert sg["inliers"] == so["inliers"]
    g.close()


def test_empty_overlap_scores(hiplib, oracle):
    # no pair within the radius: value 0, num forced to 1 (cvo.cpp:455-456), Hessian = identity (cvo.cpp:755)
    gd = load("small_pair_11.npz")
    g = hiplib.Cvo(); g.set_pcd(gd["fixed_xyz"], gd["fixed_feat"]); g.set_pcd(gd["moving_xyz"] + np.array([10, 0, 0], np.float32), gd["moving_feat"])
    assert g.function_inner_product(hiplib.api.SLOT_MOVING, None, hiplib.api.SLOT_FIXED) == (0.0, 1, 0)
    H, inl = g.se3_hessian(hiplib.api.SLOT_MOVING, None, hiplib.api.SLOT_FIXED)
    assert inl == 0
    np.testing.assert_array_equal(H, np.eye(6))
    tr = g.align(trace_cap=4)                                              # and align stops at k=0 with the pose untouched
    assert len(tr) == 1 and tr[0]["nnz"] == 0 and tr[0]["step"] == pytest.approx(0.2)
    assert g.get_iteration_number() == 0
    np.testing.assert_array_equal(g.transform, np.eye(3, 4, dtype=np.float32))
    g.close()


# ----------------------------------------------------------------------------- error behaviour
def test_error_conventions(hiplib):
    gd = load("small_pair_11.npz")
    g = hiplib.Cvo()
    with pytest.raises(hiplib.CvoError) as e:                                                    # cvo.cpp:463-466: "cvo not initialized !"
        g.match_odometry(gd["moving_xyz"], gd["moving_feat"])
    assert e.value.code == 1 and "not initialized" in str(e.value)
    assert not g.init
    g.set_pcd(gd["fixed_xyz"], gd["fixed_feat"])
    assert g.init and g.first_frame
    with pytest.raises(hiplib.CvoError) as e:                                                    # no moving cloud yet (Q8)
        g.align()
    assert e.value.code == 2
    with pytest.raises(hiplib.CvoError) as e:
        g.match_keyframe(np.zeros((0, 3), np.float32), np.zeros((5, 0), np.float32))              # empty moving cloud
    assert e.value.code == 2
    with pytest.raises(hiplib.CvoError) as e:
        g.set_pcd(np.zeros((70000, 3), np.float32), np.zeros((5, 70000), np.float32))
    assert e.value.code == 4
    g.first_frame = False
    assert not g.first_frame
    g.close()


def test_max_iter_leaves_iter_stale(hiplib, oracle):
    # Q4: `iter` is only written on a break; hitting MAX_ITER leaves the previous value
    gd = load("small_pair_12.npz")
    pg = hiplib.default_params(); pg.max_iter = 5
    po_ = oracle.default_params(); po_.max_iter = 5
    fixed, moving = (gd["fixed_xyz"], gd["fixed_feat"]), (gd["moving_xyz"], gd["moving_feat"])
    g, gtr = gpu_align(hiplib, fixed, moving, trace_cap=16, params=pg)
    o, otr = oracle_align(oracle, fixed, moving, trace_cap=16, params=po_)
    assert len(gtr) == len(otr) == 5
    assert g.get_iteration_number() == 0 == o.get_state()["iter"]
    assert_pose_close(g.transform, o.get_state()["transform"])
    g.close()


# ----------------------------------------------------------------------------- tracker call sequence
def test_tracker_sequence_replay(hiplib, oracle):
    """The call sequence of LocalTracker (local_tracker.cpp:228-251, 356-431, 506; SURVEY
    Appendix B) on two objects, three frames: warm starts (Q2), carried ell (Q1), slot moves."""
    from cvo_slam_amd import synth
    rng = np.random.default_rng(4)
    base = synth.make_small_pair(91, n=600)
    frames = [(base.fixed.xyz, base.fixed.feat)]
    for k in range(3):                                                     # frames drifting away from the keyframe
        tf = make_tf(rng.normal(size=3), 0.01 * (k + 1), 0.01 * (k + 1) * rng.normal(size=3)).astype(np.float64)
        xyz = ((base.moving.xyz.astype(np.float64) - tf[:, 3]) @ tf[:, :3]).astype(np.float32)
        frames.append((xyz, base.moving.feat))

    def run(make):
        odo, kf = make(), make()
        out = []
        odo.set_pcd(*frames[0]); kf.set_pcd(*frames[0])                    # :228,:231
        t_odo = odo.match_odometry(*frames[1])                             # :233
        s = odo.compute_innerproduct(np.asarray(t_odo, np.float32))        # :251
        out.append((t_odo, s))
        odo.update_fixed_pcd()                                             # :277
        kf.first_frame = False; kf.reset_transform(np.asarray(t_odo, np.float32))   # :330-333
        for f in frames[2:]:
            t_odo = odo.match_odometry(*f)                                 # :356
            s1 = odo.compute_innerproduct(np.asarray(t_odo, np.float32))   # :375
            odo.update_fixed_pcd()                                         # :403
            guess = kf.reset_initial(np.asarray(t_odo, np.float32))        # :407
            t_kf = kf.match_keyframe(*f)                                   # :415
            s2 = kf.compute_innerproduct(np.asarray(t_kf, np.float32))     # :431
            kf.update_previous_pcd()                                       # :506
            out.append((t_odo, s1)); out.append((t_kf, s2)); out.append((guess, None))
        return out

    class OAdapter:                                                        # same method names over the oracle
        def __init__(self): self.o = oracle.OracleCvo(); self.first_frame = True
        def set_pcd(self, x, f): self.o.set_pcd(x, f)
        def match_odometry(self, x, f): rc, t = self.o.match(x, f); assert rc == 0; return t
        match_keyframe = match_odometry
        def compute_innerproduct(self, t): rc, s = self.o.compute_innerproduct(t); assert rc == 0; return s
        def update_fixed_pcd(self): self.o.update_fixed_pcd()
        def update_previous_pcd(self): self.o.update_previous_pcd()
        def reset_transform(self, t): self.o.reset_transform(t)
        def reset_initial(self, t): return self.o.reset_initial(t)

    want = run(OAdapter)
    got = run(hiplib.Cvo)
    assert len(want) == len(got)
    for (tw, sw), (tg, sg) in zip(want, got):
        assert_pose_close(tg, tw)
        if sw is not None:
            assert sg["inn_post"][1] == sw["inn_post"][1] and sg["inliers"] == sw["inliers"]
            assert sg["inn_post"][0] == pytest.approx(sw["inn_post"][0], rel=1e-5)
            assert sg["cos_angle"] == pytest.approx(sw["cos_angle"], rel=1e-5)


def test_reset_keyframe_slot_moves(hiplib):
    gd = load("small_pair_13.npz")
    A, B = (gd["fixed_xyz"], gd["fixed_feat"]), (gd["moving_xyz"], gd["moving_feat"])
    S = hiplib.api
    g = hiplib.Cvo(); g.set_pcd(*A); g.set_pcd(*B)
    r1 = g.function_inner_product(S.SLOT_MOVING, None, S.SLOT_FIXED)
    g.update_fixed_pcd()
    with pytest.raises(hiplib.CvoError):
        g.function_inner_product(S.SLOT_MOVING, None, S.SLOT_FIXED)        # MOVING moved away
    g.set_pcd(*A)
    r2 = g.function_inner_product(S.SLOT_FIXED, None, S.SLOT_MOVING)
    assert r1[1] == r2[1]
    g.update_previous_pcd(); g.set_pcd(*B)
    odom = make_tf([0, 0, 1], 0.01, [0.01, 0, 0])
    g.reset_keyframe(odom)                                                 # FIXED <- PREVIOUS, MOVING -> PREVIOUS, transform <- odom
    np.testing.assert_array_equal(g.transform, odom)
    g.function_inner_product(S.SLOT_FIXED, None, S.SLOT_PREVIOUS)
    g.close()


def test_score_functions_on_host_clouds_match_slots_and_oracle(hiplib, oracle):
    """function_inner_product(point_cloud*, point_cloud*) / se3_Hessian(point_cloud*, point_cloud*, int&) as the reference
    declares them (cvo.hpp:222, 260): clouds handed in directly give what the slot forms and the oracle give."""
    S = hiplib.api
    gd = load("small_pair_12.npz")
    A, B = (gd["fixed_xyz"], gd["fixed_feat"]), (gd["moving_xyz"], gd["moving_feat"])
    g = hiplib.Cvo(); g.set_pcd(*A); g.set_pcd(*B)
    o = oracle.OracleCvo(); o.set_pcd(*A); o.set_pcd(*B)
    for ell in (0.15, 0.06):
        st = g.get_state(); g.set_state(st["R"], st["T"], ell)
        so = o.get_state(); o.set_state(so["R"], so["T"], ell)
        r_slot = g.function_inner_product(S.SLOT_MOVING, None, S.SLOT_FIXED)
        r_host = g.function_inner_product_clouds(B[0], B[1], A[0], A[1])
        rc, r_orc = o.function_inner_product(S.SLOT_MOVING, None, S.SLOT_FIXED)
        assert rc == 0 and r_host == r_slot
        assert r_host[1] == r_orc[1] and r_host[0] == pytest.approx(r_orc[0], rel=1e-6)
        H_slot, n_slot = g.se3_hessian(S.SLOT_MOVING, None, S.SLOT_FIXED, inliers=3)
        H_host, n_host = g.se3_hessian_clouds(B[0], B[1], A[0], A[1], inliers=3)
        rc, H_orc, n_orc, _ = o.se3_hessian(S.SLOT_MOVING, None, S.SLOT_FIXED)
        np.testing.assert_array_equal(H_host, H_slot)
        assert n_host == n_slot == n_orc + 3                              # the caller's counter accumulates (cvo.cpp:708)
    g.close()


def test_reset_keyframe_matches_the_oracle_state_machine(hiplib, oracle):
    """cvo.cpp:591-604 against orc_reset_keyframe, both branches: before any update_previous_pcd (FIXED <- MOVING) and after
    one (FIXED <- PREVIOUS, MOVING -> PREVIOUS); the slots are identified by what an alignment and the inner products on
    them return afterwards."""
    S = hiplib.api
    from cvo_slam_amd import synth
    clouds = [synth.make_small_pair(300 + i, n=350 + 40 * i) for i in range(2)]
    A = (clouds[0].fixed.xyz, clouds[0].fixed.feat); B = (clouds[0].moving.xyz, clouds[0].moving.feat)
    Cc = (clouds[1].moving.xyz, clouds[1].moving.feat); D = (clouds[1].fixed.xyz, clouds[1].fixed.feat)
    odom = make_tf([0.2, 0.1, 1], 0.012, [0.01, -0.004, 0.006])

    def drive(obj, fip, is_gpu):
        out = []
        obj.set_pcd(*A); obj.set_pcd(*B)
        obj.reset_keyframe(odom)                                  # no PREVIOUS yet: FIXED <- MOVING (cvo.cpp:593-595)
        out.append(np.array(obj.transform if is_gpu else obj.get_state()["transform"], np.float64))
        obj.set_pcd(*Cc)                                          # new MOVING against the moved FIXED (= B)
        out.append(fip(obj, S.SLOT_MOVING, S.SLOT_FIXED))
        obj.update_previous_pcd()                                 # MOVING -> PREVIOUS (cvo.cpp:584-589)
        obj.set_pcd(*D)
        obj.reset_keyframe(odom)                                  # FIXED <- PREVIOUS (= Cc), MOVING (= D) -> PREVIOUS (cvo.cpp:596-601)
        out.append(fip(obj, S.SLOT_FIXED, S.SLOT_PREVIOUS))
        obj.set_pcd(*A)
        out.append(fip(obj, S.SLOT_MOVING, S.SLOT_PREVIOUS))
        out.append(fip(obj, S.SLOT_MOVING, S.SLOT_FIXED))
        return out

    g = hiplib.Cvo()
    got = drive(g, lambda o, a, b: o.function_inner_product(a, None, b), True)
    o = oracle.OracleCvo()
    want = drive(o, lambda q, a, b: q.function_inner_product(a, None, b)[1], False)
    np.testing.assert_array_equal(got[0], np.float64(odom)); np.testing.assert_array_equal(want[0], np.float64(odom))
    for gv, wv in zip(got[1:], want[1:]):
        assert gv[1] == wv[1] and gv[0] == pytest.approx(wv[0], rel=1e-6), (gv, wv)      # pair counts exact, sums to f64-order noise
    g.close()


def test_many_threads_with_cooperating_workgroups_never_time_out(hiplib):
    """12 host threads, each with its own cvo object forced to 32 cooperating workgroups per pair: 384 workgroups wanted, 256 CUs.
    Workgroups of a pair wait for each other inside the kernel, so unbounded this can leave every launch partially resident
    until the in-kernel timeout (CVO_ERR_TIMEOUT).  The library bounds the cooperative workgroups in flight per device (later
    launches take fewer workgroups per pair): every call returns CVO_OK and -- results do not depend on G -- the same bits."""
    import threading
    from cvo_slam_amd import synth
    p = synth.make_pair(3)
    ref = hiplib.Cvo(); ref.set_workgroups(1); ref.set_pcd(p.fixed.xyz, p.fixed.feat); ref.set_pcd(p.moving.xyz, p.moving.feat); ref.align()
    want = ref.transform.copy(); want_it = ref.get_iteration_number(); ref.close()
    n_threads, rounds = 12, 3
    objs = []
    for _ in range(n_threads):
        g = hiplib.Cvo(); g.set_workgroups(32); g.set_pcd(p.fixed.xyz, p.fixed.feat); g.set_pcd(p.moving.xyz, p.moving.feat)
        objs.append(g)
    errors, results = [], [None] * n_threads
    barrier = threading.Barrier(n_threads)

    def work(i):
        try:
            for _ in range(rounds):
                objs[i].set_state(np.eye(3), np.zeros(3), 0.15)
                barrier.wait()                                       # all twelve launch together
                objs[i].align()
            results[i] = (objs[i].transform.copy(), objs[i].get_iteration_number())
        except Exception as e:                                       # CvoError(6) = CVO_ERR_TIMEOUT would land here
            errors.append((i, repr(e)))
            barrier.abort()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors, errors
    for tf, it in results:
        np.testing.assert_array_equal(tf, want); assert it == want_it
    for g in objs:
        g.close()


# ----------------------------------------------------------------------------- batches
def test_batch_matches_single_objects_and_warm_start(hiplib, oracle):
    from cvo_slam_amd import synth
    pairs = [synth.make_small_pair(200 + i, n=n) for i, n in enumerate((300, 450, 64, 900, 300, 777))]
    B = hiplib.CvoBatch(len(pairs))
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    warm = make_tf([0, 1, 0], 0.004, [0.002, 0, -0.001])
    B.set_state(4, warm[:, :3], warm[:, 3], 0.06)                          # reset_initial-style warm start + carried ell (Q1, Q2)
    res = B.align(len(pairs))
    info = B.last_launch()
    assert info["kernel_ms"] > 0 and info["iterations_total"] == sum(r["iterations_run"] for r in res)
    for i, (p, r) in enumerate(zip(pairs, res)):
        assert r["status"] == 0
        o = oracle.OracleCvo()
        o.set_pcd(p.fixed.xyz, p.fixed.feat); o.set_pcd(p.moving.xyz, p.moving.feat)
        if i == 4:
            o.set_state(warm[:, :3], warm[:, 3], 0.06)
        rc, _ = o.align(); assert rc == 0
        st = o.get_state()
        assert_pose_close(r["transform"], st["transform"])
        assert r["iter"] == st["iter"] and r["A_nonzero"] == st["A_nonzero"] and r["iterations_run"] == st["iter"] + 1
        assert r["ell"] == np.float32(st["ell"])
    # a second launch without reset continues from the converged state: one or two trips, pose unchanged
    res2 = B.align(len(pairs))
    for r, r2 in zip(res, res2):
        assert r2["iterations_run"] <= 3
        re, te = rot_trans_err(r["transform"], r2["transform"])
        assert re <= 1e-4 and te <= 1e-4
    # reset_states restores the initial states: identical results again
    B.reset_states()
    res3 = B.align(len(pairs))
    for r, r3 in zip(res, res3):
        np.testing.assert_array_equal(r["transform"], r3["transform"])
        assert r["iterations_run"] == r3["iterations_run"]
    B.close()


@pytest.mark.parametrize("wgs", [1, 2])
def test_batch_larger_than_the_gpu(hiplib, oracle, wgs):
    """More pairs than resident workgroup slots (256 CUs / workgroups per pair): every persistent workgroup (group) aligns
    several pairs one after the other and must start each from a clean state.  300 small pairs of varying size, each against
    its own oracle object."""
    from cvo_slam_amd import synth
    n = 300
    sizes = [64 + (37 * i) % 90 for i in range(n)]
    pairs = [synth.make_small_pair(5000 + i, n=sizes[i]) for i in range(n)]
    B = hiplib.CvoBatch(n); B.set_workgroups(wgs)
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    res = B.align(n)
    for i, (p, r) in enumerate(zip(pairs, res)):
        assert r["status"] == 0, i
        o = oracle.OracleCvo(); o.set_pcd(p.fixed.xyz, p.fixed.feat); o.set_pcd(p.moving.xyz, p.moving.feat)
        rc, _ = o.align(); assert rc == 0
        st = o.get_state()
        re, te = rot_trans_err(r["transform"], st["transform"])
        assert re <= 1e-6 and te <= 1e-6, (i, re, te)
        assert (r["iter"], r["A_nonzero"]) == (st["iter"], st["A_nonzero"]), i
    B.close()


def test_batch_loop_closure_verification_block(hiplib, oracle):
    """keyframe_graph.cpp:693-717 for a batch of candidates: fresh objects warm-started by reset_initial(lc_prior), aligned in
    one launch, then every pair's compute_innerproduct_lc block (6 inner products + 2 Hessians) in ONE score launch, with the
    reference's accept rule; checked pair by pair against single oracle objects driven the way the reference drives cvo::cvo."""
    from cvo_slam_amd import synth
    sizes = (300, 520, 64, 900, 410, 777, 333, 640, 250, 1000)              # detectLoopClousure_top10: up to 10 candidates (11 requests > 8 too)
    pairs = [synth.make_small_pair(400 + i, n=n) for i, n in enumerate(sizes)]
    n = len(pairs)
    priors = np.stack([make_tf([0, 1, 0], 0.002 * (i + 1), [0.001 * i, 0, -0.001]) for i in range(n)])
    lc_priors = np.stack([make_tf([1, 0, 0], 0.003 * (i % 4), [0, 0.002, 0.001 * (i % 3)]) for i in range(n)])
    lc_priors2 = np.stack([make_tf([0, 0, 1], 0.004, [0.002, -0.001 * (i % 2), 0]) for i in range(n)])
    B = hiplib.CvoBatch(n)
    single = []
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
        o = oracle.OracleCvo()
        guess = o.reset_initial(lc_priors[i])                                  # keyframe_graph.cpp:696 on a fresh object
        o.set_pcd(p.fixed.xyz, p.fixed.feat); o.set_pcd(p.moving.xyz, p.moving.feat)
        st0 = o.get_state()
        B.set_state(i, st0["R"], st0["T"], st0["ell"])
        single.append(o)
    res = B.align(n)
    got = B.compute_innerproduct_lc(priors, lc_priors, lc_priors2)
    assert len(got) == n
    for i, (o, r, g) in enumerate(zip(single, res, got)):
        rc, _ = o.align(); assert rc == 0
        st = o.get_state()
        assert_pose_close(r["transform"], st["transform"])
        rc, want = o.compute_innerproduct_lc(priors[i], lc_priors[i], lc_priors2[i], st["transform"]); assert rc == 0
        for key in ("inn_prior", "inn_lc_prior", "inn_lc_pre", "inn_lc_post", "inn_fixed_pcd", "inn_moving_pcd"):
            assert g[key][1] == want[key][1], (i, key)
            assert g[key][0] == pytest.approx(want[key][0], rel=1e-5), (i, key)
        assert (g["inliers_svd"], g["inliers_pnpransac"]) == (want["inliers_svd"], want["inliers_pnpransac"])
        assert g["cos_angle"] == pytest.approx(want["cos_angle"], rel=1e-5)
        np.testing.assert_allclose(g["post_hessian"], want["post_hessian"], rtol=1e-3, atol=1e-3 * np.abs(want["post_hessian"]).max())
        post, pre, lcp, pri = want["inn_lc_post"][0], want["inn_lc_pre"][0], want["inn_lc_prior"][0], want["inn_prior"][0]
        margin = min(abs(post - pre), abs(post - lcp), abs(post - pri)) / max(abs(post), 1e-30)
        if margin > 1e-4 and abs(want["cos_angle"] - 0.1) > 1e-4:               # the rule itself, away from ties
            assert g["accept"] == (not (post <= pre or post <= lcp or post <= pri or want["cos_angle"] < 0.1)), i
    B.close()


def test_batch_tracker_score_block_queued_behind_align(hiplib, oracle):
    """local_tracker.cpp:228-251 for a batch: align, then compute_innerproduct with tran = the alignment's own result and the
    ell it left behind (Q1), inliers counted from 0.  The score launch is queued behind the align launch (transforms read from
    the device-resident states) and collected afterwards; a second round over the same batch object (descriptor cache hit,
    warm-started states, carried ell) must match the oracle objects driven twice as well."""
    from cvo_slam_amd import synth
    sizes = (300, 520, 64, 900, 410, 777, 333, 640, 250, 1000, 1500)           # 55 requests: descriptors travel through HBM
    pairs = [synth.make_small_pair(900 + i, n=n) for i, n in enumerate(sizes)]
    n = len(pairs)
    B = hiplib.CvoBatch(n)
    single = []
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
        o = oracle.OracleCvo(); o.set_pcd(p.fixed.xyz, p.fixed.feat); o.set_pcd(p.moving.xyz, p.moving.feat)
        single.append(o)
    for rnd in range(2):
        B.align_async(n)
        B.enqueue_innerproduct(n)                                               # no host wait in between
        res = B.wait(n)
        got = B.innerproduct_results(n)
        assert len(got) == n
        for i, (o, r, g) in enumerate(zip(single, res, got)):
            rc, _ = o.align(); assert rc == 0
            st = o.get_state()
            assert_pose_close(r["transform"], st["transform"])
            rc, want = o.compute_innerproduct(st["transform"]); assert rc == 0
            for key in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd"):
                assert g[key][1] == want[key][1], (rnd, i, key)
                assert g[key][0] == pytest.approx(want[key][0], rel=1e-5), (rnd, i, key)
            assert g["inliers"] == want["inliers"], (rnd, i)
            assert g["cos_angle"] == pytest.approx(want["cos_angle"], rel=1e-5)
            np.testing.assert_allclose(g["post_hessian"], want["post_hessian"], rtol=1e-3, atol=1e-3 * np.abs(want["post_hessian"]).max())
    with pytest.raises(hiplib.CvoError):
        B.innerproduct_results(n)                                               # nothing queued any more
    B.close()


def test_handles_are_independent_across_host_threads(hiplib, oracle):
    """keyframe_graph.cpp:212-240: the optional back-end thread owns its own cvo objects while the tracker thread uses its.
    Handles share nothing (own stream, own buffers, thread-local error text): four host threads, each with its own object and
    pair, must all get their own oracle's answer."""
    import threading
    from cvo_slam_amd import synth
    pairs = [synth.make_small_pair(500 + i, n=400 + 150 * i) for i in range(4)]
    want = []
    for p in pairs:
        o, _ = oracle_align(oracle, (p.fixed.xyz, p.fixed.feat), (p.moving.xyz, p.moving.feat))
        want.append(o.get_state())
    got, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):                                              # a few rounds each, interleaving on the GPU
                g, _ = gpu_align(hiplib, (pairs[i].fixed.xyz, pairs[i].fixed.feat), (pairs[i].moving.xyz, pairs[i].moving.feat), wgs=[1, 2, 4, 0][i])
                got[i] = (g.transform.copy(), g.get_iteration_number(), g.get_A_nonzero())
                g.compute_innerproduct(g.transform)
                g.close()
        except Exception as e:                                              # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for i in range(4):
        re, te = rot_trans_err(got[i][0], want[i]["transform"])
        assert re <= 1e-6 and te <= 1e-6 and got[i][1] == want[i]["iter"] and got[i][2] == want[i]["A_nonzero"], i


def test_batch_results_to_device_records(hiplib):
    import torch
    from cvo_slam_amd import shard, synth
    pairs = [synth.make_small_pair(300 + i, n=200) for i in range(3)]
    B = hiplib.CvoBatch(3)
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    B.align_async(3)
    out = torch.zeros((3, shard.RESULT_FLOATS), dtype=torch.float32, device="cuda")
    B.results_to_device(out.data_ptr(), 3)
    res = B.wait(3)
    torch.cuda.synchronize()
    rec = out.cpu().numpy()
    for i, r in enumerate(res):
        np.testing.assert_array_equal(rec[i, :12].reshape(3, 4), r["transform"])
        assert (int(rec[i, 12]), int(rec[i, 13]), int(rec[i, 14]), int(rec[i, 15])) == (r["iter"], r["A_nonzero"], r["iterations_run"], r["status"])
    table = shard.gather_results(out, 3, 1)
    assert table.shape == (3, shard.RESULT_FLOATS)
    B.close()


def test_randomized_sweep_against_the_oracle(hiplib):
    """scripts/gpu_stress.py: random small clouds (64 ... 3072 points, ragged sizes), workgroup counts 1 ... 32, cull tiles,
    LDS layouts, list skins and capacities -- pose, iteration count, nonzeros of every iteration and the score block of each
    case against the oracle.  40 cases here; the round ran ~1000 with other seeds."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CASES="40", SEED="77")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_stress.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "40 / 40 cases identical to the oracle" in r.stdout


def test_dynamic_pair_queue_gives_the_same_results(hiplib):
    """A launch capped to fewer pair slots than pairs (cvo_batch_set_max_workgroups) hands the pairs to its slots through the
    in-kernel queue; with one and with several cooperating workgroups per pair, repeated launches: the same bits as a slot per pair."""
    from cvo_slam_amd import synth
    pairs = [synth.make_small_pair(900 + i, n=250 + 60 * (i % 4)) for i in range(13)]
    B = hiplib.CvoBatch(len(pairs))
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    B.set_workgroups(1)
    want = B.align(len(pairs))
    for wgs, cap in ((1, 3), (1, 1), (2, 4), (2, 6), (4, 8)):
        B.set_workgroups(wgs); B.set_max_workgroups(cap)
        for _ in range(2):
            B.reset_states()
            got = B.align(len(pairs))
            for a, b in zip(want, got):
                assert b["status"] == 0
                np.testing.assert_array_equal(a["transform"], b["transform"])
                assert (a["iter"], a["A_nonzero"], a["iterations_run"]) == (b["iter"], b["A_nonzero"], b["iterations_run"])
    B.close()
