"""Sequence replay end to end: a synthetic RGB-D sequence is written to disk in the TUM layout (PNGs +
association file + calibration yaml), replayed frame to frame through the C ABI (point clouds generated on the GPU from the
images), and the trajectory file is compared with the known camera motion."""
import os

import numpy as np
import pytest

from helpers import rot_trans_err

pytestmark = pytest.mark.gpu


def test_replay_synthetic_sequence_from_disk(hiplib, tmp_path):
    from cvo_slam_amd import replay, synth
    frames, poses_true = synth.make_sequence(1, n_frames=5)
    os.makedirs(tmp_path / "rgb"); os.makedirs(tmp_path / "depth")
    lines = []
    for k, (bgr, dep) in enumerate(frames):
        ts = f"{1000.0 + k / 30.0:.6f}"
        replay.write_png(str(tmp_path / "rgb" / f"{ts}.png"), bgr[..., ::-1])          # files hold R,G,B; cv::imread hands back B,G,R
        replay.write_png(str(tmp_path / "depth" / f"{ts}.png"), dep)
        lines.append(f"{ts} rgb/{ts}.png {ts} depth/{ts}.png")
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    cam = synth.TUM1
    (tmp_path / "cam.yaml").write_text(f"%YAML:1.0\nCamera.fx: {cam['fx']}\nCamera.fy: {cam['fy']}\nCamera.cx: {cam['cx']}\nCamera.cy: {cam['cy']}\n"
                                       f"DepthMapFactor: {cam['depth_factor']}\n")
    # the loader returns exactly the rendered images
    b0, d0 = replay.load_frame(str(tmp_path), lines[0].split()[1], lines[0].split()[3])
    np.testing.assert_array_equal(b0, frames[0][0]); np.testing.assert_array_equal(d0, frames[0][1])
    out = str(tmp_path / "traj.txt")
    poses, info = replay.replay_sequence(str(tmp_path), str(tmp_path / "assoc.txt"), str(tmp_path / "cam.yaml"), out)
    assert len(poses) == 5 and all(i["points"] > 2000 for i in info)
    rows = [l.split() for l in open(out).read().splitlines()]
    assert [r[0] for r in rows] == [l.split()[0] for l in lines] and all(len(r) == 8 for r in rows)
    for k in range(1, 5):
        re, te = rot_trans_err(poses[k][:3, :], poses_true[k][:3, :])
        assert re < 6e-3 and te < 2.5e-2, (k, re, te)                                   # a few millimetres / tenths of a degree per step, chained
        np.testing.assert_allclose([float(v) for v in rows[k][1:4]], poses[k][:3, 3], rtol=1e-6, atol=1e-8)
        q = replay.rotation_to_quaternion(poses[k][:3, :3])
        np.testing.assert_allclose([float(v) for v in rows[k][4:]], q, atol=1e-7)
    # the same frames straight from memory give the same poses (the disk formats are lossless)
    poses_mem, _ = replay.replay_odometry(frames, synth.camera_tuple(cam))
    for a, b in zip(poses, poses_mem):
        np.testing.assert_array_equal(a, b)


def test_full_size_tracker_sequence_from_images_vs_oracle(hiplib, oracle):
    """SURVEY 8d's second scenario at full size, from the images: the LocalTracker call sequence (local_tracker.cpp:228-251,
    356-431, 506) on 640x480 frames -- point clouds from the GPU generator, odometry object with warm start and carried ell
    (Q1, Q2), keyframe object warm-started by reset_initial -- against the oracle fed with the oracle generator's clouds of
    the same frames.  Every transform within the tolerance, every iteration count equal."""
    from cvo_slam_amd import synth
    frames, _ = synth.make_sequence(2, n_frames=5)
    cam = synth.camera_tuple(synth.TUM1)

    odo, kf = hiplib.Cvo(), hiplib.Cvo()
    got, scores = [], []
    odo.set_pcd_images(*frames[0], cam); kf.set_pcd_images(*frames[0], cam)          # :228, :231
    t = odo.match_odometry_images(*frames[1], cam); got.append((t, odo.get_iteration_number()))   # :233
    scores.append(odo.compute_innerproduct(np.asarray(t, np.float32)))               # :251
    odo.update_fixed_pcd()                                                           # :277
    kf.first_frame = False; kf.reset_transform(np.asarray(t, np.float32))            # :330-333
    for f in frames[2:]:
        t = odo.match_odometry_images(*f, cam); got.append((t, odo.get_iteration_number()))       # :356
        scores.append(odo.compute_innerproduct(np.asarray(t, np.float32)))           # :375
        odo.update_fixed_pcd()                                                       # :403
        kf.reset_initial(np.asarray(t, np.float32))                                  # :407
        tk = kf.match_keyframe_images(*f, cam); got.append((tk, kf.get_iteration_number()))       # :415
        scores.append(kf.compute_innerproduct(np.asarray(tk, np.float32)))           # :431
        kf.update_previous_pcd()                                                     # :506
    n_pts = odo.get_fixed_and_moving_number()
    # the economies of the tracker's pattern were taken: the keyframe object generated none of its clouds (frames 0, 2, 3, 4 came from the odometry object's generation) and both
    # objects' score blocks were started by their alignments from the second one on
    assert kf.shared_cloud_count() == 4 and odo.shared_cloud_count() == 0
    assert odo.queued_score_count() == 3 and kf.queued_score_count() == 2
    odo.close(); kf.close()

    clouds = [oracle.pcd_generate(b, d, cam) for (b, d) in frames]
    assert (clouds[-2]["n"], clouds[-1]["n"]) == tuple(n_pts) and min(n_pts) > 2000   # counts as of the last set_pcd (cvo.cpp:370-371)
    oo, ok = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8), oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
    want = []
    c = clouds[0]; oo.set_pcd(c["xyz"], c["feat"]); ok.set_pcd(c["xyz"], c["feat"])
    want_scores = []
    c = clouds[1]; rc, t = oo.match(c["xyz"], c["feat"]); assert rc == 0; want.append((t, oo.get_state()["iter"]))
    rc, sc = oo.compute_innerproduct(oo.get_state()["transform"]); assert rc == 0; want_scores.append(sc)
    oo.update_fixed_pcd(); ok.reset_transform(t.astype(np.float32))
    for c in clouds[2:]:
        rc, t = oo.match(c["xyz"], c["feat"]); assert rc == 0; want.append((t, oo.get_state()["iter"]))
        rc, sc = oo.compute_innerproduct(oo.get_state()["transform"]); assert rc == 0; want_scores.append(sc)
        oo.update_fixed_pcd()
        ok.reset_initial(t.astype(np.float32))
        rc, tk = ok.match(c["xyz"], c["feat"]); assert rc == 0; want.append((tk, ok.get_state()["iter"]))
        rc, sc = ok.compute_innerproduct(ok.get_state()["transform"]); assert rc == 0; want_scores.append(sc)
        ok.update_previous_pcd()
    assert len(got) == len(want) == 7 and len(scores) == len(want_scores) == 7
    for k, (g, w) in enumerate(zip(scores, want_scores)):
        for key in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd"):
            assert g[key][1] == w[key][1], (k, key)
            assert g[key][0] == pytest.approx(w[key][0], rel=1e-5), (k, key)
        assert g["inliers"] == w["inliers"], k
    for k, ((tg, ig), (tw, iw)) in enumerate(zip(got, want)):
        re, te = rot_trans_err(tg, tw)
        assert re <= 1e-4 and te <= 1e-4, (k, re, te)
        assert ig == iw, (k, ig, iw)


def test_full_size_loop_closure_candidates_vs_oracle(hiplib, oracle):
    """SURVEY 8d's loop-closure-like set at full size: motions up to 10 degrees / 15 cm, every candidate warm-started within
    ~2 degrees / 3 cm of the truth through reset_initial (keyframe_graph.cpp:693-705), all aligned by one batch launch and
    scored by one launch; poses, iteration counts, pair counts and the accept decision against single oracle objects."""
    from cvo_slam_amd import synth
    from helpers import make_tf
    rng = np.random.default_rng(12)
    n = 6
    pairs = [synth.make_pair(300 + i, max_deg=10.0, max_trans=0.15) for i in range(n)]
    lc_priors = []
    for p in pairs:                                                  # prior = truth perturbed by <= 2 deg / 3 cm
        d = np.eye(4); d[:3] = make_tf(rng.normal(size=3), np.deg2rad(rng.uniform(0.5, 2.0)), rng.normal(size=3) * 0.012)
        T = np.eye(4); T[:3] = p.true_transform
        lc_priors.append((T @ d)[:3].astype(np.float32))
    lc_priors = np.stack(lc_priors)
    priors = np.stack([np.eye(3, 4, dtype=np.float32)] * n)
    B = hiplib.CvoBatch(n)
    single = []
    for i, p in enumerate(pairs):
        B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
        o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
        o.reset_initial(lc_priors[i])
        o.set_pcd(p.fixed.xyz, p.fixed.feat); o.set_pcd(p.moving.xyz, p.moving.feat)
        st0 = o.get_state(); B.set_state(i, st0["R"], st0["T"], st0["ell"])
        single.append(o)
    res = B.align(n)
    got = B.compute_innerproduct_lc(priors, lc_priors, lc_priors)
    accepted = 0
    for i, (o, r, g, p) in enumerate(zip(single, res, got, pairs)):
        rc, _ = o.align(); assert rc == 0
        st = o.get_state()
        re, te = rot_trans_err(r["transform"], st["transform"])
        assert re <= 1e-4 and te <= 1e-4, (i, re, te)
        assert r["iter"] == st["iter"], (i, r["iter"], st["iter"])
        rc, want = o.compute_innerproduct_lc(priors[i], lc_priors[i], lc_priors[i], st["transform"]); assert rc == 0
        for key in ("inn_prior", "inn_lc_prior", "inn_lc_pre", "inn_lc_post", "inn_fixed_pcd", "inn_moving_pcd"):
            assert g[key][1] == want[key][1], (i, key)
            assert g[key][0] == pytest.approx(want[key][0], rel=1e-5), (i, key)
        assert (g["inliers_svd"], g["inliers_pnpransac"]) == (want["inliers_svd"], want["inliers_pnpransac"])
        re_t, te_t = rot_trans_err(r["transform"], p.true_transform)
        assert re_t < 5e-3 and te_t < 1e-2, (i, re_t, te_t)           # the large motion is recovered from the prior
        accepted += int(g["accept"])
    assert accepted >= n - 1                                         # verified candidates pass the reference's rule (keyframe_graph.cpp:711-712)
    B.close()
