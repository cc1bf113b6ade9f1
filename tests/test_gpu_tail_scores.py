"""cvo_batch_set_tail_scores: the tracker's score block (cvo::compute_innerproduct, cvo.cpp:475-503) answered by the align launch in the
tail of every pair's workgroup -- against the oracle, and against the score-kernel path it replaces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(got, want, rel):
    for key in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd"):
        assert got[key][1] == want[key][1], key
        assert got[key][0] == pytest.approx(want[key][0], rel=rel), key
    assert got["inliers"] == want["inliers"]
    assert got["cos_angle"] == pytest.approx(want["cos_angle"], rel=rel)
    np.testing.assert_allclose(got["post_hessian"], want["post_hessian"], rtol=1e-3, atol=1e-3 * np.abs(want["post_hessian"]).max())


def test_tail_scores_match_the_oracle_and_the_score_kernel(hiplib, oracle):
    from cvo_slam_amd import synth
    sizes = (300, 520, 64, 900, 410, 777, 1500, 2048)
    pairs = [synth.make_small_pair(1300 + i, n=n) for i, n in enumerate(sizes)] + [synth.make_pair(3)]       # one full-size pair too
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
    n = len(clouds)
    orcs = []
    for c in clouds:
        o = oracle.OracleCvo(); o.set_pcd(c[0], c[1]); o.set_pcd(c[2], c[3]); orcs.append(o)
    ref = hiplib.CvoBatch(n); ref.set_workgroups(1); ref.set_pairs(clouds)           # the path it replaces: a score launch behind the align launch
    B = hiplib.CvoBatch(n); B.set_workgroups(1); B.set_pairs(clouds); B.set_tail_scores(True)
    for rnd in range(3):                                                             # second round: warm-started states, carried ell, self products cached
        ref.align_async(n); ref.enqueue_innerproduct(n); ref.wait(n); want_gpu = ref.innerproduct_results(n)
        B.align_async(n); res = B.wait(n)
        masks = B.last_tail_answers(n)
        got = B.innerproduct_results(n)
        for i in range(n):
            assert res[i]["status"] == 0
            assert masks[i] & 0b10011 == 0b10011, (rnd, i, masks[i])                 # inn_pre, inn_post and the Hessian came from the workgroup itself
            if rnd > 0:
                assert masks[i] == 31, (rnd, i, masks[i])                            # ... and from the second round on the self products from the clouds' tables
            if rnd < 2:
                rc, _ = orcs[i].align(); assert rc == 0
                rc, want = orcs[i].compute_innerproduct(orcs[i].get_state()["transform"]); assert rc == 0
                _check(got[i], want, 1e-5)
            _check(got[i], want_gpu[i], 1e-6)
    ref.close(); B.close()


def test_tail_scores_fall_back_when_a_workgroup_cannot_answer(hiplib, oracle):
    """Pairs run by several workgroups (their partial sums go through the pair's exchange area), and a launch cut short by max_iter right
    after an ell change (the lists are for the old ell: the host's score kernel answers): the numbers are the same."""
    from cvo_slam_amd import synth
    pairs = [synth.make_small_pair(1400 + i, n=700) for i in range(4)]
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
    for wgs, max_iter in ((2, 2000), (1, 4), (1, 11)):
        prm = hiplib.default_params(); prm.max_iter = max_iter
        ref = hiplib.CvoBatch(len(clouds), prm); ref.set_workgroups(wgs); ref.set_pairs(clouds)
        B = hiplib.CvoBatch(len(clouds), prm); B.set_workgroups(wgs); B.set_pairs(clouds); B.set_tail_scores(True)
        ref.align_async(len(clouds)); ref.enqueue_innerproduct(len(clouds)); ref.wait(); want = ref.innerproduct_results(len(clouds))
        B.align_async(len(clouds)); B.wait(); masks = B.last_tail_answers(len(clouds)); got = B.innerproduct_results(len(clouds))
        if wgs > 1:
            assert all(m & 0b10011 == 0b10011 for m in masks), masks                  # the pair's workgroups add their sums up through the exchange area
        for g, w in zip(got, want):
            _check(g, w, 1e-6)
        ref.close(); B.close()


def test_tail_scores_under_adoption(hiplib):
    """Helped pairs (cvo_batch_set_adoption) answer their score block too: every member holds the lists of its own rows."""
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(i) for i in range(24)]
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
    ref = hiplib.CvoBatch(len(clouds)); ref.set_workgroups(1); ref.set_pairs(clouds)
    ref.align_async(len(clouds)); ref.enqueue_innerproduct(len(clouds)); ref.wait(); want = ref.innerproduct_results(len(clouds)); ref.close()
    B = hiplib.CvoBatch(len(clouds)); B.set_workgroups(1); B.set_adoption(True); B.set_pairs(clouds); B.set_tail_scores(True)
    helped = 0
    for rnd in range(4):
        B.reset_states(); B.align_async(len(clouds)); B.wait(); helped += B.last_adoptions()
        masks = B.last_tail_answers(len(clouds)); got = B.innerproduct_results(len(clouds))
        assert all(m & 0b10011 == 0b10011 for m in masks), masks
        for g, w in zip(got, want):
            _check(g, w, 1e-6)
    assert helped > 0
    B.close()


def test_members_with_different_list_margins_agree_in_the_tail(hiplib, oracle):
    """Two workgroups per pair; the fixed cloud alternates, in blocks of 128 rows (the unit rows are dealt in), between a dense patch a few
    decimetres from the camera and a sparse far wall: member 0 owns the dense blocks, its lists overflow the launch's margin and it rebuilds
    them with the narrow one, member 1 never does.  Their lists go stale at different iterations, so after the final transform one member's
    may be valid and the other's not -- and the tail exchanges partial sums: both must take the same branch (round 3: mismatched exchanges,
    status 6 after three seconds).  Scores against the oracle, poses against the oracle, with and without the tail."""
    rng = np.random.default_rng(11)
    nblk, per = 24, 128
    xyz = np.zeros((nblk * per, 3), np.float32); feat = np.zeros((5, nblk * per), np.float32)
    for b in range(nblk):
        sl = slice(b * per, (b + 1) * per)
        if b % 2 == 0:                                               # dense patch: 1536 points on 0.38 x 0.285 m at 0.6 m: ~545 list candidates per row at the launch margin of 0.35 (the records hold 512 per row), ~483 at the narrow 0.25
            uv = rng.uniform(-1, 1, size=(per, 2)) * np.array([0.19, 0.1425])
            z = 0.6 + 0.02 * np.sin(9 * uv[:, 0]) * np.cos(7 * uv[:, 1])
        else:                                                        # far wall: 1536 points on 3 x 2.2 m at 3 m
            uv = rng.uniform(-1, 1, size=(per, 2)) * np.array([1.5, 1.1])
            z = 3.0 + 0.05 * np.sin(3 * uv[:, 0])
        xyz[sl] = np.stack([uv[:, 0], uv[:, 1], z], axis=1)
        feat[:, sl] = np.stack([128 + 80 * np.sin(11 * uv[:, 0]), 128 + 70 * np.cos(13 * uv[:, 1]), 128 + 60 * np.sin(9 * (uv[:, 0] + uv[:, 1])),
                                15 * np.cos(17 * uv[:, 0]), 15 * np.sin(19 * uv[:, 1])], axis=0)
    from helpers import make_tf, rot_trans_err
    tf = make_tf([0.2, 1.0, 0.1], 0.012, [0.006, -0.004, 0.003])
    mov = ((xyz - tf[:, 3]) @ tf[:, :3] + rng.normal(0, 5e-4, size=xyz.shape)).astype(np.float32)
    mfeat = (feat + rng.normal(0, 1.0, size=feat.shape)).astype(np.float32)
    cloud = (xyz, feat, mov, mfeat)
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
    o.set_pcd(xyz, feat); o.set_pcd(mov, mfeat); rc, _ = o.align(); assert rc == 0
    ost = o.get_state()
    rc, want = o.compute_innerproduct(ost["transform"]); assert rc == 0
    for wgs in (2, 4, 1):
        B = hiplib.CvoBatch(1); B.set_workgroups(wgs); B.set_pairs([cloud]); B.set_tail_scores(True)
        B.align_async(1); res = B.wait(1)
        assert res[0]["status"] == 0, (wgs, res[0])
        re, te = rot_trans_err(res[0]["transform"], ost["transform"])
        assert re <= 1e-6 and te <= 1e-6 and res[0]["iter"] == ost["iter"] and res[0]["A_nonzero"] == ost["A_nonzero"], (wgs, re, te)
        got = B.innerproduct_results(1)[0]
        _check(got, want, 1e-5)
        B.close()
    # The alignment cut short at every iteration of its first phases: the clouds have moved 25 ... 40 mm by then, more than the narrow margin
    # allows (0.25 r_c = 25 mm at ell = 0.15) and less than the launch's (35 mm) -- at some of these cuts member 0's lists are stale for the
    # final transform and member 1's are not.  Against the score-kernel path of the same launch without the tail.
    import time
    t0 = time.perf_counter()
    partly = 0
    for max_iter in range(2, 16):
        prm = hiplib.default_params(); prm.max_iter = max_iter
        ref = hiplib.CvoBatch(1, prm); ref.set_workgroups(2); ref.set_pairs([cloud])
        B = hiplib.CvoBatch(1, prm); B.set_workgroups(2); B.set_pairs([cloud]); B.set_tail_scores(True)
        ref.align_async(1); ref.enqueue_innerproduct(1); rres = ref.wait(1); wantg = ref.innerproduct_results(1)[0]
        B.align_async(1); res = B.wait(1); mask = B.last_tail_answers(1)[0]; got = B.innerproduct_results(1)[0]
        assert res[0]["status"] == 0 and rres[0]["status"] == 0, (max_iter, res[0]["status"])
        assert np.array_equal(res[0]["transform"], rres[0]["transform"]), max_iter
        _check(got, wantg, 1e-6)
        partly += int(mask & 0b10010 != 0b10010)                      # inn_post / Hessian left to the host: somebody's lists were stale
        ref.close(); B.close()
    assert time.perf_counter() - t0 < 3.0                              # nobody waited for an exchange that never came (3 s each)
    assert partly >= 1, partly


def test_single_handle_compute_innerproduct_answered_by_the_align_launch(hiplib, oracle):
    """cvo_set_tail_scores: match_* + compute_innerproduct(tran = the result), the tracker's sequence (local_tracker.cpp:356-375), on ONE handle -- the score block is
    started by the alignment itself ("queue": the score kernel queued behind the align kernel, the default; "kernel": the align launch's own tail); against the
    score-kernel path (off) and the oracle, for several workgroup counts; another transform than the result must not be answered from what was started, nor may
    answers survive a change of the clouds."""
    import os
    from cvo_slam_amd import synth
    for seed, n, wgs in ((41, 700, 1), (42, 1500, 2), (43, 2048, 0), (3, 0, 0), (3, 0, 4)):
        p = synth.make_pair(seed) if n == 0 else synth.make_small_pair(1700 + seed, n=n)
        o = oracle.OracleCvo(); o.set_pcd(p.fixed.xyz, p.fixed.feat); rc, _ = o.match(p.moving.xyz, p.moving.feat); assert rc == 0
        rc, want = o.compute_innerproduct(o.get_state()["transform"]); assert rc == 0
        got = {}
        for tail in ("queue", "kernel", None):
            g = hiplib.Cvo(); g.set_workgroups(wgs)
            if tail:
                os.environ["CVO_HIP_HANDLE_TAIL"] = tail
            try:
                g.set_tail_scores(tail is not None)
            finally:
                os.environ.pop("CVO_HIP_HANDLE_TAIL", None)
            g.set_pcd(p.fixed.xyz, p.fixed.feat)
            tf = g.match_keyframe(p.moving.xyz, p.moving.feat)
            if tail == "queue" and seed == 42:      # asked for another transform FIRST: what was queued is dropped, the score kernel answers
                other = np.array(g.transform, np.float32).copy(); other[0, 3] += 0.01
                first = g.compute_innerproduct(other)
            got[tail] = g.compute_innerproduct(g.transform)
            if tail == "queue" and seed == 42:
                assert first["inn_post"][0] != got[tail]["inn_post"][0]
            if tail:      # not the align's own transform: the score kernel answers, and differently
                other = np.array(g.transform, np.float32).copy(); other[0, 3] += 0.01
                sc2 = g.compute_innerproduct(other)
                assert sc2["inn_post"][0] != got[tail]["inn_post"][0]
                again = g.compute_innerproduct(g.transform)
                assert again["inn_post"] == got[tail]["inn_post"] and again["inliers"] == got[tail]["inliers"]
                # a new MOVING cloud between the alignment and the question: nothing started for the old one may answer
                q = synth.make_small_pair(900 + seed, n=600)
                g.match_keyframe(p.moving.xyz, p.moving.feat)
                tf_old = np.array(g.transform, np.float32).copy()
                g.set_pcd(q.moving.xyz, q.moving.feat)
                after = g.compute_innerproduct(tf_old)
                ref = hiplib.Cvo(); ref.set_workgroups(wgs); ref.set_pcd(p.fixed.xyz, p.fixed.feat); ref.match_keyframe(p.moving.xyz, p.moving.feat); ref.match_keyframe(p.moving.xyz, p.moving.feat)
                np.testing.assert_array_equal(np.array(ref.transform, np.float32), np.array(g.transform, np.float32))
                ref.set_pcd(q.moving.xyz, q.moving.feat)
                want_after = ref.compute_innerproduct(tf_old)
                assert after["inn_post"] == want_after["inn_post"] and after["inliers"] == want_after["inliers"] and after["inn_pre"] == want_after["inn_pre"]
                ref.close()
            g.close()
        for tail in ("queue", "kernel"):
            _check(got[tail], got[None], 1e-6)
            _check(got[tail], want, 1e-5)


def test_a_handle_learns_the_trackers_pattern(hiplib):
    """cvo_set_tail_scores' default (2): nothing is queued until an alignment has been followed by compute_innerproduct(its transform); from then on every alignment
    queues the block, until one is not asked.  A tracker's object (asked after every frame) against the same sequence with the mode off: identical numbers; a
    loop-closure object (compute_innerproduct_lc) never queues."""
    from cvo_slam_amd import synth
    p = synth.make_small_pair(2100, n=1500); q = synth.make_small_pair(2101, n=1500)
    keys = ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd", "inliers", "cos_angle")
    auto, off = hiplib.Cvo(), hiplib.Cvo(); off.set_tail_scores(0)
    seen = []
    for g in (auto, off):
        g.set_pcd(p.fixed.xyz, p.fixed.feat)
        out = []
        for frame in range(5):
            mv = (p if frame % 2 == 0 else q).moving
            g.match_odometry(mv.xyz, mv.feat)
            if frame != 3:                                   # frame 3 is not asked: frame 4 queues nothing, frame 4's question teaches it again
                sc = g.compute_innerproduct(np.asarray(g.transform, np.float32))
                out.append([sc[k] for k in keys] + [sc["post_hessian"].tolist()])
        seen.append(out)
    assert seen[0] == seen[1]
    assert off.queued_score_count() == 0
    assert auto.queued_score_count() == 2                    # frames 1 and 2 (frame 0: nothing learned yet; frame 3 queued but was not asked; frame 4: not queued)
    auto.close(); off.close()
    lc = hiplib.Cvo(); lc.set_pcd(p.fixed.xyz, p.fixed.feat)
    eye = np.eye(3, 4, dtype=np.float32)
    for frame in range(3):
        lc.match_keyframe(p.moving.xyz, p.moving.feat); lc.compute_innerproduct_lc(eye, eye, eye, np.asarray(lc.transform, np.float32))
    assert lc.queued_score_count() == 0
    lc.close()
