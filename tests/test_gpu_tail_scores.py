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
