"""Adoption (cvo_batch_set_adoption): finished workgroups help with the pairs of their launch that still run.  The results must be
those of the same batch without it, and it must actually happen when alignments of very different lengths share a launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(hiplib, pairs, adoption):
    ca = hiplib
    b = ca.CvoBatch(len(pairs))
    b.set_workgroups(1)
    b.set_adoption(adoption)
    for i, p in enumerate(pairs):
        b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    return b


def test_adoption_happens_and_changes_no_result(hiplib):
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(i) for i in range(24)]                   # 33 ... 92 iterations: the short ones finish with the long ones half way
    ref = _batch(hiplib, pairs, False)
    ref.align_async(len(pairs)); want = ref.wait(len(pairs))
    assert ref.last_adoptions() == 0
    b = _batch(hiplib, pairs, True)
    helped = 0
    for rep in range(8):                                            # every launch: same inputs, same results, whoever helped whom (pairs grow to four workgroups)
        b.reset_states(); b.align_async(len(pairs)); got = b.wait(len(pairs))
        helped += b.last_adoptions()
        for w, g in zip(want, got):
            assert g["status"] == 0 and g["iter"] == w["iter"] and g["iterations_run"] == w["iterations_run"] and g["A_nonzero"] == w["A_nonzero"]
            assert np.array_equal(g["transform"], w["transform"])
    assert helped >= 8, helped
    ref.close(); b.close()


def test_adoption_with_several_launches_in_flight(hiplib):
    """Four batch objects on their own streams, launches overlapping: helpers only offer themselves once nothing is queued on the
    device, and every launch still returns the results of the batch without adoption."""
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(100 + i) for i in range(32)]
    ref = _batch(hiplib, pairs, False)
    ref.align_async(len(pairs)); want = ref.wait(len(pairs)); ref.close()
    bs = [_batch(hiplib, pairs, True) for _ in range(4)]
    helped = 0
    for rnd in range(3):
        for b in bs:
            b.reset_states(); b.align_async(len(pairs))
        for b in bs:
            got = b.wait(len(pairs)); helped += b.last_adoptions()
            for w, g in zip(want, got):
                assert g["status"] == 0 and g["iter"] == w["iter"] and g["A_nonzero"] == w["A_nonzero"]
                assert np.array_equal(g["transform"], w["transform"])
    assert helped >= 1, helped
    for b in bs: b.close()


@pytest.mark.parametrize("knobs", [dict(CVO_HIP_Y_MODE="2"), dict(CVO_HIP_RESORT="2"), dict(CVO_HIP_ADOPT_KMAX="1000"), dict(CVO_HIP_ADOPT_KMAX="0"), dict(CVO_HIP_ADOPT_DWELL_US="50")])
def test_adoption_under_other_layouts_and_cutoffs(hiplib, knobs):
    """Member regions and re-dealt rows under the 12-byte plane layout (the rebuild scratch lives inside the cull tile there), with
    rows re-sorted at every list refinement, with helpers joining at any iteration, with no pair ever eligible, and with helpers that
    wait for the device's queue to have been dry for a while before they offer themselves: the bits of the same batch without adoption."""
    import os
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(200 + i) for i in range(16)]
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        ref = _batch(hiplib, pairs, False)
        ref.align_async(len(pairs)); want = ref.wait(len(pairs)); ref.close()
        b = _batch(hiplib, pairs, True)
        helped = 0
        for rep in range(4):
            b.reset_states(); b.align_async(len(pairs)); got = b.wait(len(pairs)); helped += b.last_adoptions()
            for w, g in zip(want, got):
                assert g["status"] == 0 and g["iter"] == w["iter"] and g["A_nonzero"] == w["A_nonzero"]
                assert np.array_equal(g["transform"], w["transform"])
        assert (helped > 0) == (knobs.get("CVO_HIP_ADOPT_KMAX") != "0"), helped
        b.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def test_a_helper_that_leaves_after_the_acceptance_cannot_fail_the_pair(hiplib):
    """Injection (CVO_HIP_ADOPT_INJECT=1, read when the handle is made): every helper whose offer an owner has accepted walks away instead of
    confirming.  The owner must take the acceptance back (cvo_batch_last_adoption_retractions counts them) and carry on with the workgroups it
    has: status 0 and the bits of the batch without adoption on 24 full-size pairs -- not CVO_ERR_TIMEOUT after three seconds."""
    import os
    import time
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(i) for i in range(24)]
    ref = _batch(hiplib, pairs, False)
    ref.align_async(len(pairs)); want = ref.wait(len(pairs)); ref.close()
    old = os.environ.get("CVO_HIP_ADOPT_INJECT")
    os.environ["CVO_HIP_ADOPT_INJECT"] = "1"
    try:
        b = _batch(hiplib, pairs, True)
    finally:
        if old is None: os.environ.pop("CVO_HIP_ADOPT_INJECT", None)
        else: os.environ["CVO_HIP_ADOPT_INJECT"] = old
    retracted = 0
    t0 = time.perf_counter()
    for rep in range(4):
        b.reset_states(); b.align_async(len(pairs)); got = b.wait(len(pairs))
        assert b.last_adoptions() == 0                               # nobody ever joined ...
        retracted += b.last_adoption_retractions()                   # ... although offers were accepted
        for w, g in zip(want, got):
            assert g["status"] == 0 and g["iter"] == w["iter"] and g["iterations_run"] == w["iterations_run"] and g["A_nonzero"] == w["A_nonzero"]
            assert np.array_equal(g["transform"], w["transform"])
    assert retracted >= 4, retracted
    assert time.perf_counter() - t0 < 2.0                            # no wait ran into a timeout
    b.close()
