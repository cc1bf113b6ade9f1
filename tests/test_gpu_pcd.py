"""GPU point-cloud generator (cvo_set_pcd_images) against the oracle restatement of the reference's
pcd_generator: the selected pixels, positions and features must be IDENTICAL (integer / per-pixel float work), and the
tracker driven from images must land on the oracle's poses."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import rot_trans_err

pytestmark = pytest.mark.gpu


def frames(index, cam):
    from cvo_slam_amd import synth
    (fa, da), (fb, db), tf = synth.make_frames(index, cam=cam)
    return (fa, da), (fb, db), synth.camera_tuple(cam), tf


def check_cloud(hiplib, g, slot, want):
    xyz, feat = g.get_cloud(slot)
    px = g.get_selected_points(slot)
    assert xyz.shape[0] == want["n"]
    np.testing.assert_array_equal(px, want["px"])
    np.testing.assert_array_equal(xyz, want["xyz"])
    np.testing.assert_array_equal(feat, want["feat"])


@pytest.mark.parametrize("index,shape,num_want", [(0, "tum", 3000), (5, "tum", 3000), (2, "eth3d", 3000), (1, "tum", 300), (4, "tum", 12000), (2, "eth3d", 10000)])
def test_generated_cloud_identical_to_oracle(hiplib, oracle, index, shape, num_want):
    from cvo_slam_amd import synth
    cam = synth.TUM1 if shape == "tum" else synth.ETH3D
    (fa, da), (fb, db), camt, _ = frames(index, cam)
    g = hiplib.Cvo(); g.set_num_want(num_want)
    g.set_pcd_images(fa, da, camt)                                     # first call fills the fixed slot (cvo.cpp:352-360)
    g.set_pcd_images(fb, db, camt)                                     # then the moving one
    check_cloud(hiplib, g, hiplib.api.SLOT_FIXED, oracle.pcd_generate(fa, da, camt, num_want=num_want, cap=40000))
    check_cloud(hiplib, g, hiplib.api.SLOT_MOVING, oracle.pcd_generate(fb, db, camt, num_want=num_want, cap=40000))
    nf, nm = g.get_fixed_and_moving_number()
    assert nf == g.get_cloud(hiplib.api.SLOT_FIXED)[0].shape[0] and nm == g.get_cloud(hiplib.api.SLOT_MOVING)[0].shape[0]
    g.close()


def test_committed_fixture_and_edge_images(hiplib, oracle):
    gd = np.load(os.path.join(GOLDEN, "pcd_frame_small.npz"))
    g = hiplib.Cvo(); g.set_num_want(int(gd["num_want"]))
    g.set_pcd_images(gd["bgr"], gd["depth"], tuple(gd["camera"]))
    xyz, feat = g.get_cloud(hiplib.api.SLOT_FIXED)
    np.testing.assert_array_equal(g.get_selected_points(hiplib.api.SLOT_FIXED), gd["px"])
    np.testing.assert_array_equal(xyz, gd["xyz"]); np.testing.assert_array_equal(feat, gd["feat"])
    g.close()
    # a flat image has no gradient above the thresholds: empty cloud, then align reports the empty cloud (Q8)
    g = hiplib.Cvo()
    flat = np.full((128, 160, 3), 90, np.uint8); dep = np.full((128, 160), 5000, np.uint16)
    g.set_pcd_images(flat, dep, tuple(gd["camera"]))
    assert g.get_cloud(hiplib.api.SLOT_FIXED)[0].shape[0] == 0
    assert oracle.pcd_generate(flat, dep, tuple(gd["camera"]))["n"] == 0
    # all-invalid depth: pixels are selected, none survives the depth test (pcd_generator.cpp:471)
    g2 = hiplib.Cvo()
    g2.set_pcd_images(gd["bgr"], np.zeros_like(gd["depth"]), tuple(gd["camera"]))
    assert g2.get_cloud(hiplib.api.SLOT_FIXED)[0].shape[0] == 0
    with pytest.raises(hiplib.CvoError):
        g2.set_pcd_images(gd["bgr"][:32, :32], gd["depth"][:32, :32], tuple(gd["camera"]))    # smaller than the selector's blocks
    g.close(); g2.close()


def test_tracking_from_images_matches_oracle(hiplib, oracle):
    """cvo.set_pcd(img A); cvo.match_keyframe(img B) on both sides: the clouds are identical, so the alignment must be too."""
    from cvo_slam_amd import synth
    (fa, da), (fb, db), camt, true_tf = frames(6, synth.TUM1)
    g = hiplib.Cvo()
    g.set_pcd_images(fa, da, camt)
    tf_g = g.match_keyframe_images(fb, db, camt)
    ca, cb = oracle.pcd_generate(fa, da, camt), oracle.pcd_generate(fb, db, camt)
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=8)
    o.set_pcd(ca["xyz"], ca["feat"]); o.set_pcd(cb["xyz"], cb["feat"]); rc, _ = o.align(); assert rc == 0
    st = o.get_state()
    re, te = rot_trans_err(tf_g, st["transform"])
    assert re <= 1e-6 and te <= 1e-6
    assert g.get_iteration_number() == st["iter"] and g.get_A_nonzero() == st["A_nonzero"]
    # and the alignment recovers the synthetic motion to a few millimetres / a tenth of a degree
    re, te = rot_trans_err(tf_g, true_tf)
    assert re < 5e-3 and te < 2e-2
    g.close()


def test_two_handles_given_one_frame_generate_its_cloud_once(hiplib, oracle):
    """The tracker's pattern (local_tracker.cpp:356, 415): cvo_odometry and cvo_keyframe are handed the same frame.  The second handle takes a device
    copy of the cloud the first one generated -- identical to the oracle's and to a cloud generated from scratch; a frame that differs in ONE byte (same buffer, changed in
    place), another num_want or another camera is generated; the copies are independent objects (the first handle moves on, the second keeps its cloud)."""
    from cvo_slam_amd import synth
    (fa, da), (fb, db), camt, _ = frames(3, synth.TUM1)
    want_a, want_b = oracle.pcd_generate(fa, da, camt), oracle.pcd_generate(fb, db, camt)
    odo, kf = hiplib.Cvo(), hiplib.Cvo()
    odo.set_pcd_images(fa, da, camt); kf.set_pcd_images(fa, da, camt)                     # FIXED clouds of both objects: one frame
    assert odo.shared_cloud_count() == 0 and kf.shared_cloud_count() == 1
    check_cloud(hiplib, odo, 0, want_a); check_cloud(hiplib, kf, 0, want_a)
    tf_o = odo.match_odometry_images(fb, db, camt); tf_k = kf.match_keyframe_images(fb, db, camt)   # MOVING clouds: one frame again
    assert odo.shared_cloud_count() == 0 and kf.shared_cloud_count() == 2
    check_cloud(hiplib, odo, 1, want_b); check_cloud(hiplib, kf, 1, want_b)
    np.testing.assert_array_equal(tf_o, tf_k)                                               # same clouds, same start: same alignment
    # the first object moves on (update_fixed_pcd, a new frame); the second one's clouds stay what they were
    odo.update_fixed_pcd(); odo.match_odometry_images(fa, da, camt)
    check_cloud(hiplib, kf, 1, want_b); check_cloud(hiplib, kf, 0, want_a); check_cloud(hiplib, odo, 0, want_b); check_cloud(hiplib, odo, 1, want_a)
    # one byte of the frame changed in place: not the same frame
    third = hiplib.Cvo()
    fa2 = fa.copy(); third.set_pcd_images(fa2, da, camt)
    assert third.shared_cloud_count() == 1                                                   # (fa was the thread's last generated frame: odo's second match)
    fa2[-1, -1, 2] ^= 1
    fourth = hiplib.Cvo(); fourth.set_pcd_images(fa2, da, camt)
    assert fourth.shared_cloud_count() == 0
    check_cloud(hiplib, fourth, 0, oracle.pcd_generate(fa2, da, camt))
    da2 = da.copy(); da2[-1, -1] ^= 1
    fifth = hiplib.Cvo(); fifth.set_pcd_images(fa2, da2, camt)
    assert fifth.shared_cloud_count() == 0
    sixth = hiplib.Cvo(); sixth.set_num_want(1000); sixth.set_pcd_images(fa2, da2, camt)
    assert sixth.shared_cloud_count() == 0
    check_cloud(hiplib, sixth, 0, oracle.pcd_generate(fa2, da2, camt, num_want=1000))
    cam2 = list(camt); cam2[1] *= 1.01
    seventh = hiplib.Cvo(); seventh.set_num_want(1000); seventh.set_pcd_images(fa2, da2, tuple(cam2))
    assert seventh.shared_cloud_count() == 0
    eighth = hiplib.Cvo(); eighth.set_num_want(1000); eighth.set_pcd_images(fa2, da2, tuple(cam2))
    assert eighth.shared_cloud_count() == 1
    a, b = seventh.get_cloud(0), eighth.get_cloud(0)
    np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    # the generator of the shared frame closes first: the copy lives on
    seventh.close()
    b2 = eighth.get_cloud(0); np.testing.assert_array_equal(b[0], b2[0])
    for g in (odo, kf, third, fourth, fifth, sixth, eighth):
        g.close()


def test_a_frame_staged_ahead_is_taken_by_the_set_pcd_that_asks_for_it(hiplib, oracle):
    """cvo_stage_next_frame: frame t + 1's cloud generated on a worker thread while frame t is aligned.  The set_pcd / match_* that is later given those very images takes
    the staged cloud (identical to the oracle's and to a cloud generated on the spot), the frame's second object takes its copy as always; a frame that differs in a
    byte, or was never staged, is generated; a staged frame nobody asks for is dropped; nothing about the alignments changes."""
    from cvo_slam_amd import synth
    (fa, da), (fb, db), camt, _ = frames(3, synth.TUM1)
    (fc, dc), _, _, _ = frames(6, synth.TUM1)
    want_a, want_b, want_c = oracle.pcd_generate(fa, da, camt), oracle.pcd_generate(fb, db, camt), oracle.pcd_generate(fc, dc, camt)
    # reference run without staging
    ref_o, ref_k = hiplib.Cvo(), hiplib.Cvo()
    ref_o.set_pcd_images(fa, da, camt); ref_k.set_pcd_images(fa, da, camt)
    tf_ref_o = ref_o.match_odometry_images(fb, db, camt); tf_ref_k = ref_k.match_keyframe_images(fb, db, camt)
    ref_o.update_fixed_pcd(); tf_ref_o2 = ref_o.match_odometry_images(fc, dc, camt)
    # the same sequence with every next frame staged while the current one is tracked
    odo, kf = hiplib.Cvo(), hiplib.Cvo()
    odo.set_pcd_images(fa, da, camt); kf.set_pcd_images(fa, da, camt)
    odo.stage_next_frame(fb, db, camt)                                                     # frame 1, ahead of its set_pcd
    tf_o = odo.match_odometry_images(fb, db, camt)
    assert odo.staged_frame_count() == 1 and odo.shared_cloud_count() == 0
    odo.stage_next_frame(fc, dc, camt)                                                     # frame 2, staged while frame 1's keyframe alignment runs
    tf_k = kf.match_keyframe_images(fb, db, camt)                                          # (frame 1's second object: the copy of the thread's last taken frame)
    assert kf.shared_cloud_count() == 2 and kf.staged_frame_count() == 0
    check_cloud(hiplib, odo, 1, want_b); check_cloud(hiplib, kf, 1, want_b)
    np.testing.assert_array_equal(tf_o, tf_ref_o); np.testing.assert_array_equal(tf_k, tf_ref_k)
    odo.update_fixed_pcd()
    tf_o2 = odo.match_odometry_images(fc, dc, camt)
    assert odo.staged_frame_count() == 2
    check_cloud(hiplib, odo, 1, want_c); np.testing.assert_array_equal(tf_o2, tf_ref_o2)
    # staged, then asked for with one byte changed: generated, not taken
    other = hiplib.Cvo()
    other.stage_next_frame(fa, da, camt)
    fa2 = fa.copy(); fa2[0, 0, 0] ^= 1
    other.set_pcd_images(fa2, da, camt)
    assert other.staged_frame_count() == 0
    check_cloud(hiplib, other, 0, oracle.pcd_generate(fa2, da, camt))
    # the staged frame is still there for whoever asks for it; a second staging replaces one nobody asked for
    again = hiplib.Cvo(); again.set_pcd_images(fa, da, camt)
    assert again.staged_frame_count() == 1
    check_cloud(hiplib, again, 0, want_a)
    again.stage_next_frame(fb, db, camt); again.stage_next_frame(fc, dc, camt)
    again.set_pcd_images(fb, db, camt)                                                     # fb was dropped: generated here
    assert again.staged_frame_count() == 1
    check_cloud(hiplib, again, 1, want_b)
    for g in (ref_o, ref_k, odo, kf, other, again):
        g.close()
