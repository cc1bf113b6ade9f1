"""Hand-over of host clouds (cvo_set_pcd / cvo_batch_set_pair / cvo_batch_set_pairs): the arrays cross the boundary in the reference
layout (positions AoS data_type.h:30, features channel-major data_type.h:75), are copied as they are and packed into the device
layout by a kernel.  Plus the per-cloud cache of fip(cloud, cloud) (cvo.cpp:496-497)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cloud_round_trip_through_the_device_layout(hiplib):
    rng = np.random.default_rng(3)
    for n in (1, 31, 256, 257, 3072, 9300):                           # below / at / above the pack kernel's 256-point blocks
        xyz = rng.standard_normal((n, 3)).astype(np.float32); feat = (255 * rng.random((5, n))).astype(np.float32)
        g = hiplib.Cvo()
        g.set_pcd(xyz, feat)
        x2, f2 = g.get_cloud(hiplib.api.SLOT_FIXED)
        np.testing.assert_array_equal(x2, xyz); np.testing.assert_array_equal(f2, feat)
        g.close()


def test_set_pairs_is_set_pair_for_every_pair(hiplib):
    from cvo_slam_amd import synth
    pairs = [synth.make_small_pair(40 + i, n=[300, 1, 777, 2048, 64, 1500, 333][i]) for i in range(7)]
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz[: max(1, p.moving.n - i)], p.moving.feat[:, : max(1, p.moving.n - i)]) for i, p in enumerate(pairs)]
    a = hiplib.CvoBatch(len(clouds)); b = hiplib.CvoBatch(len(clouds) + 2)
    for i, c in enumerate(clouds):
        a.set_pair(i, *c)
    prep = hiplib.CvoBatch.prepare_pairs(clouds)
    b.set_pairs(prep)                                                 # pairs 0 .. 6 in one hand-over
    ra, rb = a.align(len(clouds)), b.align(len(clouds))
    for x, y in zip(ra, rb):
        assert x["status"] == y["status"] and x["iter"] == y["iter"] and x["A_nonzero"] == y["A_nonzero"]
        np.testing.assert_array_equal(x["transform"], y["transform"])
    # again over the same object (the staging ring is reused), at an offset, after the states were reset
    b.set_pairs(prep, first=2)
    rc = b.align(len(clouds) + 2)
    for x, y in zip(ra, rc[2:]):
        np.testing.assert_array_equal(x["transform"], y["transform"])
    with pytest.raises(hiplib.CvoError):
        b.set_pairs(prep, first=3)                                    # does not fit
    a.close(); b.close()


def test_large_hand_over_uses_the_copy_threads_and_matches(hiplib):
    """64 full-size pairs = 12.6 MB per hand-over: above the threshold where several threads copy into the staging block."""
    from cvo_slam_amd import synth
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in (synth.make_pair(i) for i in range(12))] * 6
    b = hiplib.CvoBatch(len(clouds))
    b.set_pairs(clouds)
    ref = hiplib.CvoBatch(12)
    for i in range(12):
        ref.set_pair(i, *clouds[i])
    want = ref.align(12); got = b.align(len(clouds))
    for i, g in enumerate(got):
        np.testing.assert_array_equal(g["transform"], want[i % 12]["transform"])
    b.close(); ref.close()


def _env(**kw):
    import contextlib

    @contextlib.contextmanager
    def cm():
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
    return cm()


def test_self_inner_product_cache_changes_nothing(hiplib, oracle):
    """fip(fixed, fixed) and fip(moving, moving) (cvo.cpp:496-497) are kept with the cloud per ell and travel with it through
    update_fixed_pcd (cvo.cpp:578-582).  Tracker sequence of four frames, scores after every alignment: with the cache (default) and
    with CVO_HIP_SELF_CACHE=0 the same numbers, and the oracle's."""
    from cvo_slam_amd import synth
    frames = [synth.make_small_pair(60 + i, n=700) for i in range(4)]
    clouds = [(frames[0].fixed.xyz, frames[0].fixed.feat)] + [(f.moving.xyz, f.moving.feat) for f in frames]

    def run(make):
        g = make(); out = []
        g.set_pcd(*clouds[0])
        for c in clouds[1:]:
            tf = g.match_odometry(*c)
            for rep in range(2):                                      # twice: the second block finds both self products cached
                sc = g.compute_innerproduct(np.asarray(tf, np.float32))
                out.append((sc["inn_fixed_pcd"], sc["inn_moving_pcd"], sc["inn_post"], sc["cos_angle"], sc["inliers"]))
            g.update_fixed_pcd()
        return out

    with_cache = run(lambda: hiplib.Cvo())
    with _env(CVO_HIP_SELF_CACHE=0):
        without = run(lambda: hiplib.Cvo())
    assert with_cache == without

    class O:                                                          # the oracle object behind the same calls
        def __init__(self): self.o = oracle.OracleCvo()
        def set_pcd(self, x, f): self.o.set_pcd(x, f)
        def match_odometry(self, x, f):
            rc, tf = self.o.match(x, f); assert rc == 0; return tf
        def compute_innerproduct(self, tf):
            rc, s = self.o.compute_innerproduct(tf); assert rc == 0; return s
        def update_fixed_pcd(self): self.o.update_fixed_pcd()
    want = run(lambda: O())
    for g, w in zip(with_cache, want):
        for k in range(3):
            assert g[k][1] == w[k][1] and g[k][0] == pytest.approx(w[k][0], rel=1e-5)
        assert g[3] == pytest.approx(w[3], rel=1e-5) and g[4] == w[4]


def test_batch_score_block_with_cached_self_products(hiplib):
    """The batch's queued tracker block (cvo_batch_enqueue_innerproduct) run three times over the same clouds: the first computes and
    keeps fip(cloud, cloud) on the device (ell comes from the device-resident state), the others find it there -- same results."""
    from cvo_slam_amd import synth
    pairs = [synth.make_small_pair(80 + i, n=500 + 40 * i) for i in range(9)]      # 45 requests: descriptors through HBM
    B = hiplib.CvoBatch(len(pairs))
    B.set_pairs([(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs])
    runs = []
    for rnd in range(3):
        B.reset_states(); B.align_async(len(pairs)); B.enqueue_innerproduct(len(pairs)); B.wait(len(pairs))
        runs.append(B.innerproduct_results(len(pairs)))
    for later in runs[1:]:
        for a, b in zip(runs[0], later):
            for key in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd"):
                assert a[key] == b[key]
            assert a["cos_angle"] == b["cos_angle"] and a["inliers"] == b["inliers"]
            np.testing.assert_array_equal(a["post_hessian"], b["post_hessian"])
    B.close()


def test_done_polls_without_blocking_and_the_record_table_is_on_the_device(hiplib):
    """cvo_batch_done (a caller with several batches in flight reuses whichever finished) and cvo_batch_result_records (the 64-byte
    records the align kernel writes, where a launcher's own collective can pick them up)."""
    import time
    import torch
    from cvo_slam_amd import shard, synth
    pairs = [synth.make_pair(i) for i in range(8)]
    b = hiplib.CvoBatch(len(pairs)); b.set_workgroups(1)
    assert b.done()                                                   # nothing launched yet
    b.set_pairs([(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs])
    b.align_async(len(pairs))
    polls = 0
    t0 = time.perf_counter()
    while not b.done():
        polls += 1
        assert time.perf_counter() - t0 < 10.0
    res = b.wait(len(pairs))
    assert polls > 0 and b.done()                                     # full-size pairs take milliseconds: the first poll came back before they were done
    rec = shard.device_view(b.result_records(), len(pairs)).cpu().numpy()
    for i, r in enumerate(res):
        np.testing.assert_array_equal(rec[i, :12].reshape(3, 4), r["transform"])
        assert (int(rec[i, 12]), int(rec[i, 13]), int(rec[i, 14]), int(rec[i, 15])) == (r["iter"], r["A_nonzero"], r["iterations_run"], r["status"])
    b.close()


def test_clouds_in_registered_memory_are_read_in_place(hiplib):
    """cvo_host_register: clouds handed over from inside a registered range are not staged -- the align launch reads them where they lie (one workgroup per
    pair), or the pack kernel does (cooperative launches, score blocks).  Same results as the staged hand-over; a batch may mix both kinds; the caller may
    rewrite the arrays between steps (waited-for launches have read them)."""
    from cvo_slam_amd import synth, api
    pairs = [synth.make_small_pair(60 + i, n=[900, 333, 1500, 64, 2048][i]) for i in range(5)]
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
    ref = hiplib.CvoBatch(len(clouds)); ref.set_pairs(clouds)
    want = ref.align(len(clouds))
    # one arena for all arrays of pairs 0 .. 3 (pair 4 stays in ordinary memory: a mixed hand-over)
    total = sum(a.size for c in clouds[:4] for a in c)
    arena = np.zeros(total + 64, np.float32)
    views, off = [], 0
    for c in clouds[:4]:
        vs = []
        for a in c:
            v = arena[off: off + a.size].reshape(a.shape); v[...] = a; off += a.size; vs.append(v)
        views.append(tuple(vs))
    api.host_register(arena)
    try:
        mixed = views + [clouds[4]]
        for wgs in (1, 4):                                             # in-kernel packing (G = 1) and the pack kernel (cooperative launch)
            b = hiplib.CvoBatch(len(clouds)); b.set_workgroups(wgs)
            b.set_pairs(mixed)
            got = b.align(len(clouds))
            for x, y in zip(want, got):
                assert x["status"] == y["status"] == 0 and x["iter"] == y["iter"] and x["A_nonzero"] == y["A_nonzero"]
                np.testing.assert_array_equal(x["transform"], y["transform"])
            b.close()
        # rewrite: pair k's slot now holds a cloud of the same shape scaled away from the camera a little; results follow the new content
        b = hiplib.CvoBatch(1)
        b.set_pairs([views[0]]); r0 = b.align(1)[0]
        views[0][2][...] = clouds[0][2] + np.float32(0.004)            # moving cloud shifted by 4 mm
        b.set_pairs([views[0]]); r1 = b.align(1)[0]
        plain = hiplib.CvoBatch(1); plain.set_pairs([(clouds[0][0], clouds[0][1], clouds[0][2] + np.float32(0.004), clouds[0][3])]); r2 = plain.align(1)[0]
        np.testing.assert_array_equal(r0["transform"], want[0]["transform"])
        np.testing.assert_array_equal(r1["transform"], r2["transform"])
        assert not np.array_equal(r0["transform"], r1["transform"])
        b.close(); plain.close()
    finally:
        api.host_unregister(arena)
    # unregistered again: staged like any memory
    c = hiplib.CvoBatch(4); c.set_pairs(views)
    views[1][2][...] = 0                                               # (staged: the caller may scribble right away)
    got = c.align(4)
    np.testing.assert_array_equal(got[1]["transform"], want[1]["transform"])
    with pytest.raises(hiplib.CvoError):
        api.host_unregister(arena)
    ref.close(); c.close()
