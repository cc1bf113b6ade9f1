"""Which pair a position of a launch works on (cvo_batch_align_async ranks the pairs by the density of their clouds and deals them to
the positions: eighths of the ranking per XCD class for launches with a slot per pair, densest first for launches that pull from the
in-kernel queue; a pair's workgroups on one XCD when the slots come in eights) is a scheduling decision: results, states and records
stay indexed by pair and must not depend on it.  CVO_HIP_ORDER_PAIRS / CVO_HIP_COLOCATE are read when a handle is created."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kv): self.kv = kv
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _run(hiplib, pairs, wgs, max_wgs=0, n=None, adoption=False, **env):
    with _Env(**env):
        b = hiplib.CvoBatch(len(pairs))
    b.set_workgroups(wgs)
    if max_wgs: b.set_max_workgroups(max_wgs)
    b.set_adoption(adoption)
    b.set_pairs([(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs])
    n = n or len(pairs)
    out = []
    for _ in range(2):                                               # twice: the descriptors of the second launch are the cached ones
        b.reset_states(); b.align_async(n); out.append(b.wait(n))
    b.close()
    return out


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x["status"] == 0 and y["status"] == 0
        assert x["iter"] == y["iter"] and x["A_nonzero"] == y["A_nonzero"] and x["iterations_run"] == y["iterations_run"]
        assert np.array_equal(x["transform"], y["transform"])


def _pairs(n, seed0, lo=260, step=41):
    """Clouds of different sizes and depths: the ranking is far from the index order."""
    from cvo_slam_amd import synth
    out = []
    for i in range(n):
        p = synth.make_small_pair(seed0 + i, n=lo + step * (i % 7))
        s = np.float32(0.75 + 0.125 * ((5 * i) % 7))                 # the scene nearer or farther: the density hint (mean 1/z^2) spreads 4x
        out.append(synth.Pair(fixed=synth.Cloud(xyz=p.fixed.xyz * s, feat=p.fixed.feat), moving=synth.Cloud(xyz=p.moving.xyz * s, feat=p.moving.feat),
                              true_transform=p.true_transform))
    return out


@pytest.mark.parametrize("mode", ["1", "2", "4", "5", "8", "10", "11", "23", "36"])
def test_slot_per_pair_launch_is_indifferent_to_the_order(hiplib, mode):
    pairs = _pairs(19, 4100)                                         # 19: the eighths of the ranking have 3,3,3,2,2,2,2,2 pairs
    want = _run(hiplib, pairs, 1, CVO_HIP_ORDER_PAIRS="0")
    got = _run(hiplib, pairs, 1, CVO_HIP_ORDER_PAIRS=mode)
    for w, g in zip(want, got): _same(w, g)
    _same(want[0], want[1])


def test_partial_launch_uses_the_ranking_of_its_own_pairs(hiplib):
    """A launch over the first n pairs of the batch ranks those n only (positions beyond n do not exist)."""
    pairs = _pairs(13, 4200)
    want = _run(hiplib, pairs, 1, CVO_HIP_ORDER_PAIRS="0")
    for n in (1, 2, 7, 9):
        got = _run(hiplib, pairs, 1, n=n)
        _same(want[0][:n], got[0]); _same(want[0][:n], got[1])


def test_order_with_adoption(hiplib):
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(300 + i) for i in range(24)]            # full-size pairs: helpers do join
    want = _run(hiplib, pairs, 1, CVO_HIP_ORDER_PAIRS="0")
    got = _run(hiplib, pairs, 1, adoption=True)
    for w, g in zip(want, got): _same(w, g)


@pytest.mark.parametrize("G,max_wgs,colocate", [(2, 16, "1"), (2, 16, "0"), (4, 32, "1"), (4, 32, "0"), (4, 12, "1"), (1, 8, "1")])
def test_queue_launch_densest_first_and_members_on_one_xcd(hiplib, G, max_wgs, colocate):
    """Fewer slots than pairs: pairs pulled densest first; slots in eights (16/2, 32/4): a pair's members are the blocks congruent mod 8;
    12/4 = 3 slots: consecutive blocks whatever the knob says."""
    pairs = _pairs(21, 4300, lo=420, step=53)
    want = _run(hiplib, pairs, 1, CVO_HIP_ORDER_PAIRS="0")           # one workgroup and one slot per pair, index order
    got = _run(hiplib, pairs, G, max_wgs=max_wgs, CVO_HIP_COLOCATE=colocate)
    for w, g in zip(want, got): _same(w, g)
    got0 = _run(hiplib, pairs, G, max_wgs=max_wgs, CVO_HIP_COLOCATE=colocate, CVO_HIP_ORDER_PAIRS="0")
    _same(want[0], got0[0])


@pytest.mark.parametrize("G", [1, 2, 4])
def test_both_builds_of_the_kernel_agree_on_plane_layout_launches(hiplib, G):
    """Launches in the plane layout (CVO_HIP_Y_MODE=2 forces it for small clouds; 9 k-point clouds get it by themselves) run the build with three
    waves per SIMD (768-thread workgroups); CVO_HIP_WIDE=0 keeps them on the two-wave build.  Same results, and those of the resident layout."""
    pairs = _pairs(9, 4400, lo=900, step=97)                          # 900 ... 1482 points: more rows than 768 threads, several blocks per wave
    want = _run(hiplib, pairs, 1, CVO_HIP_ORDER_PAIRS="0")            # default layout (16-byte points resident), two waves per SIMD
    wide = _run(hiplib, pairs, G, CVO_HIP_Y_MODE="2", CVO_HIP_WIDE="1")
    narrow = _run(hiplib, pairs, G, CVO_HIP_Y_MODE="2", CVO_HIP_WIDE="0")
    for w, a, b in zip(want, wide, narrow):
        _same(w, a); _same(w, b)
