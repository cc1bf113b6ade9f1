"""BASELINE config 5 AS BENCHMARKED (`bench.py --shape eth3d`): 64 ETH3D-shape 736x456 pairs (~9.3 k points per cloud), four
workgroups per pair, launches capped to 40 workgroups (10 pair slots: the pairs are handed out by the in-kernel queue), six
batch objects in flight side by side -- and EVERY launch's 64 results compared with the oracle (KD-tree search, host threads):
pose within the north-star tolerance, iteration count and nnz of the last iteration equal."""
import os
from concurrent.futures import ThreadPoolExecutor

import pytest

from helpers import rot_trans_err

pytestmark = pytest.mark.gpu

N_PAIRS, DEPTH, ROUNDS = 64, 6, 2


def _oracle_result(args):
    import pyoracle as po
    fx, ff, mx, mf = args
    o = po.OracleCvo(search=po.SEARCH_KDTREE, threads=1)
    o.set_pcd(fx, ff); o.set_pcd(mx, mf)
    assert o.align()[0] == 0
    st = o.get_state()
    return st["transform"].copy(), st["iter"], st["A_nonzero"]


def test_config5_as_benchmarked_every_concurrent_launch_matches_the_oracle(hiplib, oracle):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(i, cam=synth.ETH3D) for i in range(N_PAIRS)]     # the pairs `bench.py --shape eth3d` times
    assert min(p.fixed.n for p in pairs) > 8000
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(workers) as ex:
        want = list(ex.map(_oracle_result, clouds))
    batches = []
    for _ in range(DEPTH):
        b = hiplib.CvoBatch(N_PAIRS)
        b.set_workgroups(4); b.set_max_workgroups(40)                          # bench.py's config-5 defaults
        b.set_pairs(clouds)
        batches.append(b)
    checked = 0
    for rnd in range(ROUNDS):
        for b in batches:
            b.reset_states(); b.align_async(N_PAIRS)
        for bi, b in enumerate(batches):
            res = b.wait(N_PAIRS)
            for i, (r, (tf, it, nnz)) in enumerate(zip(res, want)):
                assert r["status"] == 0 and r["dense_fallbacks"] == 0, (rnd, bi, i, r["status"])
                rot, tr = rot_trans_err(r["transform"], tf)
                assert rot <= 1e-4 and tr <= 1e-4, (rnd, bi, i, rot, tr)
                assert r["iter"] == it and r["A_nonzero"] == nnz, (rnd, bi, i, r["iter"], it, r["A_nonzero"], nnz)
                checked += 1
    assert checked == ROUNDS * DEPTH * N_PAIRS
    for b in batches:
        b.close()
