"""BASELINE config 3 AS BENCHMARKED: 64 full-size 640x480 TUM-shape pairs (~3 k points per cloud), eight batch objects in
flight on eight HIP streams (bench.py's timed region), and EVERY launch's 64 results compared with the oracle: pose within the
north-star tolerance (rotation <= 1e-4 rad, translation <= 1e-4 m), iteration count and nnz of the last iteration equal.
The oracle aligns the 64 pairs once (KD-tree search, host threads); three rounds of eight concurrent launches are checked."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from helpers import rot_trans_err

pytestmark = pytest.mark.gpu

N_PAIRS, DEPTH, ROUNDS = 64, 8, 3


def _oracle_result(args):
    import pyoracle as po
    fx, ff, mx, mf = args
    o = po.OracleCvo(search=po.SEARCH_KDTREE, threads=1)
    o.set_pcd(fx, ff); o.set_pcd(mx, mf)
    assert o.align()[0] == 0
    st = o.get_state()
    return st["transform"].copy(), st["iter"], st["A_nonzero"]


def test_config3_as_benchmarked_every_concurrent_launch_matches_the_oracle(hiplib, oracle):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")          # read when the runtime initialises; harmless later
    from cvo_slam_amd import synth
    pairs = [synth.make_pair(i) for i in range(N_PAIRS)]     # the pairs bench.py times
    clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(workers) as ex:                   # ctypes releases the GIL inside the oracle
        want = list(ex.map(_oracle_result, clouds))

    batches = []
    for _ in range(DEPTH):
        b = hiplib.CvoBatch(N_PAIRS)
        b.set_workgroups(1)                                  # the bench's throughput configuration
        for i, c in enumerate(clouds):
            b.set_pair(i, *c)
        batches.append(b)
    checked = 0
    for rnd in range(ROUNDS):
        for b in batches:                                    # eight launches queued back to back on eight streams ...
            b.reset_states(); b.align_async(N_PAIRS)
        for bi, b in enumerate(batches):                     # ... and every one of them checked
            res = b.wait(N_PAIRS)
            assert b.last_launch()["kernel_ms"] > 0
            for i, (r, (tf, it, nnz)) in enumerate(zip(res, want)):
                assert r["status"] == 0, (rnd, bi, i)
                rot, tr = rot_trans_err(r["transform"], tf)
                assert rot <= 1e-4 and tr <= 1e-4, (rnd, bi, i, rot, tr)
                assert r["iter"] == it and r["A_nonzero"] == nnz, (rnd, bi, i, r["iter"], it, r["A_nonzero"], nnz)
                checked += 1
    assert checked == ROUNDS * DEPTH * N_PAIRS
    for b in batches:
        b.close()
