"""Score-block timing (development aid): the tracker score block of a 64-pair batch, alone on the GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import cvo_slam_amd as ca
n = int(os.environ.get("PAIRS", "64"))
pairs = bench.generate_pairs(0, n, 1 if os.environ.get("NOFORK") else 0)
B = ca.CvoBatch(n)
B.set_workgroups(1)
for i, (_, fx, ff, mx, mf) in enumerate(pairs):
    B.set_pair(i, fx, ff, mx, mf)
B.align(n)
B.compute_innerproduct(n)
ts = []
for _ in range(int(os.environ.get("REPS", "20"))):
    t = time.perf_counter(); r = B.compute_innerproduct(n); ts.append(1e3 * (time.perf_counter() - t))
print(f"score block of {n} pairs ({5 * n} requests): median {np.median(ts):.3f} ms, min {min(ts):.3f} ms; "
      f"mean pairs in inn_post {np.mean([x['inn_post'][1] for x in r]):.0f}, inn_fixed {np.mean([x['inn_fixed_pcd'][1] for x in r]):.0f}, "
      f"inn_pre {np.mean([x['inn_pre'][1] for x in r]):.0f}, inliers {np.mean([x['inliers'] for x in r]):.0f}")
te, tw, tr = [], [], []
for _ in range(20):
    t0 = time.perf_counter(); B.enqueue_innerproduct(n); t1 = time.perf_counter(); B.wait(); t2 = time.perf_counter(); B.innerproduct_results(n); t3 = time.perf_counter()
    te.append(1e3 * (t1 - t0)); tw.append(1e3 * (t2 - t1)); tr.append(1e3 * (t3 - t2))
print(f"host time: enqueue {np.median(te):.3f} ms, wait {np.median(tw):.3f} ms, results {np.median(tr):.3f} ms")
ta, twa = [], []
for _ in range(10):
    B.reset_states(); t0 = time.perf_counter(); B.align_async(n); t1 = time.perf_counter(); B.wait(n); t2 = time.perf_counter()
    ta.append(1e3 * (t1 - t0)); twa.append(1e3 * (t2 - t1))
print(f"host time: align_async {np.median(ta):.3f} ms, wait+results {np.median(twa):.3f} ms")
from cvo_slam_amd.api import TrackScores, _check
out = (TrackScores * n)()
tc = []
for _ in range(10):
    B.enqueue_innerproduct(n); B.wait()
    t0 = time.perf_counter(); _check(B.L.cvo_batch_innerproduct_results(B.h, n, out)); tc.append(1e3 * (time.perf_counter() - t0))
print(f"host time inside the C call cvo_batch_innerproduct_results: {np.median(tc):.3f} ms")
