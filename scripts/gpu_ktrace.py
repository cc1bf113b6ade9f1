"""Experiment (needs a build with EXTRA=-DCVO_KTRACE): per-iteration phase times of one pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import cvo_slam_amd as ca
idx = int(os.environ.get("PAIR", "0"))
bench._SHAPE = os.environ.get("SHAPE", "tum")              # eth3d: the BASELINE config 5 shape
(_, fx, ff, mx, mf), = bench.generate_pairs(idx, 1)
prm = ca.default_params()
if os.environ.get("MAX_ITER"): prm.max_iter = int(os.environ["MAX_ITER"])      # experiment builds whose results are garbage must still end
h = ca.Cvo(prm)
for rep in range(2):
    h = ca.Cvo(prm)
    h.set_pcd(fx, ff); h.set_pcd(mx, mf)
    out = h.align(trace_cap=128)
rows = out
print("k ell cand nnz | lists cand ls epi | cand: prologue rows reduce | ls: walk reduce | epi: scalar transform (us)")
tot = np.zeros(4)
for k, r in enumerate(rows):
    t = r["BCDE"] / 100.0; tot += t
    sub = [float(x) / 100.0 for x in list(r["omega"]) + list(r["v"]) + [r["step"]]]
    print(k, round(float(r["ell"]), 2), r["candidates"], r["nnz"], "|", *(round(float(x), 1) for x in t), "|", *(round(x, 1) for x in sub[:3]), "|", *(round(x, 1) for x in sub[3:5]), "|", *(round(x, 1) for x in sub[5:]))
print("totals us", tot.round(0), "sum", tot.sum().round(0))
