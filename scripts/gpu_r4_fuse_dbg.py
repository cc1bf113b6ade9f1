import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth

def run(pair, fuse, predict="0.7", steps="8"):
    os.environ["CVO_HIP_FUSE_REFINE"] = fuse; os.environ["CVO_HIP_PREDICT"] = predict; os.environ["CVO_HIP_WGS"] = "1"; os.environ["CVO_HIP_PREDICT_STEPS"] = steps
    g = ca.Cvo(); g.set_pcd(pair.fixed.xyz, pair.fixed.feat); g.set_pcd(pair.moving.xyz, pair.moving.feat)
    tr = g.align(400)
    it = g.get_iteration_number(); g.close()
    return [(r["nnz"], r["candidates"]) for r in tr], it

for idx in (1, 2, 4, 10, 11):
    p = synth.make_pair(idx)
    a, ia = run(p, "0")
    for steps in ("0.001", "0.3", "1", "8"):
        b, ib = run(p, "1", "0.7", steps)
        first = next((k for k in range(min(len(a), len(b))) if a[k][0] != b[k][0]), None)
        print(f"pair {idx} steps cap {steps}: iterations {ia} / fused {ib}; first nnz difference at k = {first}",
              "" if first is None else f"  nnz {a[first][0]} vs {b[first][0]}, candidates {[x[1] for x in a[max(0,first-2):first+2]]} vs {[x[1] for x in b[max(0,first-2):first+2]]}")
