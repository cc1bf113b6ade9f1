"""Quick GPU-vs-oracle check used during development (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as po
import cvo_slam_amd as ca
from cvo_slam_amd import synth

def rot_err(A, B):
    Ra, Rb = A[:, :3].astype(np.float64), B[:, :3].astype(np.float64)
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1))), float(np.linalg.norm(A[:, 3] - B[:, 3]))

small = synth.make_small_pair(1, n=500)
pair = synth.make_pair(0)
for name, pr in [("small500", small), ("tum3072", pair)]:
    o = po.OracleCvo(search=po.SEARCH_KDTREE, threads=8)
    o.set_pcd(pr.fixed.xyz, pr.fixed.feat); o.set_pcd(pr.moving.xyz, pr.moving.feat)
    t0 = time.time(); rc, otr = o.align(trace_cap=2000); t_cpu = time.time() - t0
    ost = o.get_state()
    for wgs in (1, 4, 0):
        g = ca.Cvo()
        g.set_workgroups(wgs)
        g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
        t0 = time.time(); gtr = g.align(trace_cap=2000); t_gpu = time.time() - t0
        re, te = rot_err(g.transform, ost["transform"])
        print(f"{name} wgs={wgs}: oracle iters {ost['iter']} ({len(otr)} rows, {t_cpu*1e3:.1f} ms) gpu iters {g.get_iteration_number()} ({len(gtr)} rows, {t_gpu*1e3:.1f} ms) rot_err {re:.3e} trans_err {te:.3e} nnz {g.get_A_nonzero()} vs {ost['A_nonzero']}")
        for k in range(min(3, len(gtr), len(otr))):
            a, b = gtr[k], otr[k]
            print("   k", k, "nnz", a["nnz"], b["nnz"], "cand", a["candidates"], "domega", np.abs(a["omega"] - b["omega"]).max(), "dv", np.abs(a["v"] - b["v"]).max(),
                  "step", a["step"], b["step"], "BCDE rel", np.abs(a["BCDE"] - b["BCDE"]).max() / np.abs(b["BCDE"]).max())
        nmis = sum(1 for a, b in zip(gtr, otr) if a["nnz"] != b["nnz"])
        print("   rows with nnz mismatch:", nmis, "of", min(len(gtr), len(otr)))
        # scores
        tf = g.transform
        gs = g.compute_innerproduct(tf)
        rc, os_ = o.compute_innerproduct(tf)
        print("   inn_post", gs["inn_post"], os_["inn_post"], "cos", gs["cos_angle"], os_["cos_angle"], "inliers", gs["inliers"], os_["inliers"])
        print("   hessian max rel diff", np.abs(gs["post_hessian"] - os_["post_hessian"]).max() / np.abs(os_["post_hessian"]).max())
        g.close()
# batch
B = ca.CvoBatch(8)
pairs = [synth.make_pair(i) for i in range(4)]
for i in range(8):
    p = pairs[i % 4]
    B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
t0 = time.time(); res = B.align(8); dt = time.time() - t0
print("batch 8:", dt * 1e3, "ms", B.last_launch(), [r["iterations_run"] for r in res], [r["status"] for r in res])
B.reset_states(); res = B.align(8); print("batch 8 again:", B.last_launch())
