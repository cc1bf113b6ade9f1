"""Where the per-pair work is (development aid): candidates and nonzeros per iteration of a few bench pairs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import cvo_slam_amd as ca
pairs = bench.generate_pairs(0, 4, 1)
for (_, fx, ff, mx, mf) in pairs:
    g = ca.Cvo(); g.set_workgroups(1); g.set_pcd(fx, ff); g.set_pcd(mx, mf)
    tr = g.align(trace_cap=400)
    nnz = np.array([r["nnz"] for r in tr], float); cand = np.array([r["candidates"] for r in tr], float); ell = np.array([r["ell"] for r in tr])
    print(f"iterations {len(tr)}  nnz total {nnz.sum():.3g}  candidates total {cand.sum():.3g}  ratio {cand.sum() / nnz.sum():.2f}")
    for l in (0.15, 0.10, 0.06, 0.03):
        m = np.isclose(ell, l)
        if m.any(): print(f"   ell {l}: {m.sum():3d} iterations, nnz/iter {nnz[m].mean():9.0f}, cand/iter {cand[m].mean():9.0f}, share of nnz {nnz[m].sum() / nnz.sum():.2f}")
    g.close()
