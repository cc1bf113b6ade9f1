"""Latency of the single-object path (BASELINE config 2): align() of one full-size pair for several workgroup
counts, and the post-align score block (compute_innerproduct: 4 inner products + 1 Hessian, cvo.cpp:475-503)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
pr = synth.make_pair(int(os.environ.get("PAIR", "0")), cam=synth.ETH3D if os.environ.get("SHAPE") == "eth3d" else synth.TUM1)
print("points", pr.fixed.n, pr.moving.n)
for wgs in [int(x) for x in os.environ.get("WGS", "0,32,16,8,4,1").split(",")]:
    ts = []
    for rep in range(5):
        g = ca.Cvo(); g.set_workgroups(wgs)
        g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
        t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0)
        its = g.get_iteration_number()
        if rep < 4: g.close()
    tf = g.transform
    sc = []
    for rep in range(5):
        t0 = time.perf_counter(); g.compute_innerproduct(tf); sc.append(time.perf_counter() - t0)
    print(f"wgs={wgs}: align {1e3*min(ts):.2f} ms (median {1e3*np.median(ts):.2f}), iterations {its + 1}; score block {1e3*min(sc):.2f} ms (median {1e3*np.median(sc):.2f})")
    g.close()
