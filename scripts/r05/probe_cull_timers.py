import os, sys
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/cvo_slam_amd') else os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
pr = synth.make_pair(0)
for wgs in (8, 1):
    g = ca.Cvo(); g.set_workgroups(wgs)
    g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
    rows = g.align(trace_cap=100)
    for k, r in enumerate(rows):
        if r["BCDE"][0] > 1000:
            om = [x / 100 for x in r["omega"]]
            print(f"wgs {wgs} k {k}: lists {r['BCDE'][0] / 100:.1f} us; cull: staging {om[0]:.2f}, wave 0's units {om[1]:.2f}, wait for the others {om[2]:.2f}")
    g.close()
