import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
p = synth.make_small_pair(11, n=300)
g = ca.Cvo(); g.set_workgroups(int(os.environ.get("WGS", "1")))
g.set_pcd(p.fixed.xyz, p.fixed.feat); g.set_pcd(p.moving.xyz, p.moving.feat)
tr = g.align(trace_cap=8)
for r in tr[:4]:
    print(r)
print(g.transform)
