import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["CVO_HIP_FUSE_REFINE"] = "1"; os.environ["CVO_HIP_WGS"] = "1"
import cvo_slam_amd as ca
from cvo_slam_amd import synth
p = synth.make_pair(1)
g = ca.Cvo(); g.set_pcd(p.fixed.xyz, p.fixed.feat); g.set_pcd(p.moving.xyz, p.moving.feat)
tr = g.align(12); print([r["nnz"] for r in tr]); g.close()
