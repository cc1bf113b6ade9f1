"""Times cvo_set_pcd_images (GPU point-cloud generator) on one synthetic 640x480 frame; run under rocprofv3 --kernel-trace --stats
for the per-kernel split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
cam = synth.ETH3D if os.environ.get("SHAPE") == "eth3d" else synth.TUM1
(fa, da), (fb, db), _ = synth.make_frames(int(os.environ.get("PAIR", "0")), cam=cam)
camt = synth.camera_tuple(cam)
g = ca.Cvo()
ts = []
for i in range(int(os.environ.get("REPS", "40"))):
    t0 = time.perf_counter(); g.set_pcd_images(fa if i % 2 == 0 else fb, da if i % 2 == 0 else db, camt); ts.append(time.perf_counter() - t0)
print(f"set_pcd_images: median {1e3*np.median(ts[4:]):.3f} ms, min {1e3*min(ts):.3f} ms; points {g.get_cloud(1)[0].shape[0]}")
