"""Where does a tracker frame's time go with and without the next frame staged?  (host wall per call, ms)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import cvo_slam_amd as ca
from cvo_slam_amd import synth
(fa, da), (fb, db), _ = synth.make_frames(0)
(fc, dc), _, _ = synth.make_frames(1)
camt = synth.camera_tuple(synth.TUM1)
def run(stage, reps=6):
    rows = []
    for _ in range(reps):
        odo, kf = ca.Cvo(), ca.Cvo()
        odo.set_tail_scores(True); kf.set_tail_scores(True)
        odo.set_pcd_images(fa, da, camt); kf.set_pcd_images(fa, da, camt)
        if stage:
            odo.stage_next_frame(fb, db, camt); time.sleep(0.003)
        t = [time.perf_counter()]
        tfo = odo.match_odometry_images(fb, db, camt); t.append(time.perf_counter())
        if stage == 1:
            odo.stage_next_frame(fc, dc, camt)
        t.append(time.perf_counter())
        tfk = kf.match_keyframe_images(fb, db, camt); t.append(time.perf_counter())
        kf.compute_innerproduct(tfk.astype(np.float32)); t.append(time.perf_counter())
        if stage == 1:
            odo.set_pcd_images(fc, dc, camt)
        rows.append([1e3 * (t[i + 1] - t[i]) for i in range(4)] + [1e3 * (t[-1] - t[0])])
        odo.close(); kf.close()
    return np.median(np.array(rows), axis=0)
print("                 match_odometry  stage_call  match_keyframe  scores  total")
print("nothing staged  ", run(0).round(3))
print("staged, + next  ", run(1).round(3))
print("staged, no next ", run(2).round(3))
