#!/usr/bin/env python3
"""Experiment build (-DCVO_KTRACE): the tracker probe's alignment (clouds generated from the images of frames A and B, fresh object, automatic workgroups) phase by phase,
beside the same scene's synthetic pair 0: sums over the iterations (us) and the heavy iterations one by one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
camt = synth.camera_tuple(synth.TUM1)
(fa, da), (fb, db), _ = synth.make_frames(0)
pr = synth.make_pair(0)
def show(name, rows):
    S = np.zeros(9)
    for r in rows:
        B, C, D, E = [x / 100 for x in r["BCDE"]]; om = [x / 100 for x in r["omega"]]; v = [x / 100 for x in r["v"]]
        S += np.array([B, C, om[1], om[2], C - sum(om), D, v[0], D - v[0] - v[1], E])
    print(f"{name}: {len(rows)} iterations; sums (us): lists {S[0]:.0f} | candidates {S[1]:.0f} (rows {S[2]:.0f}, reduce {S[3]:.0f}, waiting {S[4]:.0f}) | line search {S[5]:.0f} (walk {S[6]:.0f}, waiting {S[7]:.0f}) | epilogue {S[8]:.0f} | all {S[0] + S[1] + S[5] + S[8]:.0f}")
for rep in range(2):
    g = ca.Cvo(); g.set_pcd_images(fa, da, camt); g.set_pcd_images(fb, db, camt)
    rows = g.align(trace_cap=120); show("from images", rows); g.close()
    g = ca.Cvo(); g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
    rows = g.align(trace_cap=120); show("synthetic pair 0", rows); g.close()
