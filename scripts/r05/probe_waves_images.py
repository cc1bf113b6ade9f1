#!/usr/bin/env python3
"""Experiment build (-DCVO_KTRACE -DCVO_KTRACE_WAVES): when each wave of workgroup 0 leaves the candidate walk, for the tracker probe's alignment (clouds generated from images) and the synthetic pair 0, automatic workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
camt = synth.camera_tuple(synth.TUM1)
(fa, da), (fb, db), _ = synth.make_frames(0)
pr = synth.make_pair(0)
for name in ("from images", "synthetic pair 0"):
    g = ca.Cvo()
    if name == "from images": g.set_pcd_images(fa, da, camt); g.set_pcd_images(fb, db, camt)
    else: g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
    rows = g.align(trace_cap=120)
    print(f"{name}: {len(rows)} iterations; walk's end by wave (us): k: min / mean / max, wave 0, latest wave")
    for k, r in enumerate(rows):
        if k in (1, 2, 5, 8, 12, 15, 25, 35):
            om = [x / 100 for x in r["omega"]]; v = r["v"]
            print(f"  {k:2d}: {om[0]:6.2f} / {om[1]:6.2f} / {om[2]:6.2f}, {v[0] / 100:6.2f}, wave {int(v[1])}")
    g.close()
