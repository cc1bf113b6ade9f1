#!/usr/bin/env python3
"""Synthetic pair 0 (3 072 points) and the image-generated pair (2 817) at different workgroup counts: host wall per alignment, median of 7."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
camt = synth.camera_tuple(synth.TUM1)
(fa, da), (fb, db), _ = synth.make_frames(0)
for pid in (0, 3, 6):
    pr = synth.make_pair(pid)
    out = []
    for wgs in (6, 8, 10, 12, 16):
        ts = []
        for _ in range(7):
            g = ca.Cvo(); g.set_workgroups(wgs); g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
            t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0); g.close()
        out.append(f"{wgs}: {1e3 * np.median(ts):.3f}")
    print(f"pair {pid} (3072 points), ms by workgroups: " + " | ".join(out))
out = []
for wgs in (6, 8, 10, 12):
    ts = []
    for _ in range(7):
        g = ca.Cvo(); g.set_workgroups(wgs); g.set_pcd_images(fa, da, camt); g.set_pcd_images(fb, db, camt)
        t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0); g.close()
    out.append(f"{wgs}: {1e3 * np.median(ts):.3f}")
print("from images (2817 points), ms by workgroups: " + " | ".join(out))
