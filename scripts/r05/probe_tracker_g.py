#!/usr/bin/env python3
"""The tracker probe's alignment (clouds generated from the images of frames A and B: 2 817 points, fresh object) at different workgroup counts: host wall per alignment, median of 7."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
camt = synth.camera_tuple(synth.TUM1)
(fa, da), (fb, db), _ = synth.make_frames(0)
for rep in range(2):
    for wgs in (0, 4, 6, 8, 12):
        ts = []
        for _ in range(7):
            g = ca.Cvo(); g.set_workgroups(wgs); g.set_pcd_images(fa, da, camt); g.set_pcd_images(fb, db, camt)
            t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0); g.close()
        print(f"workgroups {wgs or 'auto'}: {1e3 * np.median(ts):.3f} ms per alignment (min {1e3 * min(ts):.3f})")
