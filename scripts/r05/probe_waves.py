#!/usr/bin/env python3
"""Experiment build (-DCVO_KTRACE -DCVO_KTRACE_WAVES): when each wave of workgroup 0 leaves the candidate walk (100 MHz ticks -> us), iteration by iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
for name, cam, wgs in (("eth3d", synth.ETH3D, 4), ("tum", synth.TUM1, 1)):
    pr = synth.make_pair(0, cam=cam)
    g = ca.Cvo(); g.set_workgroups(wgs)
    g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
    rows = g.align(trace_cap=100)
    print(f"{name} pair 0 on {wgs} workgroup(s): {len(rows)} iterations; walk's end by wave (us): k: min / mean / max, wave 0, latest wave | candidate phase")
    for k, r in enumerate(rows):
        if k in (0, 1, 2, 4, 6, 9, 12, 15, 20, 25, 30, 40, 50, 60, 70):
            om = [x / 100 for x in r["omega"]]; v = r["v"]
            print(f"  {k:2d}: {om[0]:7.2f} / {om[1]:7.2f} / {om[2]:7.2f}, {v[0] / 100:7.2f}, wave {int(v[1])} | {r['BCDE'][1] / 100:7.2f}")
    g.close()
