#!/usr/bin/env python3
"""Experiment build (-DCVO_KTRACE -DCVO_KTRACE_EPI): lane 0's scalar work in the epilogue, piece by piece (100 MHz ticks -> us), one pair alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
for pid in (0, 5):
    pr = synth.make_pair(pid)
    for wgs in (8, 1):
        g = ca.Cvo(); g.set_workgroups(wgs)
        g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
        rows = g.align(trace_cap=100)
        v = np.array([r["v"] for r in rows]); st = np.array([r["step"] for r in rows]); E = np.array([r["BCDE"][3] for r in rows])
        print(f"pair {pid} wgs {wgs}: {len(rows)} iterations; us per iteration: step polynomial {v[:, 0].mean() / 100:.2f}, pose update {v[:, 1].mean() / 100:.2f}, second stop test and the rest {v[:, 2].mean() / 100:.2f}, "
              f"epilogue from lane 0's start to the end of the fused transform {st.mean() / 100:.2f}; epilogue phase {E.mean() / 100:.2f}; step polynomial by iteration: " + " ".join(f"{x / 100:.1f}" for x in v[:12, 0]))
        g.close()
