#!/usr/bin/env python3
"""Experiment build (-DCVO_KTRACE -DCVO_KTRACE_EPI=2): the epilogue's timeline from the end of lane 0's part A (100 MHz ticks -> us), one pair alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
        v = np.array([r["v"] for r in rows]) / 100; st = np.array([r["step"] for r in rows]) / 100; om = np.array([r["omega"] for r in rows]) / 100; E = np.array([r["BCDE"][3] for r in rows]) / 100; l0 = np.array([r["ell"] for r in rows]) / 100; w0 = np.array([r["dist"] for r in rows]) / 100
        m = lambda a: f"{np.median(a):.2f}"
        print(f"pair {pid} wgs {wgs}: {len(rows)} iterations; medians, us: part A {m(v[:, 0])} | from its end: part B done {m(v[:, 1])}, thread 64 past the barrier {m(om[:, 2])}, its points done {m(v[:, 2])}, "
              f"lane 0 at the barrier {m(l0)}, thread 64 at the barrier {m(w0)}, lane 0 has the maximum {m(st)}, decision made {m(om[:, 0])}, end {m(om[:, 1])}; epilogue phase {m(E)}")
        g.close()
