#!/usr/bin/env python3
"""Experiment build (-DCVO_KTRACE): one pair alone, phase times of workgroup 0 iteration by iteration (100 MHz ticks -> us)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
for pid in (0,):
    pr = synth.make_pair(pid)
    for wgs in (1, 8):
        g = ca.Cvo(); g.set_workgroups(wgs)
        g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
        rows = g.align(trace_cap=100)
        print(f"pair {pid} wgs {wgs}: {len(rows)} iterations; per iteration (us): k: lists | candidates = prologue + rows + wait/reduce (+exchange) | line search = walk + reduce(+exchange) | epilogue = scalar + transform ; nnz")
        for k, r in enumerate(rows):
            if k in (0, 1, 2, 3, 5, 8, 10, 12, 15, 20, 21, 25, 30, 35, 40, 44):
                B, C, D, E = [x / 100 for x in r["BCDE"]]; om = [x / 100 for x in r["omega"]]; v = [x / 100 for x in r["v"]]; st = r["step"] / 100
                print(f"  {k:2d}: {B:6.2f} | {C:6.2f} = {om[0]:.2f} + {om[1]:.2f} + {om[2]:.2f} | {D:6.2f} = {v[0]:.2f} + {v[1]:.2f} | {E:5.2f} = {v[2]:.2f} + {st:.2f} ; {r['nnz']}")
        g.close()
