#!/usr/bin/env python3
"""Experiment build (-DCVO_KTRACE): per pair on eight workgroups, what member 0 spends waiting for the other members (candidate phase and line search less their own walk and reduction), and the iterations where that exceeds 15 us."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
for pid in range(8):
    pr = synth.make_pair(pid)
    g = ca.Cvo(); g.set_workgroups(8)
    g.set_pcd(pr.fixed.xyz, pr.fixed.feat); g.set_pcd(pr.moving.xyz, pr.moving.feat)
    rows = g.align(trace_cap=120)
    wc = []; wl = []; tot = 0.0; big = []
    for k, r in enumerate(rows):
        B, C, D, E = [x / 100 for x in r["BCDE"]]; om = [x / 100 for x in r["omega"]]; v = [x / 100 for x in r["v"]]
        a = C - sum(om); b = D - v[0] - v[1]; wc.append(a); wl.append(b); tot += B + C + D + E
        if a > 15 or b > 15 or (k > 0 and B > 15): big.append((k, round(B, 1), round(a, 1), round(b, 1)))
    print(f"pair {pid}: {len(rows)} iterations, phases sum {tot:.0f} us; waiting in the candidate phase {sum(wc):.0f} us, in the line search {sum(wl):.0f} us; iterations (k, lists, wait C, wait L) above 15 us: {big}")
    g.close()
