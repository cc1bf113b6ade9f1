#!/usr/bin/env python3
"""One pair alone on eight workgroups: the phase seconds of workgroup 0 (cvo_batch_last_phase_seconds), cull and sort apart."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cvo_slam_amd as ca
from cvo_slam_amd import synth
for pid in (0, 5):
    pr = synth.make_pair(pid)
    for wgs in (8, 4, 1):
        b = ca.CvoBatch(1); b.set_workgroups(wgs)
        b.set_pairs([(pr.fixed.xyz, pr.fixed.feat, pr.moving.xyz, pr.moving.feat)])
        for rep in range(3):
            b.reset_states(); b.align_async(1); r = b.wait(1)[0]
        ph = b.last_phase_seconds(); masks, _ = b.last_cull_masks(1)
        print(f"pair {pid} wgs {wgs}: iterations {r['iterations_run']}, culls {bin(masks[0]).count('1')}; us per pair: " + ", ".join(f"{k} {1e6 * v:.1f}" for k, v in ph.items()))
        b.close()
