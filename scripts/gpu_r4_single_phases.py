#!/usr/bin/env python3
"""One pair alone (BASELINE config 2 / the tracker's call): phase timers of workgroup 0 per iteration for several member counts, the device span of the
pair against the host wall time of the call.  usage: PAIR=0 WGS=8,4,1 python scripts/gpu_r4_single_phases.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
for pid in [int(x) for x in os.environ.get("PAIR", "0,5").split(",")]:
    pr = synth.make_pair(pid, cam=synth.TUM1)
    for wgs in [int(x) for x in os.environ.get("WGS", "8,4,1").split(",")]:
        B = ca.CvoBatch(1, device=0); B.set_workgroups(wgs)
        B.set_pair(0, pr.fixed.xyz, pr.fixed.feat, pr.moving.xyz, pr.moving.feat)
        walls, spans = [], []
        for rep in range(6):
            B.reset_states()
            t0 = time.perf_counter(); r = B.align(1); walls.append(time.perf_counter() - t0)
            s0, s1, _ = B.last_pair_spans(1); spans.append(float(s1[0] - s0[0]))
        it = r[0]["iterations_run"]; ph = B.last_phase_seconds()
        print(f"pair {pid} wgs {wgs}: wall {1e3 * np.median(walls):.3f} ms, device span {1e3 * np.median(spans):.3f} ms, iterations {it}, rebuilds {r[0]['rebuilds']}; "
              f"span/iteration {1e6 * np.median(spans) / it:.1f} us; phases us/iteration: " + str({k: round(1e6 * v / it, 2) for k, v in ph.items()}), flush=True)
        B.close()
