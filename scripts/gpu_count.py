"""experiment (CVO_COUNT build): list entries evaluated by the candidate phase (padding included) against listed candidates and nonzeros"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
n = 16
pairs = [synth.make_pair(i) for i in range(n)]
B = ca.CvoBatch(n); B.set_workgroups(1)
for i, p in enumerate(pairs):
    B.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
B.align_async(n); res = B.wait(n)
info = B.last_launch(); ph = B.last_phase_seconds()
its = sum(r["iterations_run"] for r in res)
slots = ph["cand_exchange"] * 1e8
print(json.dumps(dict(iterations=its, candidates=info["candidates_total"], slots_evaluated=slots, slots_per_candidate=slots / info["candidates_total"],
                      candidates_per_iteration=info["candidates_total"] / its)))
