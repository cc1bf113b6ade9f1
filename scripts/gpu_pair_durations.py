"""development aid: per-pair durations of the bench's 64 pairs (one launch alone and under load) + cheap predictors, for the drain analysis"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
pairs = [synth.make_pair(i) for i in range(64)]
clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
bs = []
for _ in range(8):
    b = ca.CvoBatch(64); b.set_workgroups(1); b.set_pairs(clouds); bs.append(b)
b = bs[0]
b.align_async(64); res = b.wait(64); alone = b.last_pair_seconds(64)
for rnd in range(3):
    for x in bs: x.reset_states(); x.align_async(64)
    for x in bs: x.wait()
load = np.mean([x.last_pair_seconds(64) for x in bs], axis=0)
its = np.array([r["iterations_run"] for r in res])
meanz = np.array([float(np.mean(p.fixed.xyz[:, 2])) for p in pairs]); invz2 = np.array([float(np.mean(1.0 / p.fixed.xyz[:, 2] ** 2)) for p in pairs])
print(json.dumps({"alone_ms": (1e3 * alone).round(3).tolist(), "load_ms": (1e3 * load).round(3).tolist(), "iterations": its.tolist(), "mean_z": meanz.round(3).tolist(), "mean_inv_z2": invz2.round(4).tolist()}))
