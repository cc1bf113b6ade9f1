"""development aid: are full-size results bit-identical across workgroups per pair?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
ca.load_library()
pairs = [synth.make_pair(i) for i in range(24)]
def run(G):
    b = ca.CvoBatch(len(pairs)); b.set_workgroups(G)
    for i, p in enumerate(pairs): b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    b.align_async(24); r = b.wait(24); b.close(); return r
want = run(1)
for G in (2, 3, 4, 8):
    got = run(G)
    bad = [(i, w["iter"], g["iter"], float(np.abs(g["transform"] - w["transform"]).max())) for i, (w, g) in enumerate(zip(want, got)) if g["iter"] != w["iter"] or not np.array_equal(g["transform"], w["transform"])]
    print("G", G, "bad", bad)
