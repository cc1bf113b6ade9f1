"""development aid: which pairs differ with adoption on"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
ca.load_library()
pairs = [synth.make_pair(i) for i in range(24)]
def mk(on):
    b = ca.CvoBatch(len(pairs)); b.set_workgroups(1); b.set_adoption(on)
    for i, p in enumerate(pairs): b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    return b
ref = mk(False); ref.align_async(24); want = ref.wait(24)
b = mk(True)
for rep in range(4):
    b.reset_states(); b.align_async(24); got = b.wait(24)
    bad = [(i, w["iter"], g["iter"], w["A_nonzero"], g["A_nonzero"], float(np.abs(g["transform"] - w["transform"]).max())) for i, (w, g) in enumerate(zip(want, got)) if g["iter"] != w["iter"] or not np.array_equal(g["transform"], w["transform"])]
    print("rep", rep, "helped", b.last_adoptions(), "bad", bad)
