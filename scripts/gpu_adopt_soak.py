"""development aid: many launches with adoption, several in flight, every result against the batch without it"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
ca.load_library()
N = int(os.environ.get("PAIRS", "48")); ROUNDS = int(os.environ.get("ROUNDS", "40")); DEPTH = int(os.environ.get("DEPTH", "4"))
pairs = [synth.make_pair(200 + i) for i in range(N)]
def mk(on):
    b = ca.CvoBatch(N); b.set_workgroups(1); b.set_adoption(on)
    for i, p in enumerate(pairs): b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    return b
ref = mk(False); ref.align_async(N); want = ref.wait(N); ref.close()
bs = [mk(True) for _ in range(DEPTH)]
bad = 0; helped = 0; t0 = time.time()
for rnd in range(ROUNDS):
    for b in bs: b.reset_states(); b.align_async(N)
    for b in bs:
        got = b.wait(N); helped += b.last_adoptions()
        for i, (w, g) in enumerate(zip(want, got)):
            if g["status"] != 0 or g["iter"] != w["iter"] or not np.array_equal(g["transform"], w["transform"]):
                bad += 1; print("MISMATCH round", rnd, "pair", i, w["iter"], g["iter"], g["status"])
print(f"{ROUNDS * DEPTH} launches of {N} pairs, {helped} pairs helped, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
