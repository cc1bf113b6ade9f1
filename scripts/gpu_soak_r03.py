"""development aid: soak of the round-3 step -- adoption, score block in the tail, hand-over every step, several batches in flight, reuse of whichever
finished -- every launch's poses against a reference launch without any of it, every score block against the score-kernel path"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
ca.load_library()
N = int(os.environ.get("PAIRS", "64")); STEPS = int(os.environ.get("STEPS", "400")); DEPTH = int(os.environ.get("DEPTH", "8"))
pairs = [synth.make_pair(300 + i) for i in range(N)]
clouds = [(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs]
prep = ca.CvoBatch.prepare_pairs(clouds)
ref = ca.CvoBatch(N); ref.set_workgroups(1); ref.set_pairs(prep)
ref.align_async(N); ref.enqueue_innerproduct(N); want = ref.wait(N); want_sc = ref.innerproduct_results(N); ref.close()
bs = []
for _ in range(DEPTH):
    b = ca.CvoBatch(N); b.set_workgroups(1); b.set_adoption(True); b.set_tail_scores(True); b.set_pairs(prep); bs.append(b)
bad = helped = tails = 0; t0 = time.time(); inflight = []
def check(b):
    global bad, helped, tails
    got = b.wait(N); helped += b.last_adoptions(); m = b.last_tail_answers(N); tails += sum(1 for x in m if x == 31); sc = b.innerproduct_results(N)
    for i, (w, g, ws, gs) in enumerate(zip(want, got, want_sc, sc)):
        ok = g["status"] == 0 and g["iter"] == w["iter"] and np.array_equal(g["transform"], w["transform"])
        ok = ok and gs["inn_post"][1] == ws["inn_post"][1] and gs["inn_pre"][1] == ws["inn_pre"][1] and gs["inliers"] == ws["inliers"] and abs(gs["inn_post"][0] - ws["inn_post"][0]) <= 1e-6 * abs(ws["inn_post"][0])
        if not ok:
            bad += 1; print("MISMATCH pair", i, w["iter"], g["iter"], g["status"], gs["inn_post"], ws["inn_post"])
for step in range(STEPS):
    k = next((j for j in range(DEPTH) if j not in inflight), None)
    if k is None:
        k = next((j for j in inflight if bs[j].done()), inflight[0])
        inflight.remove(k); check(bs[k])
    b = bs[k]
    if step % 3 == 0: b.set_pairs(prep)          # hand-over: packed by the launch (tails on: by the pack kernel first)
    else: b.reset_states()
    b.align_async(N); inflight.append(k)
while inflight: check(bs[inflight.pop(0)])
print(f"{STEPS} launches of {N} pairs, {helped} pairs helped, {tails} score blocks fully answered in the tail, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
