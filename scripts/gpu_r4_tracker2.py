#!/usr/bin/env python3
"""The tracker's per-frame sequence on two handles (local_tracker.cpp:356-431), piece by piece (host wall, median of 7 fresh handle pairs): cloud generation, alignment, score block,
with and without the queued score block (cvo_set_tail_scores); and what hipMalloc / hipFree of a cloud-sized block cost on this box."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cvo_slam_amd as ca
from cvo_slam_amd import synth
(fa, da), (fb, db), _ = synth.make_frames(0)
camt = synth.camera_tuple(synth.TUM1)
med = lambda v: 1e3 * float(np.median(v))
for tail in (False, True):
    T = {k: [] for k in ("odo_pcd", "odo_align", "odo_score", "kf_pcd", "kf_align", "kf_score", "frame")}
    shared = 0
    for rep in range(7):
        odo, kf = ca.Cvo(device=0), ca.Cvo(device=0)
        odo.set_tail_scores(tail); kf.set_tail_scores(tail)
        odo.set_pcd_images(fa, da, camt); kf.set_pcd_images(fa, da, camt)
        t = [time.perf_counter()]
        odo.set_pcd_images(fb, db, camt); t.append(time.perf_counter())
        odo.align(); tfo = odo.transform; t.append(time.perf_counter())
        odo.compute_innerproduct(np.asarray(tfo, np.float32)); t.append(time.perf_counter())
        kf.set_pcd_images(fb, db, camt); t.append(time.perf_counter())
        kf.align(); tfk = kf.transform; t.append(time.perf_counter())
        kf.compute_innerproduct(np.asarray(tfk, np.float32)); t.append(time.perf_counter())
        for k, name in enumerate(("odo_pcd", "odo_align", "odo_score", "kf_pcd", "kf_align", "kf_score")):
            T[name].append(t[k + 1] - t[k])
        T["frame"].append(t[-1] - t[0])
        shared = kf.shared_cloud_count()
        odo.close(); kf.close()
    print(f"queued score block {'on ' if tail else 'off'}: " + ", ".join(f"{k} {med(v):.3f}" for k, v in T.items()) + f" ms; clouds the keyframe object took over: {shared}; iterations {odo.get_iteration_number() if False else ''}", flush=True)
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
for size in (100 << 10, 1 << 20, 16 << 20):
    tm, tf = [], []
    for _ in range(20):
        t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), C.c_size_t(size)); t1 = time.perf_counter(); hip.hipFree(p); t2 = time.perf_counter()
        tm.append(t1 - t0); tf.append(t2 - t1)
    print(f"hipMalloc({size >> 10} KiB) {1e6 * np.median(tm):.1f} us, hipFree {1e6 * np.median(tf):.1f} us")
