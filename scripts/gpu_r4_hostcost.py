#!/usr/bin/env python3
"""What the host spends per step in bench.py's loop (8 batch objects in flight, 64 pairs): wait (results of a finished launch), reset_states, align_async -- wall time per call, us."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import bench
pairs = bench.generate_pairs(0, 64, 0)
import torch
import cvo_slam_amd as ca
torch.cuda.set_device(0)
n = 64; depth = 8
batches = []
for _ in range(depth):
    b = ca.CvoBatch(n, device=0); b.set_workgroups(1); b.set_adoption(True); batches.append(b)
prepared = ca.CvoBatch.prepare_pairs([(fx, ff, mx, mf) for (_, fx, ff, mx, mf) in pairs])
for b in batches:
    b.set_pairs(prepared)
T = {"done_poll": [], "wait": [], "reset": [], "launch": []}
inflight = []
def step(i):
    bi = None
    for k in range(depth):
        if k not in inflight: bi = k; break
    if bi is None:
        t0 = time.perf_counter()
        while bi is None:
            for k in inflight:
                if batches[k].done(): bi = k; break
        T["done_poll"].append(time.perf_counter() - t0)
        inflight.remove(bi)
        t0 = time.perf_counter(); batches[bi].wait(); batches[bi].last_launch(); T["wait"].append(time.perf_counter() - t0)
    b = batches[bi]
    t0 = time.perf_counter(); b.reset_states(); t1 = time.perf_counter(); b.align_async(n); t2 = time.perf_counter()
    T["reset"].append(t1 - t0); T["launch"].append(t2 - t1)
    inflight.append(bi)
for i in range(40): step(i)
for k in T: T[k].clear()
t0 = time.perf_counter()
for i in range(200): step(i)
while inflight: batches[inflight.pop(0)].wait()
el = time.perf_counter() - t0
print(f"200 steps: {1e3 * el / 200:.3f} ms per step; host us per call (median / mean): " + ", ".join(f"{k} {1e6 * np.median(v):.1f} / {1e6 * np.mean(v):.1f}" for k, v in T.items()))
# raw C calls without the Python result objects
import ctypes as C
L = batches[0].L
t = []
for _ in range(20):
    t0 = time.perf_counter(); L.cvo_batch_reset_states(batches[0].h); t1 = time.perf_counter(); L.cvo_batch_align_async(batches[0].h, n, None); t2 = time.perf_counter(); batches[0].wait(n)
    t.append((t1 - t0, t2 - t1))
print("one object alone, C calls: reset_states %.1f us, align_async %.1f us" % (1e6 * np.median([a for a, _ in t]), 1e6 * np.median([b for _, b in t])))
