#!/usr/bin/env python3
"""The driver's command, pair by pair on the device's own clock: K steps of 64 pairs, `--streams` batch objects in flight, the next step to whichever launch has
completed (bench.py's loop), every pair's (start, end) read back from the pair states (cvo_batch_last_pair_spans).  Prints, for the timed steps: when each launch's
first and last workgroup started and when it ended, how many compute units the pairs' own workgroups kept busy over time, the idle CU-time of the job and the
lower bounds (work / 256, the longest pair behind the last start).  usage: gpu_timeline.py [--steps 20] [--warmup 5] [--streams 8] [--no-adoption] [--out file.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5); ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--no-adoption", action="store_true"); ap.add_argument("--out", default=""); ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=1)
    args = ap.parse_args()
    pairs = bench.generate_pairs(0, args.pairs, 0)
    import torch
    import cvo_slam_amd as ca
    torch.cuda.set_device(0)
    n = args.pairs
    batches = []
    for _ in range(args.streams):
        b = ca.CvoBatch(n, device=0); b.set_workgroups(1); b.set_adoption(not args.no_adoption); batches.append(b)
    prepared = ca.CvoBatch.prepare_pairs([(fx, ff, mx, mf) for (_, fx, ff, mx, mf) in pairs])
    for b in batches:
        b.set_pairs(prepared)
    inflight = []; log = []; held = {}

    def finish(bi):
        it = [r["iterations_run"] for r in batches[bi].wait(n)]
        t0, t1, j = batches[bi].last_pair_spans(n)
        log.append({"step": held[bi][0], "submit_host": held[bi][1], "start": t0.tolist(), "end": t1.tolist(), "joined_at": j.tolist(), "iterations": it})

    def pick():
        for k in range(args.streams):
            if k not in inflight:
                return k
        t_poll = time.perf_counter()
        while time.perf_counter() - t_poll < 0.05:
            for k in inflight:
                if batches[k].done():
                    return k
        return inflight[0]

    def step(i):
        bi = pick()
        if bi in inflight:
            inflight.remove(bi); finish(bi)
        b = batches[bi]; b.reset_states(); held[bi] = (i, time.perf_counter()); b.align_async(n); inflight.append(bi)

    def drain():
        while inflight:
            finish(inflight.pop(0))

    for i in range(args.streams):
        step(-1000 + i)
    drain()
    for rep in range(args.repeat):
        for i in range(args.warmup):
            step(-100 + i)
        drain(); torch.cuda.synchronize(); log.clear()
        t_a = time.perf_counter()
        for i in range(args.steps):
            step(i)
        drain(); torch.cuda.synchronize()
        wall = time.perf_counter() - t_a
        L = sorted(log, key=lambda r: r["step"])
        T0 = min(min(r["start"]) for r in L); sub0 = min(r["submit_host"] for r in L)
        S = np.array([r["start"] for r in L]) - T0; E = np.array([r["end"] for r in L]) - T0
        J = np.array([r["joined_at"] for r in L])
        dur = E - S
        Tend = E.max()
        print(f"repeat {rep}: {args.steps} steps x {n} pairs, {args.streams} in flight, adoption {'off' if args.no_adoption else 'on'}: host wall {wall * 1e3:.2f} ms = {args.steps * n / wall:.0f} alignments/s; "
              f"device span {Tend * 1e3:.2f} ms; pair time mean {dur.mean() * 1e3:.2f} max {dur.max() * 1e3:.2f} min {dur.min() * 1e3:.2f} ms; sum {dur.sum() * 1e3:.0f} CU-ms -> / 256 = {dur.sum() / 256 * 1e3:.2f} ms")
        print("step  submitted(host)  first start  last start   end    lives  helped  longest pair")
        for k, r in enumerate(L):
            print(f"{r['step']:4d}  {1e3 * (r['submit_host'] - sub0):9.2f}  {S[k].min() * 1e3:11.2f}  {S[k].max() * 1e3:10.2f}  {E[k].max() * 1e3:6.2f}  {1e3 * (E[k].max() - S[k].min()):6.2f}  {int((J[k] > 0).sum()):5d}  {dur[k].max() * 1e3:8.2f}")
        # compute units held by the pairs' own workgroups over time (helpers are not in the pair states)
        grid = np.linspace(0, Tend, 53)
        busy = [(int(((S <= t) & (E > t)).sum())) for t in grid]
        print("owners' workgroups alive at t (ms): " + " ".join(f"{t * 1e3:.1f}:{b}" for t, b in zip(grid, busy)))
        idle = 256 * Tend - dur.sum()
        print(f"idle CU-time (owners only) {idle * 1e3:.0f} CU-ms of {256 * Tend * 1e3:.0f}; last start at {S.max() * 1e3:.2f} ms; pairs per position: mean duration by position (ms): "
              + " ".join(f"{d * 1e3:.1f}" for d in dur.mean(axis=0)[:16]) + " ...")
        if args.out:
            json.dump({"steps": args.steps, "streams": args.streams, "wall_s": wall, "start": S.tolist(), "end": E.tolist(), "joined_at": J.tolist()}, open(args.out if args.repeat == 1 else f"{args.out}.{rep}", "w"))


if __name__ == "__main__":
    main()
