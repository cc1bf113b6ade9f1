#!/usr/bin/env python3
"""bench.py -- CVO frame-pair alignments/s on MI355X (BASELINE.json metric).

A "step" aligns one batch of 64 independent synthetic 640x480 RGB-D frame pairs
(~3000 points per cloud, TUM fr1 intrinsics) per GPU through the C ABI's batched
path: one persistent launch of cvo_align_kernel, clouds already resident in HBM,
R=I, T=0, ell=0.15 at the start of every step (fresh-object semantics).  Eight
steps are in flight on separate HIP streams (BASELINE config 3: "HIP streams"),
one workgroup per pair: a launch covers 64 of the 256 CUs, its pairs have
data-dependent iteration counts, the other launches fill what it leaves idle.  At N GPUs
every rank owns its own 64 pairs (weak scaling: BASELINE config 3 at N=1, config 4
= 512 pairs at N=8); the only collective is an RCCL all-gather of the 64-byte
result records.  Rank 0 prints one JSON line.

Launch: python bench.py --gpus 1            (driver: torch.distributed.run for N>1)
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per step in flight (the HIP runtime maps streams onto 4 queues by default; launches that share a
# queue run one after the other).  Read by the runtime when it initialises, i.e. after this line.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PAIRS_PER_GPU = 64
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_VALU_PEAK_TF = 157.3      # MI355X_MICROARCH.md: peak FP32 vector


_SHAPE = "tum"


def _gen_pair(idx):
    from cvo_slam_amd import synth
    p = synth.make_pair(idx, cam=synth.ETH3D if _SHAPE == "eth3d" else synth.TUM1)
    return idx, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat


def generate_pairs(first: int, count: int, workers: int = 0):
    """Seeded synthetic pairs first..first+count-1, rendered on host worker processes
    (forked before anything touches the GPU).  workers = 1: in this process (under rocprofv3 --pmc the profiler has
    initialised the GPU before the program starts, and forking after that hangs now and then)."""
    workers = max(1, min(workers or host_threads(), count))
    if workers == 1:
        return [_gen_pair(first + i) for i in range(count)]
    ctx = mp.get_context("fork")
    with ctx.Pool(workers) as pool:
        out = pool.map(_gen_pair, [first + i for i in range(count)])
    return out


def alg_bytes_iter(nf, nm):   # SURVEY 8d / BASELINE.md section 3
    return 2 * (32 * nf + 32 * nm) + 2 * 12 * nm


def alg_flops_iter(nf, nm):
    return 2 * nf * nm * 8 + 18 * nm


def rot_trans_err(A, B):
    A = np.asarray(A, np.float64).reshape(3, 4); B = np.asarray(B, np.float64).reshape(3, 4)
    D = A[:, :3].T @ B[:, :3]
    w = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    ang = float(np.arctan2(np.linalg.norm(w), (np.trace(D) - 1.0) / 2.0))     # robust near 0, unlike arccos of the trace
    return ang, float(np.linalg.norm(A[:, 3] - B[:, 3]))


def host_threads() -> int:
    """Threads for the CPU baseline: the box's CPU share (16 per GPU on the bench pool),
    never the host's full core count (oversubscribed OpenMP teams crawl)."""
    if os.environ.get("CVO_BENCH_CPU_THREADS"):
        return max(1, int(os.environ["CVO_BENCH_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(pairs, threads, budget_s=20.0):
    """Oracle (CPU restatement with the reference's structure: KD-tree rebuilt every
    iteration, row-parallel loops) on the host cores.  Pass 1 aligns as many of the batch's
    pairs as fit in ~budget_s seconds once each (their transforms are the parity check);
    the first 16 pairs are then repeated four more times (rate = median of the five passes over
    those 16) and the first 3 pairs once on a single thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    po.build()

    def run(sel, nthreads):
        tfs, iters = [], []
        t0 = time.perf_counter()
        for (_, fx, ff, mx, mf) in sel:
            o = po.OracleCvo(search=po.SEARCH_KDTREE, threads=nthreads)
            o.set_pcd(fx, ff); o.set_pcd(mx, mf)
            o.align()
            st = o.get_state()
            tfs.append(st["transform"].copy()); iters.append(st["iter"] + 1)
            if time.perf_counter() - t0 > budget_s:
                break
        return tfs, iters, time.perf_counter() - t0

    sub = pairs[: min(16, len(pairs))]
    t_all0 = time.perf_counter()
    tfs_s, it_s, dt_s = run(sub, threads)                               # pass 1, first 16
    tfs_r, it_r, dt_r = run(pairs[len(sub):], threads) if len(pairs) > len(sub) else ([], [], 0.0)
    rates = [len(tfs_s) / dt_s]
    for _ in range(4):
        if time.perf_counter() - t_all0 > 1.5 * budget_s:
            break
        r_tfs, _, r_dt = run(sub, threads)
        rates.append(len(r_tfs) / r_dt)
    one = pairs[: min(3, len(pairs))]
    o_tfs, _, o_dt = run(one, 1)
    return dict(rate=float(np.median(rates)), rates=rates, first_pass_rate=(len(tfs_s) + len(tfs_r)) / (dt_s + dt_r), seconds=time.perf_counter() - t_all0,
                tfs=tfs_s + tfs_r, iters=it_s + it_r, single_thread_rate=len(o_tfs) / o_dt, single_thread_pairs=len(o_tfs))


def latency_probe(ca, pairs, device):
    """BASELINE config 2 and SURVEY 8f next-2, for the record (not part of `value`): one pair alone through a cvo::cvo-style
    object (align, then the tracker's score block, cvo.cpp:475-503), and ten loop-closure candidates through the batch API
    (one align launch + one launch for all 10 x (6 inner products + 2 Hessians), keyframe_graph.cpp:693-717)."""
    _, fx, ff, mx, mf = pairs[0]
    al, sc = [], []
    for _ in range(5):
        g = ca.Cvo(device=device)
        g.set_pcd(fx, ff); g.set_pcd(mx, mf)
        t0 = time.perf_counter(); g.align(); al.append(time.perf_counter() - t0)
        tf = g.transform
        t0 = time.perf_counter(); g.compute_innerproduct(tf); sc.append(time.perf_counter() - t0)
        g.close()
    n = min(10, len(pairs))
    B = ca.CvoBatch(n, device=device)
    for i in range(n):
        B.set_pair(i, pairs[i][1], pairs[i][2], pairs[i][3], pairs[i][4])
    eye = np.tile(np.eye(3, 4, dtype=np.float32), (n, 1, 1))
    la, ls = [], []
    for _ in range(5):
        B.reset_states()
        t0 = time.perf_counter(); B.align(n); la.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); B.compute_innerproduct_lc(eye, eye, eye); ls.append(time.perf_counter() - t0)
    B.close()
    # the point-cloud generator (cvo.set_pcd on the images): GPU vs the oracle's CPU restatement
    from cvo_slam_amd import synth
    (fa, da), _, _ = synth.make_frames(0)
    camt = synth.camera_tuple(synth.TUM1)
    g = ca.Cvo(device=device)
    pg = []
    for _ in range(7):
        t0 = time.perf_counter(); g.set_pcd_images(fa, da, camt); pg.append(time.perf_counter() - t0)
    n_pts = g.get_fixed_and_moving_number()[1] or g.get_cloud(0)[0].shape[0]
    g.close()
    # one live tracking step from images, the reference's per-frame sequence (local_tracker.cpp:356-431): odometry object
    # match_odometry(frame), keyframe object match_keyframe(frame) + compute_innerproduct; fresh objects per repetition
    (_, _), (fb, db), _ = synth.make_frames(0)
    tr = []
    for _ in range(5):
        odo, kf = ca.Cvo(device=device), ca.Cvo(device=device)
        odo.set_pcd_images(fa, da, camt); kf.set_pcd_images(fa, da, camt)
        t0 = time.perf_counter()
        odo.match_odometry_images(fb, db, camt)
        tfk = kf.match_keyframe_images(fb, db, camt)
        kf.compute_innerproduct(tfk.astype(np.float32))
        tr.append(time.perf_counter() - t0)
        odo.close(); kf.close()
    pc = []
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        po.build()
        for _ in range(5):
            t0 = time.perf_counter(); po.pcd_generate(fa, da, camt); pc.append(time.perf_counter() - t0)
    except Exception:
        pc = [float("nan")]
    med = lambda v: 1e3 * float(np.median(v))
    img_bytes = fa.size + 2 * da.size
    return {"single_pair_align_ms": med(al), "single_pair_score_block_ms": med(sc), "lc_candidates": n,
            "set_pcd_images_ms": med(pg[2:]), "set_pcd_images_points": int(n_pts), "set_pcd_images_input_MB": img_bytes / 1e6,
            "set_pcd_images_cpu_port_ms": med(pc), "tracker_frame_from_images_ms": med(tr),
            "lc_batch_align_ms": med(la), "lc_batch_score_block_ms": med(ls), "note": "host wall time per call, median of 5, automatic workgroup count"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="frame pairs per GPU per step")
    ap.add_argument("--workgroups", type=int, default=1, help="workgroups per pair (0 = auto: lowest latency of one batch alone; 1 = highest throughput)")
    ap.add_argument("--streams", type=int, default=8, help="steps in flight (batch objects on separate HIP streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shape", choices=("tum", "eth3d"), default="tum", help="tum: 640x480, ~3 k points per cloud (the metric's configuration); eth3d: 736x456, ~9.3 k points (BASELINE config 5)")
    ap.add_argument("--gen-workers", type=int, default=0, help="host processes rendering the synthetic pairs (0 = one per host thread, 1 = no fork)")
    ap.add_argument("--no-latency-probe", action="store_true", help="skip the single-pair / loop-closure / point-cloud latency measurements (counter passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    global _SHAPE
    _SHAPE = args.shape
    # host-side input generation first (forks; no GPU state yet)
    pairs = generate_pairs(rank * args.pairs, args.pairs, args.gen_workers)
    if rank == 0:
        print(f"[bench] generated {len(pairs)} pairs per rank; starting GPU work", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    import cvo_slam_amd as ca

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    backend = os.environ.get("CVO_BENCH_BACKEND", "nccl")          # "gloo" + CVO_BENCH_SHARE_GPU=1: rehearsal of the N>1 code path on a one-GPU box
    if os.environ.get("CVO_BENCH_SHARE_GPU"):
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # `depth` batch objects, each on its own HIP stream, hold the same pairs; consecutive steps go to
    # alternating objects so that the next step's persistent kernel fills the CUs the previous step's
    # last, longest pairs have already left (alignments have data-dependent iteration counts).
    depth = max(1, args.streams)
    batches = []
    for _ in range(depth):
        b = ca.CvoBatch(args.pairs, device=local_rank)
        b.set_workgroups(args.workgroups)
        for i, (_, fx, ff, mx, mf) in enumerate(pairs):
            b.set_pair(i, fx, ff, mx, mf)
        batches.append(b)
    batch = batches[0]

    from cvo_slam_amd import shard
    n = args.pairs
    assert list(shard.shard_range(world * n, rank, world)) == list(range(rank * n, rank * n + n))
    sends = [torch.zeros((n, shard.RESULT_FLOATS), dtype=torch.float32, device="cuda") for _ in range(depth)]
    gathered = None
    inflight = []                                  # batch objects with a launch not yet waited for

    kernel_ms = []                                 # HIP-event duration of every launch, on the stream it ran on

    def finish(bi):
        nonlocal gathered
        batches[bi].wait()
        kernel_ms.append(batches[bi].last_launch()["kernel_ms"])
        if world > 1:
            gathered = shard.gather_results(sends[bi], world * n, world)   # RCCL: the SE(3) results of every rank, everywhere

    def step(i):
        bi = i % depth
        if bi in inflight:                         # the object is reused: its previous step must be complete first
            inflight.remove(bi); finish(bi)
        b = batches[bi]
        b.reset_states()                           # every step starts from R=I, T=0, ell=0.15
        b.align_async(n)
        b.results_to_device(sends[bi].data_ptr(), n)   # same stream, behind the kernel
        inflight.append(bi)

    def drain():
        while inflight:
            finish(inflight.pop(0))

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(depth):                         # setup: every batch object's first launch uploads its descriptors
        step(i)
    drain()
    for i in range(args.warmup):
        step(i)
    drain()
    # un-pipelined latency of one step (one batch alone on the GPU), for the record
    sync_all()
    t1 = time.perf_counter(); step(0); drain(); torch.cuda.synchronize(); single_step_ms = 1e3 * (time.perf_counter() - t1)
    single_kernel_ms = batches[0].last_launch()["kernel_ms"]
    sync_all()
    kernel_ms.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    sync_all()
    elapsed = time.perf_counter() - t0
    assert len(kernel_ms) == args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    results = batch.wait(n)
    info = batch.last_launch()

    # Secondary figure ("alignments incl. the post-align score block"): every step also queues the tracker's score
    # block (4 inner products + 1 Hessian per pair, cvo.cpp:475-503) behind its align launch and collects it with the results.
    with_scores = None
    if world == 1 and not args.no_latency_probe:
        k2 = max(depth, min(args.steps, 32))
        scored = []

        def step_scored(i):
            bi = i % depth
            if bi in scored:
                scored.remove(bi); batches[bi].wait(); batches[bi].innerproduct_results(n)
            b = batches[bi]
            b.reset_states(); b.align_async(n); b.enqueue_innerproduct(n)
            scored.append(bi)

        def drain_scored():
            last = None
            while scored:
                bi = scored.pop(0); batches[bi].wait(); last = batches[bi].innerproduct_results(n)
            return last

        for i in range(depth):
            step_scored(i)
        drain_scored(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(k2):
            step_scored(i)
        last_scores = drain_scored(); torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        with_scores = {"value": n * k2 / el2, "unit": "alignments/s", "steps": k2, "ms_per_step": 1e3 * el2 / k2,
                       "score_block": "compute_innerproduct per pair (4 inner products + 1 Hessian, dense all-pairs sweeps at the ell align() left behind), "
                                      "one launch per step queued behind the align launch",
                       "mean_cos_angle": float(np.mean([r["cos_angle"] for r in last_scores]))}
        batch.reset_states(); batch.align_async(n); batch.wait()       # leave batch 0 as the timed region left it

    # PCIe-inclusive rate, for the record (never `value`): every step first hands its 64 pairs over as host buffers in the
    # reference layout (cvo_batch_set_pair: packing + host-to-device copies), then aligns them.
    with_upload = None
    if world == 1 and not args.no_latency_probe:
        k3 = max(depth, min(args.steps, 32))
        busy = []

        def step_upload(i):
            bi = i % depth
            if bi in busy:
                busy.remove(bi); batches[bi].wait()
            b = batches[bi]
            for q, (_, fx, ff, mx, mf) in enumerate(pairs):
                b.set_pair(q, fx, ff, mx, mf)
            b.align_async(n)
            busy.append(bi)

        for i in range(depth):
            step_upload(i)
        while busy:
            batches[busy.pop(0)].wait()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for i in range(k3):
            step_upload(i)
        while busy:
            batches[busy.pop(0)].wait()
        torch.cuda.synchronize()
        el3 = time.perf_counter() - t3
        mb = sum(fx.nbytes + ff.nbytes + mx.nbytes + mf.nbytes for (_, fx, ff, mx, mf) in pairs) / 1e6
        with_upload = {"value": n * k3 / el3, "unit": "alignments/s", "steps": k3, "ms_per_step": 1e3 * el3 / k3, "host_MB_per_step": mb,
                       "note": "clouds cross the boundary as host buffers every step (one host thread packs and copies them)"}
        batch.reset_states(); batch.align_async(n); batch.wait()
    if rank == 0 and os.environ.get("CVO_BENCH_PHASES"):
        ph = batch.last_phase_seconds(); its_all = sum(r["iterations_run"] for r in results)
        print("[bench] phase us/iteration under load (workgroup 0 of every pair of the last launch): " +
              str({k: round(1e6 * v / its_all, 1) for k, v in ph.items()}), file=sys.stderr, flush=True)
    bad = [r["status"] for r in results if r["status"] != 0]
    if bad:
        raise SystemExit(f"align kernel reported errors: {bad}")

    if rank == 0:
        its = [r["iterations_run"] for r in results]
        nfs = [p[1].shape[0] for p in pairs]; nms = [p[3].shape[0] for p in pairs]
        bytes_launch = float(sum(it * alg_bytes_iter(a, b) for it, a, b in zip(its, nfs, nms)))
        flops_launch = float(sum(it * alg_flops_iter(a, b) for it, a, b in zip(its, nfs, nms)))
        k_ms = float(np.mean(kernel_ms))
        achieved_gbs = bytes_launch / (k_ms * 1e-3) / 1e9
        achieved_tf = flops_launch / (k_ms * 1e-3) / 1e12
        value = world * n * args.steps / elapsed
        step_ms_rank = 1e3 * elapsed / args.steps          # one launch retires every step_ms_rank on this GPU
        overlap = k_ms / step_ms_rank                      # launches running side by side, on average
        traffic = None                                  # HBM bytes per launch from separate rocprofv3 --pmc passes (scripts/pmc_run.sh)
        valu_instr = None                               # VALU wave-instructions one launch executes (same passes)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            traffic = float(pmc["hbm_bytes_per_launch"]); valu_instr = float(pmc.get("valu_wave_instructions_per_launch", 0)) or None
        except Exception:
            traffic = None
        out = {
            "metric": "CVO frame-pair alignments/sec (640x480, ~3k pts/cloud)",
            "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} independent synthetic {'640x480 TUM' if args.shape == 'tum' else '736x456 ETH3D'}-shape RGB-D pairs per GPU per step "
                                   f"(BASELINE config 3; {world * n} pairs per step at {world} GPU(s); config 4 = 512 pairs at 8 GPUs), "
                                   f"full align() from R=I,T=0,ell=0.15 to convergence",
                       "pairs_per_gpu": n, "points_fixed_mean": float(np.mean(nfs)), "points_moving_mean": float(np.mean(nms)),
                       "iterations_mean": float(np.mean(its)), "iterations_max": int(np.max(its)),
                       "workgroups_per_pair": args.workgroups or "auto", "steps_in_flight": depth,
                       "single_step_ms_unpipelined": single_step_ms, "single_kernel_ms_unpipelined": single_kernel_ms, "collective": (("RCCL" if backend == "nccl" else backend) + " all_gather of 64-byte result records") if world > 1 else "none (1 GPU)"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "cvo_align_kernel", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "launches_side_by_side": overlap, "achieved_all_launches": achieved_gbs * overlap,
                         "frac_all_launches": achieved_gbs * overlap / HBM_PEAK_GBS,
                         "profile": "profiles/r01_j_kernel_stats.csv: rocprofv3 --kernel-trace --stats of this command with --no-latency-probe (only "
                                    "launches of the timed regime): 11.90 ms average over 297 launches; profiles/README.md indexes the rest",
                         "note": "achieved = algorithmic bytes of ONE launch / its own HIP-event duration; a launch holds 64 of 256 CUs and "
                                 "launches_side_by_side of them run at once, achieved_all_launches is the whole GPU's rate.  The path is "
                                 "latency/VALU bound, not HBM bound (SURVEY 8d): see valu and DESIGN.md"},
            "with_score_block": with_scores,
            "with_host_upload": with_upload,
            "valu": {"dense_pair_tests_per_s": flops_launch / 8.0 / (step_ms_rank * 1e-3),
                     "algorithmic_flops_fraction": flops_launch / (step_ms_rank * 1e-3) / 157.3e12,
                     "hbm_fraction_of_achievable": bytes_launch / (step_ms_rank * 1e-3) / 6.3e12,
                     "executed_wave_instructions_per_launch": valu_instr,
                     "issue_slots_used": (valu_instr * 4.0 / (256 * 4 * 2.4e9 * step_ms_rank * 1e-3)) if valu_instr else None,
                     "note": "issue_slots_used = VALU wave-instructions of one launch (SQ_INSTS_VALU, profiles/) x 4 cycles / (256 CUs x 4 SIMDs x 2.4 GHz x "
                             "time per step): the share of the chip's plain-f32 issue slots the job keeps busy.  dense_pair_tests_per_s counts the "
                             "N*M tests per iteration the reference's radius search stands for; the kernel skips most of them (lists + box cull), so "
                             "algorithmic_flops_fraction (SURVEY 8d: 8 flop per dense pair test / 157.3 TF) exceeds 1 -- it prices work that is not executed; "
                             "hbm_fraction_of_achievable = algorithmic bytes per step / 6.3 TB/s (SURVEY 8d), whole GPU"},
        }
        if world == 1 and not args.no_latency_probe:
            out["latency"] = latency_probe(ca, pairs, local_rank)
        if not args.no_cpu_baseline and world == 1:
            cores = host_threads()
            print(f"[bench] GPU done ({value:.1f} alignments/s); timing the CPU baseline on {cores} threads ...", file=sys.stderr, flush=True)
            cb = cpu_baseline(pairs, cores)
            cpu_rate, cpu_tfs, cpu_its = cb["rate"], cb["tfs"], cb["iters"]
            errs = [rot_trans_err(results[i]["transform"], cpu_tfs[i]) for i in range(len(cpu_tfs))]
            out["cpu_baseline"] = {"value": cpu_rate, "unit": "alignments/s", "cores": cores, "kind": "port",
                                   "sample": f"oracle (CPU restatement of the reference: KD-tree rebuilt per iteration, OpenMP rows; not the icpc binary) on "
                                             f"{cores} host threads: {len(cpu_tfs)} pairs of the timed batch once each (parity check, {cb['first_pass_rate']:.1f}/s), "
                                             f"value = median of {len(cb['rates'])} passes over the first 16 pairs; {cb['seconds']:.1f} s in all",
                                   "passes": cb["rates"], "single_thread_value": cb["single_thread_rate"], "single_thread_pairs": cb["single_thread_pairs"],
                                   "iterations_mean": float(np.mean(cpu_its))}
            out["parity"] = {"pairs_checked": len(errs), "max_rot_err_rad": max(e[0] for e in errs), "max_trans_err_m": max(e[1] for e in errs),
                             "iterations_equal": bool(all(a == b for a, b in zip(its, cpu_its))), "tolerance": "1e-4 rad / 1e-4 m"}
            out["speedup_vs_cpu_baseline"] = value / cpu_rate
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
