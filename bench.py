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

Launch: python bench.py --gpus N            N > 1 without RANK in the environment: this process starts the N ranks itself (child
                                            processes, one per GPU, rendezvous on 127.0.0.1) and passes rank 0's line through;
                                            under torch.distributed.run (RANK set) it is one of the ranks.
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
ADOPTION_DEFAULT = True        # cvo_batch_set_adoption in the timed loops unless --no-adoption (tests/test_gpu_config3.py runs both modes against the oracle)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_VALU_PEAK_TF = 157.3      # MI355X_MICROARCH.md: peak FP32 vector


_SHAPE = "tum"


def _gen_pair(idx):
    from cvo_slam_amd import synth
    kw = {}
    if os.environ.get("CVO_BENCH_MOTION"):          # experiments: another inter-frame motion "max_deg,max_trans_m" (the metric's set: 2 degrees, 3 cm -- SURVEY 8d)
        deg, tr = os.environ["CVO_BENCH_MOTION"].split(","); kw = dict(max_deg=float(deg), max_trans=float(tr))
    p = synth.make_pair(idx, cam=synth.ETH3D if _SHAPE == "eth3d" else synth.TUM1, **kw)
    return idx, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat


def under_profiler() -> bool:
    """rocprofv3 preloads its tool library into the program it starts; with counter collection that library has
    initialised the GPU before main() runs, and a fork() after that point hangs now and then."""
    pre = os.environ.get("LD_PRELOAD", "")
    return ("rocprof" in pre) or any(k.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_")) for k in os.environ)


def generate_pairs(first: int, count: int, workers: int = 0):
    """Seeded synthetic pairs first..first+count-1, rendered on host worker processes (forked before anything touches
    the GPU) -- or in this process when a profiler has been preloaded (no fork then: see under_profiler) or workers = 1."""
    if workers <= 0 and under_profiler():
        workers = 1
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
    """The oracle (CPU restatement with the reference's structure: KD-tree rebuilt every iteration, two sparse sweeps) on
    the host cores, three ways:
      parity pass   the un-fused -O2 build (the checker), every pair of the batch once: transforms for the parity block
      pair_parallel the -O3 -march=native -ffp-contract=fast build, one alignment per host thread,
                    `threads` of them at once: the strongest CPU configuration for a batch of independent pairs = `value`
      row_parallel  the same build with the reference's own structure: one alignment at a time, its row loops spread over
                    `threads` OpenMP threads (the TBB parallel_for stand-in); the KD-tree build and the CSR assembly of
                    every iteration stay serial, as in the reference (cvo.cpp:135-136, 182-183)
    plus one thread alone.  All on a bounded sample (about budget_s seconds in all)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    po.build()

    def one(args):
        (_, fx, ff, mx, mf), nthreads, flavor = args
        o = po.OracleCvo(search=po.SEARCH_KDTREE, threads=nthreads, flavor=flavor)
        o.set_pcd(fx, ff); o.set_pcd(mx, mf)
        o.align()
        st = o.get_state()
        return st["transform"].copy(), st["iter"] + 1, o.get_timing()

    t_all0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:                        # ctypes drops the GIL inside the oracle
        t0 = time.perf_counter()
        par = list(ex.map(one, [(p, 1, "parity") for p in pairs]))
        dt_parity = time.perf_counter() - t0
        rates_pp = []
        for _ in range(3):
            t0 = time.perf_counter()
            fast = list(ex.map(one, [(p, 1, "fast") for p in pairs]))
            rates_pp.append(len(pairs) / (time.perf_counter() - t0))
            if time.perf_counter() - t_all0 > budget_s:
                break
    sub = pairs[: min(8, len(pairs))]
    rates_rp, serial = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        rr = [one((p, threads, "fast")) for p in sub]
        dt = time.perf_counter() - t0
        rates_rp.append(len(sub) / dt)
        serial.append(sum(r[2]["kdtree_build"] + r[2]["csr_assembly"] for r in rr) / dt)
    t0 = time.perf_counter()
    single = [one((p, 1, "fast")) for p in pairs[:3]]
    rate_1 = len(single) / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    single_par = [one((p, 1, "parity")) for p in pairs[:3]]
    rate_1_parity = len(single_par) / (time.perf_counter() - t0)
    return dict(rate=float(np.median(rates_pp)), rates=rates_pp, parity_build_rate=len(pairs) / dt_parity,
                row_parallel_rate=float(np.median(rates_rp)), row_parallel_serial_fraction=float(np.median(serial)),
                single_thread_rate=rate_1, single_thread_rate_parity_build=rate_1_parity, seconds=time.perf_counter() - t_all0,
                tfs=[r[0] for r in par], iters=[r[1] for r in par],
                fast_vs_parity_build=max((rot_trans_err(a[0], b[0]) for a, b in zip(fast, par)), key=lambda e: max(e)))


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
    (fa, da), (fb0, db0), _ = synth.make_frames(0)
    camt = synth.camera_tuple(synth.TUM1)
    g = ca.Cvo(device=device)
    pg, pg_same = [], []
    for i in range(8):                                     # the two frames in turn: every call generates (a frame repeated on the same thread is taken over, below)
        fr = (fa, da) if i % 2 == 0 else (fb0, db0)
        t0 = time.perf_counter(); g.set_pcd_images(fr[0], fr[1], camt); pg.append(time.perf_counter() - t0)
    g.set_pcd_images(fa, da, camt)
    n_pts = g.get_fixed_and_moving_number()[1] or g.get_cloud(0)[0].shape[0]
    g2 = ca.Cvo(device=device)
    for _ in range(5):                                     # the frame the thread's last generation saw: compared byte for byte, the cloud copied on the device
        t0 = time.perf_counter(); g2.set_pcd_images(fa, da, camt); pg_same.append(time.perf_counter() - t0)
    g.close(); g2.close()
    # one live tracking step from images, the reference's per-frame sequence (local_tracker.cpp:356-431): odometry object
    # match_odometry(frame), keyframe object match_keyframe(frame) + compute_innerproduct; fresh objects per repetition
    (_, _), (fb, db), _ = synth.make_frames(0)
    tr, trf = [], []
    for full in (False, True):     # full: with the odometry object's score block too, as local_tracker.cpp:356-431 has it (the series up to round 3 timed the keyframe object's only)
        for _ in range(5):
            odo, kf = ca.Cvo(device=device), ca.Cvo(device=device)
            odo.set_tail_scores(True); kf.set_tail_scores(True)       # what the adaptor sets for the tracker's two objects: every alignment queues its score block behind itself
            odo.set_pcd_images(fa, da, camt); kf.set_pcd_images(fa, da, camt)
            t0 = time.perf_counter()
            tfo = odo.match_odometry_images(fb, db, camt)
            if full:
                odo.compute_innerproduct(tfo.astype(np.float32))
            tfk = kf.match_keyframe_images(fb, db, camt)
            kf.compute_innerproduct(tfk.astype(np.float32))
            (trf if full else tr).append(time.perf_counter() - t0)
            shared = kf.shared_cloud_count()                          # 2: the keyframe object took both frames' clouds over from the odometry object's generation (same images)
            odo.close(); kf.close()
    # the same frame with the caller one image ahead (cvo_stage_next_frame): frame t's cloud was staged while frame t - 1 was tracked, frame t + 1's is staged
    # between this frame's odometry block and its keyframe block -- what a pipelined image reader does with the reference's loop (run_SLAM.cpp:70-87)
    (fc, dc), _, _ = synth.make_frames(1)
    trs = []
    for _ in range(5):
        odo, kf = ca.Cvo(device=device), ca.Cvo(device=device)
        odo.set_tail_scores(True); kf.set_tail_scores(True)
        odo.set_pcd_images(fa, da, camt); kf.set_pcd_images(fa, da, camt)
        odo.stage_next_frame(fb, db, camt)
        time.sleep(0.003)                                             # (the previous frame's tracking: the staged generation runs beside it)
        t0 = time.perf_counter()
        tfo = odo.match_odometry_images(fb, db, camt)
        odo.stage_next_frame(fc, dc, camt)
        tfk = kf.match_keyframe_images(fb, db, camt)
        kf.compute_innerproduct(tfk.astype(np.float32))
        trs.append(time.perf_counter() - t0)
        staged = odo.staged_frame_count()
        odo.set_pcd_images(fc, dc, camt)                              # (takes the frame staged above: nothing is left running when the objects close)
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
            "set_pcd_images_ms": med(pg[2:]), "set_pcd_images_same_frame_ms": med(pg_same), "set_pcd_images_points": int(n_pts), "set_pcd_images_input_MB": img_bytes / 1e6,
            "set_pcd_images_cpu_port_ms": med(pc), "tracker_frame_from_images_ms": med(tr), "tracker_frame_with_both_score_blocks_ms": med(trf), "tracker_frame_clouds_taken_over": shared,
            "tracker_frame_next_frame_staged_ms": med(trs), "tracker_frame_staged_clouds_taken": staged,
            "lc_batch_align_ms": med(la), "lc_batch_score_block_ms": med(ls), "note": "host wall time per call, median of 5, automatic workgroup count"}


def launch_ranks(n: int, argv, collect_stdout: bool = False) -> int:
    """`bench.py --gpus N` invoked bare (no RANK in the environment): start the N ranks as CHILD processes of this one -- which has not
    touched the GPU and never will --, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way
    torch.distributed.run sets them.  Rank 0 writes the JSON line to this process's stdout; the other ranks' stdout goes to stderr.
    Returns the largest exit status; when a rank fails the others get ten seconds, then SIGTERM, then ten more and SIGKILL (by PID)."""
    import socket
    import subprocess
    import tempfile

    def start(port):
        ps, outs = [], []
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "GROUP_RANK": "0", "NNODES": "1",
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "CVO_BENCH_LAUNCHED_BY": "bench.py"})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # collected output goes to a temporary file, not a pipe: a rank that prints more than a pipe holds would block with nobody reading
            f = tempfile.TemporaryFile(mode="w+") if collect_stdout else None
            outs.append(f)
            ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=f if collect_stdout else (None if r == 0 else sys.stderr)))
        return ps, outs

    def free_port():
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); p = sock.getsockname()[1]; sock.close()
        return p

    rcs = [1] * n
    for attempt in range(3):                       # the port is only known to be free when it was asked for: another process may take it before rank 0 binds it
        port = free_port()
        procs, outs = start(port)
        print(f"[bench] started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}", file=sys.stderr, flush=True)
        rcs = [None] * n
        t_fail = t_term = None
        t_start = time.time()
        while any(rc is None for rc in rcs):
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    rc = pr.poll()
                    if rc is not None:
                        rcs[i] = rc
                        if rc != 0 and t_fail is None:
                            t_fail = time.time()
                            print(f"[bench] rank {i} exited with status {rc}", file=sys.stderr, flush=True)
            if t_fail is not None and t_term is None and time.time() - t_fail > 10.0:      # the others get ten seconds, then SIGTERM, then another ten, then SIGKILL
                t_term = time.time()
                for i, pr in enumerate(procs):
                    if rcs[i] is None:
                        pr.terminate()
            if t_term is not None and time.time() - t_term > 10.0:
                for i, pr in enumerate(procs):
                    if rcs[i] is None:
                        pr.kill(); rcs[i] = pr.wait() or 1
            time.sleep(0.05)
        if collect_stdout:
            for f in outs:
                f.seek(0); sys.stdout.write(f.read()); f.close()
            sys.stdout.flush()
        # a rendezvous that failed within seconds because the port was taken: once more on another port (anything else is the ranks' own failure)
        if all(rc == 0 for rc in rcs) or time.time() - t_start > 20.0 or os.environ.get("CVO_BENCH_NO_PORT_RETRY"):
            break
        print(f"[bench] the ranks failed within {time.time() - t_start:.0f} s (statuses {rcs}): trying another rendezvous port", file=sys.stderr, flush=True)
    return max(abs(rc) for rc in rcs)


def parity_check(pairs, results, its, threads):
    """Every pair of the timed batch through the oracle's un-fused parity build, one alignment per host thread: max pose error, iteration counts."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    po.build()

    def one(p):
        _, fx, ff, mx, mf = p
        o = po.OracleCvo(search=po.SEARCH_KDTREE, threads=1, flavor="parity")
        o.set_pcd(fx, ff); o.set_pcd(mx, mf); o.align()
        st = o.get_state()
        return st["transform"].copy(), st["iter"] + 1
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        cpu = list(ex.map(one, pairs))
    errs = [rot_trans_err(results[i]["transform"], cpu[i][0]) for i in range(len(cpu))]
    return {"pairs_checked": len(errs), "against": "oracle, un-fused parity build (every pair of the timed batch)", "max_rot_err_rad": max(e[0] for e in errs),
            "max_trans_err_m": max(e[1] for e in errs), "iterations_equal": bool(all(a == b[1] for a, b in zip(its, cpu))), "tolerance": "1e-4 rad / 1e-4 m",
            "oracle_seconds": time.perf_counter() - t0, "host_threads": threads}


def config5_child():
    """BASELINE config 5 (the largest single-GPU configuration: 736x456 ETH3D shape, ~9.3 k points per cloud) as a short run of this very script in a child process,
    after the main measurement has left the device: value, roofline and the parity of the 64 pairs it timed, for the driver's line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--shape", "eth3d", "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--parity-only", "--no-latency-probe"]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
        line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"child exited {r.returncode}", "stderr_tail": r.stderr[-400:]}
        d = json.loads(line[-1])
    except Exception as e:                                           # the main line must not depend on this object
        return {"error": repr(e)}
    rf, hb = d.get("roofline", {}), d.get("hbm", {})
    return {"metric": "CVO frame-pair alignments/sec (736x456, ~9.3k pts/cloud: BASELINE config 5)", "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
            "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"], "points_fixed_mean": d["config"]["points_fixed_mean"], "iterations_mean": d["config"]["iterations_mean"],
            "workgroups_per_pair": d["config"]["workgroups_per_pair"], "steps_in_flight": d["config"]["steps_in_flight"],
            "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "valu_wave_instructions_per_launch", "counters")},
            "traffic_over_algorithmic_bytes": (hb.get("traffic") / hb["algorithmic_bytes_per_launch"]) if hb.get("traffic") and hb.get("algorithmic_bytes_per_launch") else None,
            "parity": d.get("parity"), "child_seconds": time.perf_counter() - t0, "command": " ".join(cmd[1:])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="frame pairs per GPU per step")
    ap.add_argument("--workgroups", type=int, default=1, help="workgroups per pair (0 = auto: lowest latency of one batch alone; 1 = highest throughput)")
    ap.add_argument("--streams", type=int, default=8, help="steps in flight (batch objects on separate HIP streams)")
    ap.add_argument("--max-workgroups", type=int, default=0, help="cap on the workgroups of one launch (0 = none): its share of the device when several launches run side by side")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shape", choices=("tum", "eth3d"), default="tum", help="tum: 640x480, ~3 k points per cloud (the metric's configuration); eth3d: 736x456, ~9.3 k points (BASELINE config 5)")
    ap.add_argument("--gen-workers", type=int, default=0, help="host processes rendering the synthetic pairs (0 = one per host thread, 1 = no fork)")
    ap.add_argument("--no-adoption", action="store_true", help="do not let finished workgroups help with the pairs of their launch that still run (cvo_batch_set_adoption)")
    ap.add_argument("--adoption", action="store_true", help="force adoption on (whatever ADOPTION_DEFAULT says)")
    ap.add_argument("--reuse", choices=("any", "oldest"), default="any", help="which batch object in flight the next step reuses: whichever has completed first (cvo_batch_done), or strictly the oldest")
    ap.add_argument("--total-pairs", type=int, default=0, help="pairs per step over ALL ranks, dealt in contiguous blocks (cvo_shard_range: blocks may differ by one, the gather pads); 0 = --pairs per rank")
    ap.add_argument("--no-latency-probe", action="store_true", help="skip the single-pair / loop-closure / point-cloud latency measurements (counter passes)")
    ap.add_argument("--parity-only", action="store_true", help="with --no-cpu-baseline: still check every timed pair against the oracle (parity build, all host threads), without the timed CPU legs")
    ap.add_argument("--no-config5", action="store_true", help="do not append the `config5` object (a short run of BASELINE config 5, --shape eth3d, in a child process) to the default line")
    ap.add_argument("--dry-launch", action="store_true", help="every rank prints its block of the step's pairs (cvo_shard_range) as one JSON line and exits: no GPU needed")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:      # the bare invocation: this process becomes the launcher of the N ranks (before anything imports torch or touches the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], collect_stdout=args.dry_launch))
    if args.shape == "eth3d":      # BASELINE config 5 defaults: 4 workgroups per pair, 4 launches side by side, 64 workgroups (16 pair slots) each
        dflt = ap.parse_args([])
        if args.workgroups == dflt.workgroups: args.workgroups = 4
        if args.streams == dflt.streams: args.streams = 4
        if args.max_workgroups == dflt.max_workgroups: args.max_workgroups = 64
        if args.steps == dflt.steps: args.steps = 24
        if args.warmup == dflt.warmup: args.warmup = 4

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    global _SHAPE
    _SHAPE = args.shape
    adoption = args.adoption or (ADOPTION_DEFAULT and not args.no_adoption)
    # this rank's block of the step's pairs (contiguous blocks; with --total-pairs they may differ by one between ranks)
    total_pairs = args.total_pairs or world * args.pairs
    base, rem = divmod(total_pairs, world)
    first_pair, n_mine = rank * base + min(rank, rem), base + (1 if rank < rem else 0)
    n_block = (total_pairs + world - 1) // world
    first_shard = first_pair
    if args.dry_launch:
        from cvo_slam_amd import api
        blk = list(api.shard_range(total_pairs, rank, world))            # cvo_shard_range / cvo_shard_block: arithmetic of the C ABI, no device needed
        assert blk == list(range(first_pair, first_pair + n_mine)) and api.shard_block(total_pairs, world) == n_block
        try:
            rccl = api.comm_library_path()                               # dlopen + dladdr only: no device needed
        except Exception as e:
            rccl = f"not loadable ({e})"
        want_abi = os.environ.get("CVO_BENCH_BACKEND", "nccl") == "nccl" and os.environ.get("CVO_BENCH_GATHER", "abi") == "abi"
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world": world, "first_pair": first_pair, "pairs": n_mine,
                          "block_records": n_block, "total_pairs": total_pairs, "launched_by": os.environ.get("CVO_BENCH_LAUNCHED_BY", "external launcher"),
                          # what the N > 1 line will say about the collective (the timed run fills the same keys from the live communicator)
                          "ranks_in_communicator": world if world > 1 else 1, "gather": ("rccl" if want_abi else "torch") if world > 1 else "none",
                          "gather_streams": (("one (the communicator's own: CVO_BENCH_GATHER_STREAM)" if os.environ.get("CVO_BENCH_GATHER_STREAM") else
                                              "the align launches' own (one per step in flight)") if (world > 1 and want_abi) else None),
                          "rccl_library": rccl if world > 1 else None}), flush=True)
        return
    first_pair += int(os.environ.get("CVO_BENCH_PAIR_OFFSET", "0"))      # experiments: another set of synthetic pairs (the metric's set starts at 0)
    # host-side input generation first (forks; no GPU state yet)
    pairs = generate_pairs(first_pair, n_mine, args.gen_workers) if n_mine else []
    # the other sets of the `distinct_pairs` secondary loop (every batch object in flight its own pairs: streams x pairs different pairs), rendered here too
    want_distinct = world == 1 and not args.no_latency_probe and n_mine > 0 and not os.environ.get("CVO_BENCH_NO_DISTINCT_LOOP")
    other_sets = generate_pairs(first_pair + n_mine, (max(1, args.streams) - 1) * n_mine, args.gen_workers) if want_distinct and args.streams > 1 else []
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
        b = ca.CvoBatch(max(1, n_mine), device=local_rank)
        b.set_workgroups(args.workgroups)
        b.set_adoption(adoption)                                # takes effect in launches of one workgroup and one slot per pair: the tail of the job
        if args.max_workgroups:
            b.set_max_workgroups(args.max_workgroups)
        batches.append(b)
    prepared = ca.CvoBatch.prepare_pairs([(fx, ff, mx, mf) for (_, fx, ff, mx, mf) in pairs]) if pairs else None
    distinct = bool(os.environ.get("CVO_BENCH_DISTINCT"))      # experiment: every batch object its own set of pairs (object i: the set that starts at pair first_pair + i * n)
    for bi, b in enumerate(batches):
        if prepared and distinct and bi > 0:
            other = generate_pairs(first_pair + bi * n_mine, n_mine, 1)   # in this process: the GPU is initialised, no fork
            b.set_pairs(ca.CvoBatch.prepare_pairs([(fx, ff, mx, mf) for (_, fx, ff, mx, mf) in other]))
        elif prepared:
            b.set_pairs(prepared)                               # cvo_batch_set_pairs: one hand-over for the whole batch
    batch = batches[0]

    from cvo_slam_amd import shard, api
    n = n_mine
    assert list(api.shard_range(total_pairs, rank, world)) == list(range(first_shard, first_shard + n)) and api.shard_block(total_pairs, world) == n_block   # cvo_shard_range: contiguous blocks
    recvs = [torch.zeros((world * n_block, shard.RESULT_FLOATS), dtype=torch.float32, device="cuda") for _ in range(depth)] if world > 1 else []
    gathered = None
    inflight = []                                  # batch objects with a launch not yet waited for
    # The gather: ONE RCCL all-gather of the 64-byte records per step, enqueued by the C ABI behind the align launch on its
    # stream (cvo_batch_gather_results: pack kernel + ncclAllGather, no host sync in between).  The communicator's unique id
    # travels through torch.distributed once.  CVO_BENCH_GATHER=torch (or a non-RCCL backend) gathers with torch.distributed
    # after the wait instead.
    comm = None
    comm_ranks = None                              # ranks in the C ABI's RCCL communicator (None: no communicator was made)
    gather_mode = "none"
    if world > 1:
        gather_mode = "torch"
        if backend == "nccl" and os.environ.get("CVO_BENCH_GATHER", "abi") == "abi":
            try:                                   # does RCCL load on this rank at all?  (decided by all ranks together before anything collective)
                my_id = api.comm_unique_id(); can = 1
            except Exception as e:
                print(f"[bench] rank {rank}: RCCL not loadable through the C ABI ({e})", file=sys.stderr, flush=True)
                my_id = bytes(api.COMM_ID_BYTES); can = 0
            ok = torch.tensor([can], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                idt = torch.frombuffer(bytearray(my_id), dtype=torch.uint8).to("cuda")
                dist.broadcast(idt, 0)                                 # rank 0's id, everywhere
                try:
                    comm = ca.CvoComm(bytes(idt.cpu().numpy().tobytes()), world, rank, device=local_rank)   # ncclCommInitRank (collective)
                except Exception as e:
                    print(f"[bench] rank {rank}: cvo_comm_create failed ({e})", file=sys.stderr, flush=True)
                    comm = None
                ok2 = torch.tensor([1 if comm is not None else 0], device="cuda")
                dist.all_reduce(ok2, op=dist.ReduceOp.MIN)             # every rank has its communicator, or nobody uses one
                if int(ok2.item()) == 1:
                    gather_mode = "abi"
                    if os.environ.get("CVO_BENCH_GATHER_STREAM"):     # fallback mode: all gathers on the communicator's own stream (include/cvo_hip.h, "ORDER INVARIANT")
                        comm.set_gather_stream(True)
                    comm_ranks, comm_rank = comm.info()                # ncclCommCount / ncclCommUserRank: what the RCCL communicator itself reports
                    assert (comm_ranks, comm_rank) == (world, rank), (comm_ranks, comm_rank, world, rank)
                else:
                    if comm is not None:
                        comm.close()
                    comm = None
                    if rank == 0:
                        print("[bench] gathering with torch.distributed instead", file=sys.stderr, flush=True)
            elif rank == 0:
                print("[bench] gathering with torch.distributed instead", file=sys.stderr, flush=True)

    rccl_path = None
    if world > 1 and rank == 0:
        try:
            rccl_path = api.comm_library_path()    # dladdr of the ncclAllGather the C ABI bound
        except Exception as e:
            rccl_path = f"not loaded ({e})"
    kernel_ms = []                                 # HIP-event duration of every launch, on the stream it ran on

    launch_status = [0] * depth                    # what cvo_batch_align_async returned for the step a batch object holds
    send_ptr = [0] * depth

    def finish(bi):
        nonlocal gathered
        if n and launch_status[bi] == 0:
            batches[bi].wait()
            kernel_ms.append(batches[bi].last_launch()["kernel_ms"])
        else:
            kernel_ms.append(0.0)
            torch.cuda.synchronize()               # no launch to wait for: the rank's gather ran on the batch object's own stream
        if world > 1:
            if comm is not None:
                gathered = recvs[bi]                                           # already there: the all-gather ran behind the kernel
            else:                                                              # rehearsal backends: the rank's padded block through torch.distributed
                torch.cuda.synchronize()
                if send_ptr[bi]:
                    mine_blk = shard.device_view(send_ptr[bi], n_block)
                else:
                    mine_blk = torch.zeros((n_block, shard.RESULT_FLOATS), dtype=torch.float32, device="cuda"); mine_blk[:, 15] = float(api.CVO_ERR_RANK_FAILED)
                gathered = shard.gather_blocks(mine_blk, world)

    def pick():
        """The batch object the next step goes to: a free one, else whichever launch in flight has completed (cvo_batch_done, oldest first:
        alignments take data-dependent numbers of iterations, so launches do not finish in the order they were queued), else the oldest."""
        for k in range(depth):
            if k not in inflight:
                return k
        if args.reuse == "any":
            t_poll = time.perf_counter()
            while time.perf_counter() - t_poll < 0.05:
                for k in inflight:
                    if launch_status[k] or n == 0 or batches[k].done():
                        return k
        return inflight[0]

    host_prof = [0.0, 0.0, 0.0, 0.0, 0] if os.environ.get("CVO_BENCH_HOSTPROF") else None   # development aid: host seconds in pick / finish / reset_states / align_async, steps

    def step(i):
        tp0 = time.perf_counter()
        bi = pick()
        tp1 = time.perf_counter()
        if bi in inflight:                         # the object is reused: its previous step must be complete first
            inflight.remove(bi); finish(bi)
        tp2 = time.perf_counter()
        b = batches[bi]
        launch_status[bi] = 0
        if n:
            try:
                b.reset_states()                   # every step starts from R=I, T=0, ell=0.15
                tp3 = time.perf_counter()
                b.align_async(n)
                if host_prof is not None:
                    tp4 = time.perf_counter()
                    host_prof[0] += tp1 - tp0; host_prof[1] += tp2 - tp1; host_prof[2] += tp3 - tp2; host_prof[3] += tp4 - tp3; host_prof[4] += 1
            except ca.CvoError as e:               # the rank still enters the collective below: its records carry the status
                launch_status[bi] = e.code
        # the 64-byte records are written by the align kernel itself; every rank sends n_block of them (its own, then padding)
        if comm is not None:
            b.gather_results_padded(comm, n, n_block, recvs[bi].data_ptr(), launch_status[bi])   # ONE ncclAllGather on the launch's stream
        elif world > 1:
            try:
                send_ptr[bi] = b.padded_records(n, n_block, launch_status[bi])
            except ca.CvoError as e:               # this rank cannot make its block: it still enters the collective, with failure records
                send_ptr[bi] = 0; launch_status[bi] = launch_status[bi] or e.code
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
    elapsed_rank = elapsed
    if host_prof is not None and host_prof[4]:
        print("[bench] host us per step since start (pick, finish, reset_states, align_async): " + ", ".join(f"{1e6 * x / host_prof[4]:.1f}" for x in host_prof[:4]) + f" over {host_prof[4]} steps", file=sys.stderr, flush=True)
    assert len(kernel_ms) == args.steps
    if n == 0:                                     # a rank without pairs (fewer pairs than ranks): it only takes part in the gathers
        kernel_ms[:] = [0.0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    results = batch.wait(n) if n else []
    info = batch.last_launch() if n else {}

    # Secondary figure ("alignments incl. the post-align score block"): every step also queues the tracker's score
    # block (4 inner products + 1 Hessian per pair, cvo.cpp:475-503) behind its align launch and collects it with the results.
    with_scores = None
    if world == 1 and not args.no_latency_probe:
        k2 = max(depth, args.steps)                        # as many steps as the timed region: the same share of start-up and drain
        scored = []

        tail = not os.environ.get("CVO_BENCH_SCORE_LAUNCH")      # default: the align launch answers the block in its tail (cvo_batch_set_tail_scores); else a score launch queued behind it
        for b in batches:
            b.set_tail_scores(tail)

        def pick_of(busy_list):
            """the main loop's policy (pick): a free object, else whichever launch in flight has completed, else the oldest"""
            for k in range(depth):
                if k not in busy_list:
                    return k
            if args.reuse == "any":
                t_poll = time.perf_counter()
                while time.perf_counter() - t_poll < 0.05:
                    for k in busy_list:
                        if batches[k].done():
                            return k
            return busy_list[0]

        def step_scored(i):
            bi = pick_of(scored)
            if bi in scored:
                scored.remove(bi); batches[bi].wait(); batches[bi].innerproduct_results_raw(n)   # the C call: the scores are in the caller's array (no per-pair Python objects in the loop)
            b = batches[bi]
            b.reset_states(); b.align_async(n)
            if not tail:
                b.enqueue_innerproduct(n)
            scored.append(bi)

        def drain_scored():
            last = None
            while scored:
                bi = scored.pop(0); batches[bi].wait(); last = batches[bi].innerproduct_results_raw(n)
            return last

        for i in range(depth):
            step_scored(i)
        drain_scored(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(k2):
            step_scored(i)
        last_scores = drain_scored(); torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        answered = batches[0].last_tail_answers(n) if tail else []
        if tail and os.environ.get("CVO_BENCH_PHASES"):
            ts = batches[0].last_tail_seconds(); t_h = time.perf_counter(); batches[0].innerproduct_results_raw(n); t_h = time.perf_counter() - t_h
            print(f"[bench] score block in the tail, us per pair (workgroup 0): post + Hessian walk {1e6 * ts[0] / n:.1f}, cull for inn_pre {1e6 * ts[1] / n:.1f}, its walk {1e6 * ts[2] / n:.1f}, "
                  f"all {1e6 * ts[3] / n:.1f}; collecting a step's scores on the host {1e6 * t_h:.0f} us", file=sys.stderr, flush=True)
        for b in batches:
            b.set_tail_scores(False)
        with_scores = {"value": n * k2 / el2, "unit": "alignments/s", "steps": k2, "ms_per_step": 1e3 * el2 / k2,
                       "score_block": "compute_innerproduct per pair (4 inner products + 1 Hessian at the ell align() left behind): " +
                                      ("answered by the pair's workgroup in the tail of the align launch (inn_post + Hessian from the resident lists, inn_pre from one "
                                       "cull, the self products from the clouds' tables); the score kernel only for what a workgroup could not answer" if tail else
                                       "one score launch per step queued behind the align launch"),
                       "pairs_fully_answered_in_the_tail": int(sum(1 for m in answered if m == 31)),
                       "mean_cos_angle": float(np.mean([o.cos_angle for o in last_scores]))}
        batch.reset_states(); batch.align_async(n); batch.wait()       # leave batch 0 as the timed region left it

    # PCIe-inclusive rate, for the record (never `value`): every step first hands its 64 pairs over as host buffers in the
    # reference layout (cvo_batch_set_pair: packing + host-to-device copies), then aligns them.
    with_upload = None
    if world == 1 and not args.no_latency_probe:
        k3 = max(depth, args.steps)
        busy = []

        def step_upload(i):
            bi = pick_of(busy) if with_scores is not None else i % depth
            if bi in busy:
                busy.remove(bi); batches[bi].wait()
            b = batches[bi]
            b.set_pairs(prepared)                  # cvo_batch_set_pairs: the batch's host arrays, as they are, in one hand-over
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
        # the same loop with the caller's arrays in registered memory (cvo_host_register: pinned + mapped once): nothing is staged, the launch reads them in place
        reg_rate = None
        try:
            arena = np.zeros(sum(a.size for (_, fx, ff, mx, mf) in pairs for a in (fx, ff, mx, mf)) + 64, np.float32)
            o = 0; reg_clouds = []
            for (_, fx, ff, mx, mf) in pairs:
                vs = []
                for a in (fx, ff, mx, mf):
                    v = arena[o:o + a.size].reshape(a.shape); v[...] = a; o += a.size; vs.append(v)
                reg_clouds.append(tuple(vs))
            api.host_register(arena)
            prepared_reg = ca.CvoBatch.prepare_pairs(reg_clouds)
            saved, prepared = prepared, prepared_reg
            for i in range(depth):
                step_upload(i)
            while busy:
                batches[busy.pop(0)].wait()
            torch.cuda.synchronize()
            t3r = time.perf_counter()
            for i in range(k3):
                step_upload(i)
            while busy:
                batches[busy.pop(0)].wait()
            torch.cuda.synchronize()
            reg_rate = n * k3 / (time.perf_counter() - t3r)
            prepared = saved
            for b in batches:
                b.set_pairs(prepared)                      # (nothing of the arena is pending when it is unregistered)
            batch.reset_states(); batch.align_async(n); batch.wait()
            api.host_unregister(arena)
        except Exception as e:
            print(f"[bench] registered-memory hand-over loop failed: {e}", file=sys.stderr, flush=True)
        with_upload = {"value": n * k3 / el3, "unit": "alignments/s", "steps": k3, "ms_per_step": 1e3 * el3 / k3, "host_MB_per_step": mb,
                       "from_registered_memory": {"value": reg_rate, "unit": "alignments/s", "note": "the same hand-over from a range the caller registered once (cvo_host_register): no staging copy, the align launch reads the caller's arrays over PCIe"} if reg_rate else None,
                       "note": "clouds cross the boundary as host buffers every step (cvo_batch_set_pairs: the arrays are copied as they are into a pinned block, "
                               "the align launch that follows builds the device layout itself, reading the block over PCIe)"}
        batch.reset_states(); batch.align_async(n); batch.wait()
    # The timed region's eight batch objects hold the SAME 64 pairs (every step aligns the batch BASELINE config 3 names).  For the record: the same loop with
    # every object in flight on its own set -- streams x pairs different pairs on the device at once (the launches then do not fall into step with each other).
    with_distinct = None
    if want_distinct and other_sets and with_scores is not None:
        for bi in range(1, depth):
            sl = other_sets[(bi - 1) * n:(bi) * n]
            batches[bi].set_pairs(ca.CvoBatch.prepare_pairs([(fx, ff, mx, mf) for (_, fx, ff, mx, mf) in sl]))
        k4 = max(depth, args.steps)
        busy4 = []

        def step_distinct(i):
            bi = pick_of(busy4)
            if bi in busy4:
                busy4.remove(bi); batches[bi].wait()
            batches[bi].reset_states(); batches[bi].align_async(n)
            busy4.append(bi)

        def drain_distinct():
            while busy4:
                batches[busy4.pop(0)].wait()
            torch.cuda.synchronize()

        for i in range(depth):
            step_distinct(i)
        drain_distinct()
        t4 = time.perf_counter()
        for i in range(k4):
            step_distinct(i)
        drain_distinct()
        el4 = time.perf_counter() - t4
        bad4 = [r["status"] for b in batches for r in b.wait(n) if r["status"] != 0]
        with_distinct = {"value": n * k4 / el4, "unit": "alignments/s", "steps": k4, "ms_per_step": 1e3 * el4 / k4, "distinct_pairs_in_flight": depth * n, "errors": len(bad4),
                         "note": "every batch object in flight aligns its own set of pairs (object i: pairs i*n .. i*n+n-1 of the seeded generator); the timed region's objects all hold set 0"}
        for bi in range(1, depth):
            batches[bi].set_pairs(prepared)
    if rank == 0 and os.environ.get("CVO_BENCH_PHASES"):
        ph = batch.last_phase_seconds(); its_all = sum(r["iterations_run"] for r in results)
        print("[bench] phase us/iteration under load (workgroup 0 of every pair of the last launch): " +
              str({k: round(1e6 * v / its_all, 1) for k, v in ph.items()}) + f"; culls per pair {np.mean([r['rebuilds'] for r in results]):.2f}, iterations {its_all / len(results):.1f}", file=sys.stderr, flush=True)
        masks, pmasks = batch.last_cull_masks(len(results))
        hist = [sum((m >> k) & 1 for m in masks) for k in range(64)]
        print("[bench] culls by iteration (pairs of the last launch that culled at k = 0, 1, ...; 63 = later): " + " ".join(str(h) for h in hist) +
              f"; around extrapolated positions: {sum(bin(m).count('1') for m in pmasks)}", file=sys.stderr, flush=True)
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
        value = total_pairs * args.steps / elapsed
        step_ms_rank = 1e3 * elapsed / args.steps          # one launch retires every step_ms_rank on this GPU
        overlap = k_ms / step_ms_rank                      # launches running side by side, on average
        # Counter figures come from separate rocprofv3 --pmc passes over this kernel (scripts/pmc_run.sh -> profiles/pmc_traffic.json,
        # taken at the commit named inside it): HBM bytes and VALU wave-instructions ONE launch of this workload executes.  Both are
        # properties of the work, not of the timing; the rates below divide them by times measured live in this run.
        traffic = valu_instr = pmc_src = None; pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f).get(args.shape, {})
            if pmc and int(pmc.get("pairs", n)) == n:
                traffic = float(pmc["hbm_bytes_per_launch"]); valu_instr = float(pmc.get("valu_wave_instructions_per_launch", 0)) or None
                pmc_src = pmc.get("source")
        except Exception:
            pass
        prof_ms = prof_file = None                       # rocprofv3 --kernel-trace --stats of this command, committed: its average must agree with kernel_ms
        try:
            import csv
            prof_file = "profiles/r05_kernel_stats.csv" if args.shape == "tum" else "profiles/r05_eth3d_kernel_stats.csv"
            with open(os.path.join(ROOT, prof_file), newline="") as f:
                for row in csv.DictReader(f):
                    if "cvo_align_kernel" in row.get("Name", ""):
                        prof_ms = float(row["AverageNs"]) * 1e-6
        except Exception:
            prof_file = None
        # The kernel is bound by VALU issue, not by HBM: the roofline object prices the vector pipe.
        # MI355X_MICROARCH.md: a SIMD issues a wave64 f32 instruction in 2 cycles (one wave alone: one per 4) and f64 at half that rate
        # (157.3 / 78.6 TFLOP/s vector peaks); scripts/micro/valu_rate.hip measures the same on this chip (profiles/r02_valu_issue_microbench.txt:
        # 2.35 cycles for the f32/int32 class, 4.4 for f64 arithmetic, f64 conversions and 64-bit integer ops, 8.4 for transcendentals, two
        # or more waves per SIMD).  The peak is therefore priced for THIS kernel's instruction mix, taken from the SQ_INSTS_VALU_* class
        # counters of the same PMC passes: peak = 256 CUs x 4 SIMDs x 2.4 GHz / (mean issue cycles per instruction at the guide's rates).
        # (Rounds 1-2 divided by 4 cycles per instruction -- what ONE wave per SIMD can issue, not the SIMD: that overstated the fraction
        # by the factor kept below as `frac_of_one_wave_issue_rate`.)
        step_s = step_ms_rank * 1e-3
        simd_hz = 256 * 4 * 2.4e9
        share4 = float(pmc.get("valu_half_rate_share", 0.0)) if pmc_src else 0.0        # f64 add/mul/fma + conversions + 64-bit integer: 4 cycles
        share8 = float(pmc.get("valu_transcendental_share", 0.0)) if pmc_src else 0.0   # 8 cycles
        cyc_nominal = 2.0 * (1.0 - share4 - share8) + 4.0 * share4 + 8.0 * share8
        cyc_measured = 2.35 * (1.0 - share4 - share8) + 4.4 * share4 + 8.4 * share8
        valu_peak = simd_hz / cyc_nominal
        valu_rate = (valu_instr / step_s) if valu_instr else None
        pair_tests = float(info.get("candidates_total", 0))          # list candidates the last launch evaluated (every launch of the region does the same work)
        out = {
            "metric": "CVO frame-pair alignments/sec (640x480, ~3k pts/cloud)",
            "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "timed_region_s": elapsed,
            "ranks_in_communicator": comm_ranks if world > 1 else 1,
            "gather": ("rccl" if gather_mode == "abi" else ("torch" if backend == "nccl" else backend)) if world > 1 else "none",
            "gather_streams": ("one (the communicator's own: CVO_BENCH_GATHER_STREAM)" if os.environ.get("CVO_BENCH_GATHER_STREAM") else "the align launches' own (one per step in flight)") if (world > 1 and gather_mode == "abi") else None,
            "rccl_library": rccl_path if world > 1 else None,
            "config": {"workload": f"{n} independent synthetic {'640x480 TUM' if args.shape == 'tum' else '736x456 ETH3D'}-shape RGB-D pairs per GPU per step "
                                   f"(BASELINE config {'3' if args.shape == 'tum' else '5'}; {total_pairs} pairs per step at {world} GPU(s); config 4 = 512 pairs at 8 GPUs), "
                                   f"full align() from R=I,T=0,ell=0.15 to convergence",
                       "pairs_per_gpu": n, "points_fixed_mean": float(np.mean(nfs)), "points_moving_mean": float(np.mean(nms)),
                       "iterations_mean": float(np.mean(its)), "iterations_max": int(np.max(its)),
                       "workgroups_per_pair": args.workgroups or "auto", "steps_in_flight": depth, "adoption": adoption, "reuse": args.reuse,
                       "single_step_ms_unpipelined": single_step_ms, "single_kernel_ms_unpipelined": single_kernel_ms, "collective": ((("RCCL ncclAllGather enqueued by the C ABI behind each align launch" if gather_mode == "abi" else ("RCCL" if backend == "nccl" else backend) + " all_gather via torch.distributed after the wait") + ", 64-byte result records") if world > 1 else "none (1 GPU)")},
            "roofline": {"bound": "valu_issue", "achieved": valu_rate, "peak": valu_peak, "unit": "wave-instructions/s",
                         "frac": (valu_rate / valu_peak) if valu_rate else None,
                         "traffic": traffic, "kernel": "cvo_align_kernel", "kernel_ms": k_ms,
                         "valu_wave_instructions_per_launch": valu_instr, "counters": pmc_src,
                         "issue_cycles_per_instruction": {"guide_rates": cyc_nominal, "measured_rates": cyc_measured, "half_rate_share": share4, "transcendental_share": share8},
                         "frac_of_measured_issue_rate": (valu_rate / (simd_hz / cyc_measured)) if valu_rate else None,
                         "frac_of_one_wave_issue_rate": (valu_rate / (simd_hz / 4.0)) if valu_rate else None,
                         "launches_side_by_side": overlap, "profile": prof_file, "profile_kernel_ms": prof_ms,
                         "note": "achieved = VALU wave-instructions one launch executes (SQ_INSTS_VALU, rocprofv3 --pmc at the commit named in `counters`) / "
                                 "time per step measured live (one launch retires per step); peak = 256 CUs x 4 SIMDs x 2.4 GHz / issue cycles per instruction "
                                 "of this kernel's mix at the guide's rates (f32/int32 class 2 cycles, f64 + conversions + int64 4, transcendental 8; the shares "
                                 "are SQ_INSTS_VALU_* counters of the same passes).  frac_of_measured_issue_rate uses the rates scripts/micro/valu_rate.hip "
                                 "measures on this chip instead (profiles/r02_valu_issue_microbench.txt); frac_of_one_wave_issue_rate is the round-1/2 "
                                 "definition (4 cycles per instruction: what one wave per SIMD can issue), kept for comparison only.  kernel_ms = mean HIP-event "
                                 "duration of a launch on its own stream with launches_side_by_side of them sharing the CUs (so it contains queueing; the "
                                 "rocprofv3 --kernel-trace average of the same command is `profile_kernel_ms`)"},
            "hbm": {"bound": "hbm", "achieved": bytes_launch / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_launch / step_s / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": bytes_launch, "traffic": traffic, "traffic_rate_GBs": (traffic / step_s / 1e9) if traffic else None,
                    "per_launch_achieved_GBs": achieved_gbs,
                    "note": "secondary: chip-level rate = algorithmic bytes of one step (SURVEY 8d: 2(32N+32M)+24M per iteration x executed iterations) / time "
                            "per step; traffic = FETCH_SIZE (doubled: gfx950 wide-read under-count) + WRITE_SIZE per launch from the PMC passes; "
                            "per_launch_achieved_GBs divides by one launch's own duration instead"},
            "with_score_block": with_scores,
            "with_host_upload": with_upload,
            "distinct_pairs": with_distinct,
            "work": {"nonzeros_per_launch": float(info.get("nonzeros_total", 0)) or None,
                     "valu_lane_instructions_per_nonzero": (64.0 * valu_instr / float(info["nonzeros_total"])) if (valu_instr and info.get("nonzeros_total")) else None,
                     "executed_pair_tests_per_s": pair_tests / step_s if pair_tests else None,
                     "dense_pair_tests_per_s_equivalent": flops_launch / 8.0 / step_s,
                     "note": "nonzeros = members of the sparse kernel matrix A summed over all iterations of one step (what the reference's arithmetic is defined on: cvo.cpp:166-175, "
                             "213-223, 282-306); valu_lane_instructions_per_nonzero = 64 x VALU wave-instructions of a launch (PMC) / those nonzeros -- the reference's float "
                             "sequence needs about 190-220 of them for a member (DESIGN.md 4.1), the rest is candidates that are listed but no members, list upkeep, reductions; "
                             "executed_pair_tests = list candidates the kernel evaluated with the reference's exact expression (all iterations of one step); the "
                             "dense equivalent counts the N*M tests per iteration the reference's radius search stands for (most are skipped by lists + box cull)"},
        }
        if world == 1 and not args.no_latency_probe:
            # the throughput loops are over: their eight batch objects (and streams) go before the latency probes run -- a process with more streams than hardware
            # queues has streams sharing a queue, and a generator kernel queued behind a persistent align kernel of another stream waits for all of it
            for b in batches:
                b.close()
            batches.clear()
            out["latency"] = latency_probe(ca, pairs, local_rank)
        if not args.no_cpu_baseline and world == 1:
            cores = host_threads()
            print(f"[bench] GPU done ({value:.1f} alignments/s); timing the CPU baseline on {cores} threads ...", file=sys.stderr, flush=True)
            cb = cpu_baseline(pairs, cores)
            cpu_rate, cpu_tfs, cpu_its = cb["rate"], cb["tfs"], cb["iters"]
            errs = [rot_trans_err(results[i]["transform"], cpu_tfs[i]) for i in range(len(cpu_tfs))]
            out["cpu_baseline"] = {"value": cpu_rate, "unit": "alignments/s", "cores": cores, "kind": "port",
                                   "sample": f"oracle (CPU restatement of the reference: KD-tree rebuilt per iteration, two sparse sweeps; not the icpc binary), "
                                             f"-O3 -march=native -ffp-contract=fast build, the {len(pairs)} pairs of the timed batch, one alignment per host thread, "
                                             f"{cores} at once; value = median of {len(cb['rates'])} passes; {cb['seconds']:.1f} s for all CPU legs",
                                   "passes": cb["rates"],
                                   "row_parallel_value": cb["row_parallel_rate"], "row_parallel_serial_fraction": cb["row_parallel_serial_fraction"],
                                   "row_parallel_note": f"the reference's own structure: one alignment at a time, row loops on {cores} threads; the KD-tree build and CSR assembly "
                                                        "of every iteration are serial (cvo.cpp:135-136, 182-183) and take the stated share of the wall time",
                                   "single_thread_value": cb["single_thread_rate"], "single_thread_value_parity_build": cb["single_thread_rate_parity_build"],
                                   "parity_build_value": cb["parity_build_rate"],
                                   "fast_build_vs_parity_build_max_err": {"rot_rad": cb["fast_vs_parity_build"][0], "trans_m": cb["fast_vs_parity_build"][1]},
                                   "iterations_mean": float(np.mean(cpu_its))}
            env = None
            try:
                with open(os.path.join(ROOT, "tests", "golden", "noise_envelope.json")) as f:
                    ej = json.load(f)["tum64" if args.shape == "tum" else "eth3d64"]
                env = {"max_rot_rad": ej["max_rot_rad"], "max_trans_m": ej["max_trans_m"], "pairs_beyond_1e-4": ej["pairs_beyond_1e-4"],
                       "per_variant": {k: [v["max_rot_rad"], v["max_trans_m"]] for k, v in ej["per_variant"].items()},
                       "source": "tests/golden/noise_envelope.json (scripts/make_noise_envelope.py): distance of the oracle's reference-noise variants from the base oracle on these 64 pairs"}
            except Exception:
                pass
            out["parity"] = {"pairs_checked": len(errs), "against": "oracle, un-fused parity build (every pair of the timed batch)",
                             "max_rot_err_rad": max(e[0] for e in errs), "max_trans_err_m": max(e[1] for e in errs),
                             "iterations_equal": bool(all(a == b for a, b in zip(its, cpu_its))), "tolerance": "1e-4 rad / 1e-4 m",
                             "reference_noise_envelope": env}
            out["speedup_vs_cpu_baseline"] = value / cpu_rate
        elif args.parity_only and world == 1:
            out["parity"] = parity_check(pairs, results, its, host_threads())
        if args.shape == "tum" and world == 1 and not args.no_config5 and not args.no_latency_probe and not under_profiler():
            out["config5"] = config5_child()
        print(json.dumps(out), flush=True)

    if world > 1:
        if gathered is not None and rank == 0:
            table, first_err = api.compact_records(gathered.cpu().numpy(), total_pairs, world)      # cvo_compact_records: global pair order, padding dropped
            assert table.shape == (total_pairs, shard.RESULT_FLOATS) and first_err == 0, f"gathered records incomplete or a rank reported an error (status {first_err})"
            print(f"[bench] gathered {total_pairs} records from {world} ranks (blocks of {n_block}, {world * n_block - total_pairs} padding records)", file=sys.stderr, flush=True)
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
