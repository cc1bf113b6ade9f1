"""development aid: what slows the walks down when the chip is full?  One 64-pair launch (64 of the 256 CUs) alone, beside a stream of large
device-to-device copies (HBM bandwidth taken, CUs mostly free), and beside a stream of arithmetic on an L2-resident tensor (CUs taken, HBM idle):
per-iteration phase times of the align kernel (CVO_BENCH_PHASES' timers) in the three settings."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import cvo_slam_amd as ca
from cvo_slam_amd import synth
ca.load_library()
N = 64
pairs = [synth.make_pair(i) for i in range(N)]
b = ca.CvoBatch(N); b.set_workgroups(1); b.set_adoption(False)
b.set_pairs([(p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat) for p in pairs])
b.align_async(N); want = b.wait(N)
its = sum(r["iterations_run"] for r in want)
side = torch.cuda.Stream()
src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_(); dst = torch.empty_like(src)       # 1 GiB each
small = torch.empty(1 << 18, dtype=torch.float32, device="cuda").uniform_(0.1, 1.0)                         # 1 MiB: lives in L2
wide = torch.empty(1 << 24, dtype=torch.float32, device="cuda").uniform_(0.1, 1.0)                          # 64 MiB, many workgroups


def run(label, hog):
    out = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if hog is not None:
            with torch.cuda.stream(side):
                hog()
        b.reset_states(); b.align_async(N); got = b.wait(N)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ph = b.last_phase_seconds(); ll = b.last_launch()
        for w, g in zip(want, got):
            assert g["status"] == 0 and np.array_equal(g["transform"], w["transform"])
        out.append((ll["kernel_ms"], {k: round(1e6 * v / its, 1) for k, v in ph.items() if k in ("lists_cull", "cand_rows", "linesearch", "epilogue")}, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    for o in out: print(f"{label}: kernel {o[0]:.2f} ms, phases us/iteration {o[1]}, align wall {o[2]:.1f} ms, hog done after {o[3]:.1f} ms", flush=True)


def copies():
    for _ in range(40): dst.copy_(src, non_blocking=True)            # 40 x 2 GiB of traffic: ~16 ms at 5 TB/s

def reads():
    for _ in range(60): torch.sum(src)                               # read-only stream

def arith_small():
    x = small
    for _ in range(3000): x = torch.sin(x)                           # tiny kernels, L2-resident: little HBM, few CUs

def arith_wide():
    for _ in range(200): torch.special.erf(torch.sin(wide))          # 64 MiB per op: mostly arithmetic per byte? (sin+erf), every free CU busy

run("alone", None)
run("beside 1 GiB copies (read + write)", copies)
run("beside 1 GiB reductions (read only)", reads)
run("beside sin() on 1 MiB", arith_small)
run("beside erf(sin()) on 64 MiB", arith_wide)
run("alone again", None)
b.close()
