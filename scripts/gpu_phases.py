"""Per-phase time of the align kernel on a 64-pair batch, for several workgroup counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
npairs = int(os.environ.get("NPAIRS", "64"))
pairs = bench.generate_pairs(0, npairs)
import cvo_slam_amd as ca
B = ca.CvoBatch(npairs)
for i, (_, fx, ff, mx, mf) in enumerate(pairs):
    B.set_pair(i, fx, ff, mx, mf)
for wgs in [int(x) for x in os.environ.get("WGS", "1,2,4").split(",")]:
    B.set_workgroups(wgs)
    for rep in range(2):
        B.reset_states(); res = B.align(npairs)
    info = B.last_launch(); ph = B.last_phase_seconds()
    its = info["iterations_total"]
    cand_sub = {"prologue": ph.pop("cand_prologue"), "rows": ph.pop("cand_rows"), "wg_reduce": ph.pop("cand_reduce"), "exchange": ph.pop("cand_exchange")}
    rb_sweep = ph.pop("lists_cull"); rb_sort = ph.pop("lists_sort")
    tot = sum(ph.values())
    print(f"wgs={wgs} kernel {info['kernel_ms']:.2f} ms, iterations {its} (max {max(r['iterations_run'] for r in res)}), cand/iter {info['candidates_total']/its:.0f}")
    rb = sum(r["rebuilds"] for r in res); df = sum(r["dense_fallbacks"] for r in res)
    print(f"   rebuilds {rb} ({rb/len(res):.1f}/pair), dense fallbacks {df}; cull per rebuild {1e6*rb_sweep/max(rb,1):.0f} us, sort {1e6*rb_sort/max(rb,1):.1f} us")
    print("   per-iteration us (wg0):", {k: round(1e6 * v / its, 1) for k, v in ph.items()}, "sum", round(1e6 * tot / its, 1))
    print("   inside candidates (thread 0):", {k: round(1e6 * v / its, 1) for k, v in cand_sub.items()})
