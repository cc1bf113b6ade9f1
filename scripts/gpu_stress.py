"""One-off randomized parity sweep (development aid): random small pairs, workgroup counts, tiles, layouts and skins against the
oracle; prints every mismatch.  Exit code 1 on any."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import cvo_slam_amd as ca
from cvo_slam_amd import synth
from helpers import rot_trans_err
po.build()
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
bad = 0
N = int(os.environ.get("CASES", "60"))
for case in range(N):
    n = int(rng.choice([64, 130, 257, 400, 700, 1100, 1500, 2300, 3072]))
    p = synth.make_small_pair(1000 + case, n=n, max_deg=float(rng.uniform(0.5, 3.0)), max_trans=float(rng.uniform(0.005, 0.05)))
    cut = int(rng.integers(0, 40))
    fixed = (p.fixed.xyz, p.fixed.feat); moving = (p.moving.xyz[: n - cut], p.moving.feat[:, : n - cut])
    o = po.OracleCvo(); o.set_pcd(*fixed); o.set_pcd(*moving); rc, otr = o.align(trace_cap=3000); ost = o.get_state()
    env = {}
    if rng.random() < 0.4: env["CVO_HIP_TILE"] = str(int(rng.choice([128, 256, 384, 1024])))
    if rng.random() < 0.4: env["CVO_HIP_Y_MODE"] = str(int(rng.choice([0, 1, 2])))
    if rng.random() < 0.3: env["CVO_HIP_SKIN"] = str(float(rng.choice([0.0, 0.1, 0.5, 1.0])))
    if rng.random() < 0.2: env["CVO_HIP_ROW_CAP"] = str(int(rng.choice([8, 16, 64])))
    if rng.random() < 0.2: env["CVO_HIP_FLAT_CAP"] = str(int(rng.choice([1, 4, 32])))
    if rng.random() < 0.5: env["CVO_HIP_RESORT"] = str(int(rng.choice([0, 2])))
    wgs = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 16, 32]))
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g = ca.Cvo(); g.set_workgroups(wgs); g.set_pcd(*fixed); g.set_pcd(*moving); gtr = g.align(trace_cap=3000)
        re, te = rot_trans_err(g.transform, ost["transform"])
        ok = re <= 1e-6 and te <= 1e-6 and g.get_iteration_number() == ost["iter"] and g.get_A_nonzero() == ost["A_nonzero"] and [r["nnz"] for r in gtr] == [r["nnz"] for r in otr]
        # the score block at the ell the alignment left behind (Q1): pair counts exact, sums to f32 rounding
        rc2, so = o.compute_innerproduct(ost["transform"]); sg = g.compute_innerproduct(ost["transform"])
        for key in ("inn_pre", "inn_post", "inn_fixed_pcd", "inn_moving_pcd"):
            ok = ok and sg[key][1] == so[key][1] and abs(sg[key][0] - so[key][0]) <= 3e-6 * abs(so[key][0]) + 1e-12
        ok = ok and sg["inliers"] == so["inliers"]
        g.close()
    except Exception as e:   # noqa: BLE001
        ok = False; re = te = float("nan"); print("EXC", repr(e))
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={n} cut={cut} wgs={wgs} env={env} rot {re:.2e} trans {te:.2e}")
print(f"{N - bad} / {N} cases identical to the oracle")
sys.exit(1 if bad else 0)
