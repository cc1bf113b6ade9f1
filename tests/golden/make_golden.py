"""Generates the committed fixtures under tests/golden/.

The reference ships no tests or golden vectors for this path and cannot be built or
run here, so the fixtures are:
  closed_forms.json  known answers from numpy / scipy (independent of the oracle):
                     cubic real roots (numpy.roots), SE(3) exponential (scipy expm),
                     ||logm||_F (scipy logm), symmetric 6x6 eigenvalues (numpy eigvalsh)
  small_pair_*.npz   inputs + the oracle's per-iteration trace, final state and scores
                     (single thread, brute-force search) -- regression pins for the oracle
                     and known inputs/outputs for the HIP path
  tum_pair_0.npz     one full-size (3072-point) synthetic TUM-shape pair, same content
  pcd_frame_small.npz  a 208x160 crop of a synthetic RGB-D frame + the oracle's point cloud for it (pcd generator,
                     SURVEY 8f next-1)
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po                      # noqa: E402
from cvo_slam_amd import synth             # noqa: E402
from helpers import se3_exp                # noqa: E402


def closed_forms():
    from scipy.linalg import logm
    rng = np.random.default_rng(1234)
    out = {"cubic": [], "exp": [], "dist": [], "eig": []}
    # cubic: 4E t^3 + 3D t^2 + 2C t + B, float32 coefficients; expected = smallest positive real root (numpy.roots)
    cases = [(1.0, -6.0, 11.0, -6.0), (2.0, 1.0, 3.0, 4.0), (1.0, 0.0, 0.0, -8.0), (-3.0, 2.0, 5.0, -1.0), (0.5, -0.1, -2.0, 0.3)]
    for _ in range(40):
        cases.append(tuple(float(x) for x in rng.normal(size=4) * rng.choice([1e-3, 1.0, 1e3], size=4)))
    for c in cases:
        c32 = np.array(c, np.float32)
        r = np.roots(c32.astype(np.float64))
        real = sorted(x.real for x in r if abs(x.imag) < 1e-9 * max(1.0, abs(x.real)) and x.real > 0)
        out["cubic"].append({"coef": [float(x) for x in c32], "smallest_positive_real_root": (real[0] if real else None)})
    for _ in range(12):
        w = rng.normal(size=3) * rng.choice([1e-3, 0.05, 0.5]); v = rng.normal(size=3) * 0.3; dt = float(rng.uniform(0.01, 0.8))
        X = se3_exp(w.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64), np.float64(np.float32(dt)))
        out["exp"].append({"omega": [float(np.float32(x)) for x in w], "v": [float(np.float32(x)) for x in v], "dt": float(np.float32(dt)),
                           "R": X[:3, :3].ravel().tolist(), "t": X[:3, 3].tolist()})
    for scale in (1e-6, 1e-5, 1e-4, 1e-2, 0.3):
        for _ in range(3):
            w = rng.normal(size=3) * scale; v = rng.normal(size=3) * scale
            X = se3_exp(w, v, 1.0)
            R32 = X[:3, :3].astype(np.float32); t32 = X[:3, 3].astype(np.float32)
            T = np.eye(4); T[:3, :3] = R32; T[:3, 3] = t32
            L = logm(T)
            out["dist"].append({"dR": R32.ravel().tolist(), "dT": t32.tolist(), "frob_log": float(np.linalg.norm(np.real(L)))})
    for _ in range(6):
        A = rng.normal(size=(6, 6)); S = ((A + A.T) * rng.choice([1e3, 1e5, 1e7])).astype(np.float32)
        Hs = S * np.float32(-1.0 / 100000)
        out["eig"].append({"H": S.ravel().tolist(), "eig_scaled": np.linalg.eigvalsh(Hs.astype(np.float64)).tolist()})
    with open(os.path.join(HERE, "closed_forms.json"), "w") as f:
        json.dump(out, f)


def run_pair(pair, trace_cap=400):
    o = po.OracleCvo(search=po.SEARCH_BRUTE, threads=1)
    o.set_pcd(pair.fixed.xyz, pair.fixed.feat); o.set_pcd(pair.moving.xyz, pair.moving.feat)
    rc, tr = o.align(trace_cap=trace_cap)
    assert rc == 0
    st = o.get_state()
    rc, sc = o.compute_innerproduct(st["transform"])
    assert rc == 0
    return dict(
        fixed_xyz=pair.fixed.xyz, fixed_feat=pair.fixed.feat, moving_xyz=pair.moving.xyz, moving_feat=pair.moving.feat,
        trace_omega=np.array([r["omega"] for r in tr]), trace_v=np.array([r["v"] for r in tr]),
        trace_nnz=np.array([r["nnz"] for r in tr]), trace_BCDE=np.array([r["BCDE"] for r in tr]),
        trace_step=np.array([r["step"] for r in tr], np.float32), trace_ell=np.array([r["ell"] for r in tr], np.float32),
        trace_dist=np.array([r["dist"] for r in tr], np.float32),
        final_transform=st["transform"], final_R=st["R"], final_T=st["T"], final_ell=np.float32(st["ell"]),
        iter=np.int32(st["iter"]), A_nonzero=np.int32(st["A_nonzero"]),
        inn_pre=np.array(sc["inn_pre"], np.float64), inn_post=np.array(sc["inn_post"], np.float64),
        inn_fixed=np.array(sc["inn_fixed_pcd"], np.float64), inn_moving=np.array(sc["inn_moving_pcd"], np.float64),
        cos_angle=np.float32(sc["cos_angle"]), post_hessian=sc["post_hessian"], inliers=np.int32(sc["inliers"]))


def pcd_fixture():
    (fa, da), _, _ = synth.make_frames(7)
    bgr = np.ascontiguousarray(fa[100:260, 200:408]); dep = np.ascontiguousarray(da[100:260, 200:408])      # 160 x 208
    cam = synth.camera_tuple(synth.TUM1)
    r = po.pcd_generate(bgr, dep, cam, num_want=260, debug=True)
    np.savez_compressed(os.path.join(HERE, "pcd_frame_small.npz"), bgr=bgr, depth=dep, camera=np.array(cam, np.float32), num_want=260,
                        px=r["px"], xyz=r["xyz"], feat=r["feat"], info=r["info"])
    print("pcd_frame_small.npz:", r["n"], "points, info", r["info"])



if __name__ == "__main__":
    po.build()
    if "--pcd-only" in sys.argv:
        pcd_fixture(); sys.exit(0)
    closed_forms()
    for seed in (11, 12, 13):
        np.savez_compressed(os.path.join(HERE, f"small_pair_{seed}.npz"), **run_pair(synth.make_small_pair(seed, n=300)))
    np.savez_compressed(os.path.join(HERE, "tum_pair_0.npz"), **run_pair(synth.make_pair(0)))
    pcd_fixture()
    print("golden fixtures written to", HERE)
