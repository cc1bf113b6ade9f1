"""The C++ oracle against a SECOND reading of the reference (tests/second_reading.py: dense numpy, written from cvo.cpp / LieGroup.cpp
alone).  Oracle and kernels share an author; this restatement shares neither their code nor their evaluation strategy (no neighbour
search, no lists: every pair formed and masked).  Stage-locked: every stage of every iteration is fed the ORACLE's inputs for that
stage (its pose -- from its own run cut short at that iteration --, its omega / v, its B..E) and its output is compared with the oracle's trace, so a disagreement names the stage and the
tolerances can be the rounding of one stage instead of the drift of sixty iterations:

  pose_k --[transform, se_kernel, compute_flow]--> nnz (exact), omega, v (3e-7 of the terms' magnitudes; plus an ulp of the transformed points with the plain association)
  oracle's omega, v --[compute_step_size sums]--> B, C, D, E (1e-9 with the oracle's association of 3-term sums, 1e-6 of the terms' magnitudes with the plain one)
  oracle's B..E --[poly_solver, root choice]--> step (1e-6)
  oracle's pose_k, omega, v, step --[Exp_SEK3, pose update]--> pose_{k+1} (3e-7 per entry); stop tests; ell schedule

and at the end the pose (1e-6), the iteration count, function_inner_product (count exact, value 1e-6) and the raw Hessian (1e-4).
CPU only; runs in the build container."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import rot_trans_err
import second_reading as sr

f32 = np.float32


def _close(a, b, rtol, what, k):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, f"iteration {k}: {what} differs by {err:.3e} of its size ({a} vs {b})"


@pytest.mark.parametrize("assoc", ["left", "unrolled"])
@pytest.mark.parametrize("name", ["small_pair_11.npz", "small_pair_12.npz", "small_pair_13.npz"])
def test_second_reading_agrees_with_the_oracle_stage_by_stage(oracle, name, assoc, monkeypatch):
    monkeypatch.setattr(sr, "ASSOC3", assoc)
    g = np.load(os.path.join(GOLDEN, name))
    x, fx, p, fp = g["fixed_xyz"], g["fixed_feat"], g["moving_xyz"], g["moving_feat"]
    # the oracle's own run on the same clouds (tests/test_oracle_golden.py pins it to the committed trace bit for bit)
    o = oracle.OracleCvo()
    o.set_pcd(x, fx); o.set_pcd(p, fp)
    rc, tr = o.align(trace_cap=400)
    assert rc == 0 and len(tr) >= 10
    st = o.get_state()

    def oracle_pose_before(k):
        """R, T with which the oracle enters iteration k: its own run cut short by MAX_ITER = k (cvo.cpp:768)."""
        pr = oracle.default_params(); pr.max_iter = k
        ok = oracle.OracleCvo(params=pr)
        ok.set_pcd(x, fx); ok.set_pcd(p, fp)
        assert ok.align()[0] == 0
        s = ok.get_state()
        return s["R"].astype(f32), s["T"].astype(f32)

    P = sr.Params
    ell = P.ell
    it_break = None
    R_next = T_next = None
    for k, row in enumerate(tr):
        assert f32(row["ell"]) == ell, f"iteration {k}: the ell schedule of cvo.cpp:810-812 reads {ell}, the oracle ran {row['ell']}"
        R, T = oracle_pose_before(k)
        if R_next is not None:                                                         # the pose update of iteration k - 1, from the oracle's inputs
            assert np.abs(R_next - R).max() <= 3e-7 and np.abs(T_next - T).max() <= 3e-7 * max(1.0, np.abs(T).max()), f"iteration {k - 1}: pose update"
        y, _, _ = sr.transform_cloud(R, T, p)                                          # cvo.cpp:770-771
        A, keep = sr.se_kernel(x, fx, y, fp, ell)                                      # cvo.cpp:122-184
        omega, v, nnz, fmags = sr.compute_flow(x, y, A)                                # cvo.cpp:187-236
        assert nnz == row["nnz"], f"iteration {k}: nnz(A) {nnz} vs the oracle's {row['nnz']}"
        # omega, v are sums over rows that cancel: a row's float32 sum differs in its last bit between two readings (`1/c*Ai*cross_xy` scales
        # the values before the product as it parses, the oracle after), so the scale is the terms' magnitudes, not the total
        got = np.concatenate([omega, v]).astype(np.float64); want = np.concatenate([row["omega"], row["v"]]).astype(np.float64)
        ftol = 3e-7 * fmags
        if assoc == "left":
            # with the plain association the transformed points themselves differ from the oracle's in their last bit, all of a row's neighbours the
            # same way: one ulp of a coordinate of metres, times the weights, in v; times |x| more in omega = sum a (x cross y) / c
            ulp_y = float(np.spacing(f32(np.abs(y).max()))); wsum = float(A.astype(np.float64).sum()) / float(P.c)
            ftol = ftol + ulp_y * wsum * np.array([np.abs(x).max()] * 3 + [1.0] * 3)
        assert np.all(np.abs(got - want) <= ftol + 1e-30), f"iteration {k}: omega, v {got} vs {want} (allowed {ftol})"
        # from here on with the oracle's own omega, v: the next stage is not charged with this stage's last bit
        omega_o, v_o = row["omega"].astype(f32), row["v"].astype(f32)
        BCDE, mags = sr.step_terms(x, y, A, keep, omega_o, v_o, ell)                   # cvo.cpp:239-315
        for q, nm in enumerate("BCDE"):
            if assoc == "unrolled":                                                    # the oracle's association: the same roundings term by term
                _close(BCDE[q], row["BCDE"][q], 1e-9, nm, k)
            else:                                                                      # the plain reading: the terms differ in their last float32 bit, and
                # x_i - y_j (cvo.cpp:286) by an ulp of y_j: relative to a difference of at most the radius that is ulp / radius
                rel = 1e-6 + float(np.spacing(f32(np.abs(y).max()))) / float(np.sqrt(sr.gates(ell)[1]))
                assert abs(BCDE[q] - row["BCDE"][q]) <= rel * mags[q] + 1e-30, f"iteration {k}: {nm} {BCDE[q]} vs {row['BCDE'][q]} (terms add up to {mags[q]} in magnitude, allowed {rel} of it)"
        step = sr.choose_step(row["BCDE"])                                             # cvo.cpp:317-333
        _close(step, row["step"], 1e-6, "step", k)
        wn = np.sqrt(np.sum(omega_o.astype(np.float64) ** 2)); vn = np.sqrt(np.sum(v_o.astype(np.float64) ** 2))
        if wn < P.eps and vn < P.eps:                                                  # cvo.cpp:782-786
            it_break = k
            assert row["dist"] == -1, f"iteration {k}: stop A fires here, the oracle went on"
            break
        dR, dT = sr.exp_sek3(omega_o, v_o, f32(row["step"]))                           # cvo.cpp:789-797
        T_next = (sr.matvec3(R, dT) + T).astype(f32)                                   # cvo.cpp:800
        R_next = sr.matmul3(R, dR).astype(f32)                                         # cvo.cpp:801
        dist = sr.dist_se3(dR, dT)                                                     # cvo.cpp:804
        # float32 rotations carry ~1e-7 of rounding, |logm| of them ~1e-7 of noise: relative agreement where the distance is well above that
        assert abs(dist - row["dist"]) <= 2e-3 * row["dist"] + 3e-7, f"iteration {k}: dist_se3 {dist} vs {row['dist']}"
        if row["dist"] < P.eps_2:
            it_break = k
            break
        ell = f32(0.10) if k > 2 else ell                                              # cvo.cpp:810-812
        ell = f32(0.06) if k > 9 else ell
        ell = f32(0.03) if k > 19 else ell
    assert it_break == len(tr) - 1 == st["iter"], "both readings stop at the same iteration"

    # final pose (cvo.cpp:817): transform = [R^T, -R^T T]
    _, Rt, t = sr.transform_cloud(R_next if R_next is not None else R, T_next if T_next is not None else T, p[:1])
    re, te = rot_trans_err(np.concatenate([Rt, t[:, None]], axis=1), st["transform"])
    assert re <= 1e-6 and te <= 1e-6, (re, te)

    # the scores at the ell the alignment left behind (Q1), on the oracle's final transform
    tf = st["transform"].reshape(3, 4)
    ym = (sr.matmul3(tf[:, :3], p.T).T + tf[:, 3][None, :]).astype(f32)                # cvo.cpp:485-487
    ell_f = f32(st["ell"])
    for (xa, fa, xb, fb, slot_a, tran, slot_b) in ((ym, fp, x, fx, oracle.SLOT_MOVING, tf, oracle.SLOT_FIXED),
                                                   (p, fp, x, fx, oracle.SLOT_MOVING, None, oracle.SLOT_FIXED),
                                                   (x, fx, x, fx, oracle.SLOT_FIXED, None, oracle.SLOT_FIXED)):
        val, num = sr.inner_product(xa, fa, xb, fb, ell_f)                             # cvo.cpp:388-459
        rc, (oval, onum, _) = o.function_inner_product(slot_a, tran, slot_b)
        assert rc == 0 and num == onum, (num, onum)
        assert abs(val - oval) <= 1e-6 * abs(oval) + 1e-12, (val, oval)
    H, inl = sr.hessian_raw(ym, fp, x, fx, ell_f)                                      # cvo.cpp:620-715
    rc, _, oinl, Hraw = o.se3_hessian(oracle.SLOT_MOVING, tf, oracle.SLOT_FIXED)
    assert rc == 0 and inl == oinl
    assert np.abs(H - Hraw).max() <= 1e-4 * np.abs(Hraw).max(), np.abs(H - Hraw).max() / np.abs(Hraw).max()
    assert np.abs(Hraw - Hraw.T).max() <= 1e-5 * np.abs(Hraw).max()                    # Blocks is symmetric by construction (cvo.cpp:701-704)


def test_second_reading_gates_match_the_survey_figures():
    # d2_thres(ell) = 0.446287 ell^2; d2_c_thres = 386 265
    s2, d2, d2c = sr.gates(f32(0.15))
    assert abs(float(d2) / 0.15 ** 2 - 0.446287) < 1e-4 and abs(float(d2c) - 386265) < 2


def test_second_reading_on_a_full_size_pair(oracle):
    """The same stages on the benchmark's shape (tests/golden/tum_pair_0.npz: 3 072 x 3 072 points, ~1e5 members of A at ell = 0.15): the first iterations and one at
    every later length-scale, each from the oracle's own pose at that iteration, with the oracle's association of 3-term sums."""
    g = np.load(os.path.join(GOLDEN, "tum_pair_0.npz"))
    x, fx, p, fp = g["fixed_xyz"], g["fixed_feat"], g["moving_xyz"], g["moving_feat"]
    o = oracle.OracleCvo(search=oracle.SEARCH_KDTREE, threads=4)
    o.set_pcd(x, fx); o.set_pcd(p, fp)
    rc, tr = o.align(trace_cap=400)
    assert rc == 0
    np.testing.assert_array_equal([r["nnz"] for r in tr], g["trace_nnz"])       # (the oracle is the one that wrote the fixture)
    old = sr.ASSOC3
    sr.ASSOC3 = "unrolled"
    try:
        for k in (0, 1, 5, 12, len(tr) - 2):
            pr = oracle.default_params(); pr.max_iter = k
            ok = oracle.OracleCvo(params=pr, search=oracle.SEARCH_KDTREE, threads=4)
            ok.set_pcd(x, fx); ok.set_pcd(p, fp)
            assert ok.align()[0] == 0
            s = ok.get_state()
            row = tr[k]; ell = f32(row["ell"])
            y, _, _ = sr.transform_cloud(s["R"].astype(f32), s["T"].astype(f32), p)
            A, keep = sr.se_kernel(x, fx, y, fp, ell)
            omega, v, nnz, fmags = sr.compute_flow(x, y, A)
            assert nnz == row["nnz"], f"iteration {k}: nnz(A) {nnz} vs the oracle's {row['nnz']}"
            got = np.concatenate([omega, v]).astype(np.float64); want = np.concatenate([row["omega"], row["v"]]).astype(np.float64)
            assert np.all(np.abs(got - want) <= 3e-7 * fmags + 1e-30), (k, got, want)
            BCDE, mags = sr.step_terms(x, y, A, keep, row["omega"].astype(f32), row["v"].astype(f32), ell)
            for q, nm in enumerate("BCDE"):
                # (up to 1e5 terms that cancel to a thousandth of their magnitudes, added in float64 in two different orders)
                assert abs(BCDE[q] - row["BCDE"][q]) <= 1e-9 * abs(row["BCDE"][q]) + 1e-10 * mags[q], f"iteration {k}: {nm} {BCDE[q]} vs {row['BCDE'][q]}"
            _close(sr.choose_step(row["BCDE"]), row["step"], 1e-6, "step", k)
    finally:
        sr.ASSOC3 = old
