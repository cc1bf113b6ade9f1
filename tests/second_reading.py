"""A SECOND, independent reading of the hot path -- dense numpy, written from the reference's source text alone
(thirdparty/cvo/src/cvo.cpp:106-110,122-341,388-459,620-715,763-813 and thirdparty/cvo/src/LieGroup.cpp:20-27,159-186), without
looking at oracle/cvo_oracle.cpp or the kernels.  TEST INFRASTRUCTURE: it exists so that the C++ oracle (and through it the HIP
path) is checked against a restatement by other hands, in another language, with another evaluation strategy:

* no neighbour search at all: every (i, j) pair is formed as an N x M array and masked (the reference's radius search is exact,
  nanoflann.hpp:249-253,375-416, so the masked dense set is the same set);
* element arithmetic in numpy float32 arrays (numpy never contracts a multiply and an add), `exp` in float64 because of the
  reference's `2.0` literals (cvo.cpp:172-173), cross-row sums in float64 (cvo.cpp:194-195,226-230,268-271);
* the step as the roots numpy finds for the cubic (its own companion-matrix eigenvalues, cvo.cpp:76-92), the pose update with
  Exp_SEK3 as LieGroup.cpp writes it, the stop distance through scipy's `logm` (cvo.cpp:94-104).

Where the C++ text leaves the association of a float sum to Eigen (3- and 5-term reductions, the row products of
cvo.cpp:222-223) this file adds left to right, and it applies `1/c` to the row's values BEFORE the product, as
`1/c*Ai*cross_xy` parses; the oracle's choices may differ in the last bit of a float, which is what the tolerances of
tests/test_oracle_second_reading.py allow for (and no more).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
f64 = np.float64


def _f(x):
    return np.asarray(x, dtype=f32)


# How a three-term float sum associates where the C++ text leaves it to Eigen (fixed-size products and 1x3 * 3x1 row products):
# "left" = (t0 + t1) + t2, the plain reading; "unrolled" = t0 + (t1 + t2), what Eigen's unrolled reduction of a size-3 expression
# does (redux_novec_unroller splits Length 3 into 1 + 2) and what the oracle assumes.  The test runs both: with "unrolled" the
# line-search sums must agree with the oracle's to 1e-9 of their size (same formulas, same roundings), with "left" to the float32
# rounding of the terms (1e-6 of the sum of their magnitudes) -- the distance between two legitimate readings.
ASSOC3 = "left"


def sum3(t0, t1, t2):
    return (t0 + t1) + t2 if ASSOC3 == "left" else t0 + (t1 + t2)


class Params:
    # cvo.cpp:35-51 (the constructor's initialiser list)
    ell = f32(0.15); sigma = f32(0.1); sp_thres = f32(8e-3); c = f32(7.0); d = f32(7.0)
    c_ell = f32(200); c_sigma = f32(1)
    max_iter = 2000; min_step = f32(2 * 1.0e-1); eps = f32(5 * 1.0e-5); eps_2 = f32(1.0e-5)


def skew(w):
    # LieGroup.cpp:20-27
    w = _f(w)
    z = f32(0)
    return np.array([[z, -w[2], w[1]], [w[2], z, -w[0]], [-w[1], w[0], z]], dtype=f32)


def matmul3(A, B):
    """3x3 (or 3x3 by 3xK) product in float32, terms added left to right."""
    A = _f(A); B = _f(B)
    return sum3(A[:, 0:1] * B[0:1, :], A[:, 1:2] * B[1:2, :], A[:, 2:3] * B[2:3, :])


def matvec3(A, x):
    A = _f(A); x = _f(x)
    return sum3(A[:, 0] * x[0], A[:, 1] * x[1], A[:, 2] * x[2])


def gates(ell, P=Params):
    """cvo.cpp:125-126: float d2_thres = -2.0*l*l*log(sp_thres/s2); the log of a float is the float overload, the products with the
    double literal are double, the result is stored in a float."""
    s2 = f32(P.sigma * P.sigma)
    d2_thres = f32(-2.0 * f64(ell) * f64(ell) * f64(np.log(f32(P.sp_thres / s2))))
    d2_c_thres = f32(-2.0 * f64(P.c_ell) * f64(P.c_ell) * f64(np.log(f32(f32(P.sp_thres / P.c_sigma) / P.c_sigma))))
    return s2, d2_thres, d2_c_thres


def pair_arrays(xa, fa, xb, fb):
    """All N x M squared distances (three differences squared, added in order) and squared feature distances (five)."""
    dx = xa[:, None, 0] - xb[None, :, 0]; dy = xa[:, None, 1] - xb[None, :, 1]; dz = xa[:, None, 2] - xb[None, :, 2]
    d2 = (dx * dx + dy * dy) + dz * dz
    d2c = np.zeros_like(d2)
    for ch in range(5):
        e = fa[ch][:, None] - fb[ch][None, :]
        d2c = d2c + e * e
    return d2, d2c


def se_kernel(x, fx, y, fy, ell, P=Params):
    """cvo.cpp:122-184 as a dense N x M array: A[i, j] = a if the pair is kept, else 0; the boolean mask of kept pairs."""
    s2, d2_thres, d2_c_thres = gates(ell, P)
    d2, d2c = pair_arrays(x, fx, y, fy)
    near = (d2 < d2_thres) & (d2c < d2_c_thres)
    k = (f64(s2) * np.exp(-d2.astype(f64) / (2.0 * f64(ell) * f64(ell)))).astype(f32)                       # cvo.cpp:172
    ck = (f64(f32(P.c_sigma * P.c_sigma)) * np.exp(-d2c.astype(f64) / (2.0 * f64(P.c_ell) * f64(P.c_ell)))).astype(f32)   # :173
    a = ck * k                                                                                                  # :174
    keep = near & (a > P.sp_thres)                                                                              # :175
    return np.where(keep, a, f32(0)), keep


def row_sums_in_column_order(W, V):
    """sum_j W[i, j] * V[i, j, :] in float32, j ascending (the CSR order of setFromTriplets, cvo.cpp:182, 209-220)."""
    n, m = W.shape
    acc = np.zeros((n, 3), f32)
    for j in range(m):
        col = W[:, j]
        if not col.any():
            continue
        acc = acc + col[:, None] * V[:, j, :]
    return acc


def compute_flow(x, y, A, P=Params):
    """cvo.cpp:187-236.  Returns omega, v (float32), nnz, and the sums of the magnitudes of all terms of each component."""
    xx = np.broadcast_to(x[:, None, :], (x.shape[0], y.shape[0], 3))
    yy = np.broadcast_to(y[None, :, :], (x.shape[0], y.shape[0], 3))
    cross = np.stack([xx[..., 1] * yy[..., 2] - xx[..., 2] * yy[..., 1],
                      xx[..., 2] * yy[..., 0] - xx[..., 0] * yy[..., 2],
                      xx[..., 0] * yy[..., 1] - xx[..., 1] * yy[..., 0]], axis=-1).astype(f32)                  # :216
    diff = (yy - xx).astype(f32)                                                                                # :217
    inv_c = f32(f32(1) / P.c); inv_d = f32(f32(1) / P.d)
    part_w = row_sums_in_column_order(inv_c * A, cross)                                                        # :222  (1/c*Ai)*cross_xy
    part_v = row_sums_in_column_order(inv_d * A, diff)                                                         # :223
    omega = part_w.astype(f64).sum(axis=0)                                                                     # :226-230 (any order in the reference)
    v = part_v.astype(f64).sum(axis=0)
    # the scale the float32 rounding of the row sums lives on: the magnitudes of all their terms
    mags = np.concatenate([np.abs((inv_c * A)[..., None] * cross).astype(f64).sum(axis=(0, 1)), np.abs((inv_d * A)[..., None] * diff).astype(f64).sum(axis=(0, 1))])
    return omega.astype(f32), v.astype(f32), int(np.count_nonzero(A)), mags                                     # :234-235


def step_terms(x, y, A, keep, omega, v, ell):
    """cvo.cpp:239-315: B, C, D, E (float64), and the sums of the magnitudes of their terms (the scale their rounding lives on)."""
    O = skew(omega)
    O2 = matmul3(O, O); O3 = matmul3(O2, O); O4 = matmul3(O3, O)
    yT = y.T
    xiz = (np.cross(np.broadcast_to(omega, y.shape).astype(f32), y).astype(f32) + v[None, :]).astype(f32)      # :254
    xi2z = (matmul3(O2, yT) + matvec3(O, v)[:, None]).T.astype(f32)                                            # :255-256
    xi3z = (matmul3(O3, yT) + matvec3(O2, v)[:, None]).T.astype(f32)                                           # :257-258
    xi4z = (matmul3(O4, yT) + matvec3(O3, v)[:, None]).T.astype(f32)                                           # :259-260

    def dot(a, b):
        return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]
    normxiz2 = dot(xiz, xiz)                                                                                    # :261
    xiz_dot_xi2z = -dot(xiz, xi2z)                                                                              # :262
    epsil_const = dot(xi2z, xi2z) + f32(2) * dot(xiz, xi3z)                                                     # :263
    temp_coef = f32(1 / (2.0 * f64(ell) * f64(ell)))                                                            # :267
    ii, jj = np.nonzero(keep)                                                                                   # row-major = CSR order
    diff = (x[ii] - y[jj]).astype(f32)                                                                          # :286

    def rowdot(rows, dxy):                                                                                      # (1x3 row) * (3x1 vector)
        return sum3(rows[:, 0] * dxy[:, 0], rows[:, 1] * dxy[:, 1], rows[:, 2] * dxy[:, 2])
    # Eigen applies a double literal times a float scalar to a float expression as a float scalar (the expression's scalar type)
    beta = rowdot(f32(-2.0 * f64(temp_coef)) * xiz[jj], diff)                                                   # :288
    gamma = f32(-temp_coef) * (normxiz2[jj] + rowdot(f32(2.0) * xi2z[jj], diff))                                # :290-291
    delta = f32(2.0 * f64(temp_coef)) * (xiz_dot_xi2z[jj] + rowdot(-xi3z[jj], diff))                            # :293-294
    epsil = f32(-temp_coef) * (epsil_const[jj] + rowdot(f32(2.0) * xi4z[jj], diff))                             # :296-297
    a = A[ii, jj]
    b64, g64 = beta.astype(f64), gamma.astype(f64)
    B = (a * beta).astype(f64)                                                                                  # :301
    C = a.astype(f64) * (g64 + (beta * beta).astype(f64) / 2.0)                                                 # :302
    D = a.astype(f64) * ((delta + beta * gamma).astype(f64) + (beta * beta * beta).astype(f64) / 6.0)           # :303
    E = a.astype(f64) * ((epsil + beta * delta).astype(f64) + 1 / 2.0 * b64 * b64 * g64                         # :304-305
                         + 1 / 2.0 * g64 * g64 + 1 / 24.0 * b64 * b64 * b64 * b64)
    return np.array([B.sum(), C.sum(), D.sum(), E.sum()]), np.array([np.abs(B).sum(), np.abs(C).sum(), np.abs(D).sum(), np.abs(E).sum()])


def choose_step(BCDE, P=Params):
    """cvo.cpp:317-333 with numpy's roots of the float32 coefficient vector."""
    B, C, D, E = (f32(t) for t in BCDE)
    coef = np.array([f32(4.0 * f64(E)), f32(3.0 * f64(D)), f32(2.0 * f64(C)), B], dtype=f32)                    # :318
    best = None
    with np.errstate(all="ignore"):
        mono = (coef / coef[0]).astype(f32)                                                                     # :86 (float division)
        if np.all(np.isfinite(mono)):
            for r in np.roots(mono.astype(f64)):
                if r.imag == 0 and f32(r.real) > 0 and (best is None or f32(r.real) < best):                   # :325-327
                    best = f32(r.real)
    step = P.min_step if best is None else best                                                                 # :330
    return f32(0.8) if step > 0.8 else f32(step)                                                                # :333


def exp_sek3(omega, v, dt):
    """LieGroup.cpp:159-186, K = 1.  Returns dR (3x3), dT (3)."""
    w = _f(omega); dt = f32(dt)
    theta = f32(np.sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]))
    I = np.eye(3, dtype=f32)
    if theta < f32(1e-6):                                                                                       # TOLERANCE, LieGroup.cpp:14
        R = I.copy(); Jl = I.copy()
    else:
        A = skew(w)
        theta2 = f32(theta * theta)
        stheta = f32(np.sin(f32(dt * theta))); ctheta = f32(np.cos(f32(dt * theta)))
        omc = f32((f32(1) - ctheta) / theta2)
        A2 = matmul3(A, A)
        R = (I + f32(stheta / theta) * A) + omc * A2
        Jl = (dt * I + omc * A) + f32(f32(dt * theta - stheta) / f32(theta2 * theta)) * A2
    return R.astype(f32), matvec3(Jl, v).astype(f32)


def dist_se3(dR, dT):
    """cvo.cpp:94-104: Frobenius norm of the matrix logarithm (scipy, float64)."""
    from scipy.linalg import logm
    X = np.eye(4); X[:3, :3] = dR; X[:3, 3] = dT
    return float(np.linalg.norm(np.real(logm(X))))


def transform_cloud(R, T, p):
    """update_tf + transform_pcd, cvo.cpp:106-110,336-341: y_j = R^T p_j - R^T T."""
    Rt = _f(R).T
    t = -matvec3(Rt, T)
    return ((matmul3(Rt, p.T)).T + t[None, :]).astype(f32), Rt.astype(f32), t.astype(f32)


def inner_product(xa, fa, xb, fb, ell, P=Params):
    """cvo.cpp:388-459: (sum of k*ck over the pairs inside both gates, their count; count 0 reads 1)."""
    s2 = f32(P.sigma * P.sigma)
    d2_thres = f32(-2.0 * f64(ell) * f64(ell) * f64(np.log(f32(f32(P.sp_thres / P.sigma) / P.sigma))))         # :395
    _, _, d2_c_thres = gates(ell, P)
    d2, d2c = pair_arrays(xa, fa, xb, fb)
    inside = (d2 < d2_thres) & (d2c < d2_c_thres)
    k = (f64(s2) * np.exp(-d2.astype(f64) / (2.0 * f64(ell) * f64(ell)))).astype(f32)
    ck = (f64(f32(P.c_sigma * P.c_sigma)) * np.exp(-d2c.astype(f64) / (2.0 * f64(P.c_ell) * f64(P.c_ell)))).astype(f32)
    a = (ck * k)[inside]
    n = int(inside.sum())
    return float(a.astype(f64).sum()), (n if n else 1)


def hessian_raw(xa, fa, xb, fb, ell, P=Params):
    """cvo.cpp:620-715: the 6x6 sum before the scaling and the eigenvalue shift, and the pair count.  Terms in float32 as written
    (the `0.5*` products are double, cvo.cpp:673-675), the sum over pairs in float64 here (the reference adds in float32 in an order of
    its threads' choosing, cvo.cpp:706-713)."""
    s2 = f32(P.sigma * P.sigma)
    d2_thres = f32(-2.0 * f64(ell) * f64(ell) * f64(np.log(f32(f32(P.sp_thres / P.sigma) / P.sigma))))
    _, _, d2_c_thres = gates(ell, P)
    d2, d2c = pair_arrays(xa, fa, xb, fb)
    ii, jj = np.nonzero((d2 < d2_thres) & (d2c < d2_c_thres))
    pa, pb = xa[ii], xb[jj]
    k = (f64(s2) * np.exp(-d2[ii, jj].astype(f64) / (2.0 * f64(ell) * f64(ell)))).astype(f32)                  # :661
    cdot = np.zeros(len(ii), f32)
    for ch in range(5):
        cdot = cdot + fa[ch][ii] * fb[ch][jj]                                                                   # :662
    cr = np.cross(pa, pb).astype(f32)                                                                           # :663
    il2 = f32(f32(1) / f32(ell * ell))
    n = len(ii)
    Bl = np.zeros((n, 6, 6), f32)
    dot1 = pa[:, 1] * pb[:, 1] + pa[:, 2] * pb[:, 2]; dot2 = pa[:, 0] * pb[:, 0] + pa[:, 2] * pb[:, 2]; dot3 = pa[:, 0] * pb[:, 0] + pa[:, 1] * pb[:, 1]
    A = np.zeros((n, 3, 3), f32)
    A[:, 0, 0] = il2 * cr[:, 0] * cr[:, 0] - dot1; A[:, 1, 1] = il2 * cr[:, 1] * cr[:, 1] - dot2; A[:, 2, 2] = il2 * cr[:, 2] * cr[:, 2] - dot3   # :670-672

    def offd(p, q):
        return ((il2 * cr[:, p] * cr[:, q]).astype(f64) + 0.5 * (pa[:, p] * pb[:, q] + pa[:, q] * pb[:, p]).astype(f64)).astype(f32)
    A[:, 0, 1] = A[:, 1, 0] = offd(0, 1); A[:, 0, 2] = A[:, 2, 0] = offd(0, 2); A[:, 1, 2] = A[:, 2, 1] = offd(1, 2)     # :673-675
    db = (pb - pa).astype(f32)                                                                                  # :679
    Cm = np.zeros((n, 3, 3), f32)
    for q in range(3):
        Cm[:, q, q] = il2 * cr[:, q] * db[:, q]                                                                 # :680-682
    Cm[:, 1, 0] = pa[:, 2] + il2 * db[:, 1] * cr[:, 0]; Cm[:, 2, 0] = -pa[:, 1] + il2 * db[:, 2] * cr[:, 0]     # :683-684
    Cm[:, 0, 1] = -pa[:, 2] + il2 * db[:, 0] * cr[:, 1]; Cm[:, 2, 1] = pa[:, 0] + il2 * db[:, 2] * cr[:, 1]     # :685-686
    Cm[:, 0, 2] = pa[:, 1] + il2 * db[:, 0] * cr[:, 2]; Cm[:, 1, 2] = -pa[:, 0] + il2 * db[:, 1] * cr[:, 2]     # :687-688
    Dm = np.zeros((n, 3, 3), f32)
    for p in range(3):
        for q in range(3):
            Dm[:, p, q] = il2 * db[:, p] * db[:, q] - (f32(1) if p == q else f32(0))                            # :692-697
    Bl[:, :3, :3] = A; Bl[:, :3, 3:] = np.transpose(Cm, (0, 2, 1)); Bl[:, 3:, :3] = Cm; Bl[:, 3:, 3:] = Dm      # :701-704
    w = (il2 * cdot * k).astype(f32)                                                                            # :707
    H = (w[:, None, None] * Bl).astype(f64).sum(axis=0)
    return H, n
