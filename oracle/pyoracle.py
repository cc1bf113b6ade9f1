"""ctypes binding of oracle/libcvo_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcvo_oracle.so")
REF_LIB_PATH = os.path.join(HERE, "_ref", "libref_nanoflann.so")
_SRCS = [os.path.join(HERE, f) for f in ("cvo_oracle.cpp", "pcd_oracle.cpp", "cvo_oracle.h", "ref_noise.hpp", "Makefile")]


def _cpu_tag() -> str:
    """-march=native binds the fast build to the host's CPU: the file name carries a hash of the CPU's flag list, so a
    library built on another machine (this repository travels between boxes with its built .so files) is never loaded."""
    import hashlib
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line; break
    except OSError:
        pass
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def fast_lib_path() -> str:
    return os.path.join(HERE, f"libcvo_oracle_fast_{_cpu_tag()}.so")


def _stale(path: str) -> bool:
    return (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in _SRCS)


def build(force: bool = False, fast: bool = True) -> None:
    """Compile the oracle: the parity build (-O2 -ffp-contract=off), the fast build of the same source
    (-O3 -march=native -ffp-contract=fast: the FMA member of the noise envelope and the timed CPU baseline),
    and oracle/_ref when /root/reference exists."""
    if force or _stale(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-s", LIB_PATH])
    if fast and (force or _stale(fast_lib_path())):
        subprocess.check_call(["make", "-C", HERE, "-s", "fast", f"FAST_TAG={_cpu_tag()}"])
    if os.path.exists("/root/reference/thirdparty/cvo/thirdparty/nanoflann.hpp") and (force or not os.path.exists(REF_LIB_PATH)):
        subprocess.check_call(["make", "-C", HERE, "-s", "ref"])


class Params(C.Structure):
    _fields_ = [("ell", C.c_float), ("sigma", C.c_float), ("sp_thres", C.c_float), ("c", C.c_float), ("d", C.c_float),
                ("c_ell", C.c_float), ("c_sigma", C.c_float), ("max_iter", C.c_int), ("min_step", C.c_float),
                ("eps", C.c_float), ("eps_2", C.c_float)]


class InnP(C.Structure):
    _fields_ = [("value", C.c_float), ("num", C.c_int), ("num_e", C.c_int)]


class AdaptiveParams(C.Structure):
    _fields_ = [("ell_init", C.c_float), ("ell_min", C.c_float), ("ell_max", C.c_float), ("dl_step", C.c_float), ("sigma", C.c_float),
                ("sp_thres", C.c_float), ("c", C.c_float), ("d", C.c_float), ("c_ell", C.c_float), ("c_sigma", C.c_float),
                ("max_iter", C.c_int), ("min_step", C.c_float), ("eps", C.c_float), ("eps_2", C.c_float)]


class AdaptiveRow(C.Structure):
    _fields_ = [("omega", C.c_float * 3), ("v", C.c_float * 3), ("dl", C.c_float), ("ell", C.c_float), ("step", C.c_float),
                ("nnz_xy", C.c_int), ("nnz_xx", C.c_int), ("nnz_yy", C.c_int)]


class TraceRow(C.Structure):
    _fields_ = [("omega", C.c_float * 3), ("v", C.c_float * 3), ("nnz", C.c_int), ("B", C.c_double), ("C", C.c_double),
                ("D", C.c_double), ("E", C.c_double), ("step", C.c_float), ("ell", C.c_float), ("dist", C.c_float)]


SEARCH_BRUTE, SEARCH_KDTREE = 0, 1
SLOT_FIXED, SLOT_MOVING, SLOT_PREVIOUS = 0, 1, 2
VAR_SHUFFLE, VAR_F32_ROOTS, VAR_F32_LOGM = 1, 2, 4      # cvo_oracle.h: reference-noise variants

_libs = {}


def lib(flavor: str = "parity"):
    """flavor "parity": the un-fused -O2 build every parity test compares with; "fast": the optimising build."""
    if flavor not in _libs:
        path = LIB_PATH if flavor == "parity" else fast_lib_path()
        if _stale(path):
            build(fast=(flavor != "parity"))
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float); dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_create.argtypes = [C.POINTER(Params)]; L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_exec.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_pcd.argtypes = [C.c_void_p, fp, fp, C.c_int]
        L.orc_align.argtypes = [C.c_void_p, C.POINTER(TraceRow), C.c_int, ip]
        L.orc_match.argtypes = [C.c_void_p, fp, fp, C.c_int, dp]
        for name in ("orc_update_fixed_pcd", "orc_update_previous_pcd"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_reset_keyframe.argtypes = [C.c_void_p, fp]
        L.orc_reset_transform.argtypes = [C.c_void_p, fp]
        L.orc_reset_initial.argtypes = [C.c_void_p, fp, fp]
        L.orc_function_inner_product.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, C.POINTER(InnP)]
        L.orc_se3_hessian.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, dp, ip, dp]
        L.orc_compute_innerproduct.argtypes = [C.c_void_p, C.POINTER(InnP), C.POINTER(InnP), dp, fp, ip,
                                               C.POINTER(InnP), C.POINTER(InnP), fp]
        L.orc_compute_innerproduct_lc.argtypes = [C.c_void_p] + [C.POINTER(InnP)] * 4 + [dp, fp, fp, fp, fp, ip, ip,
                                                                                          C.POINTER(InnP), C.POINTER(InnP), fp]
        L.orc_get_state.argtypes = [C.c_void_p, fp, fp, fp, fp, ip, ip, ip, ip]
        L.orc_set_state.argtypes = [C.c_void_p, fp, fp, C.c_float]
        L.orc_get_accum.argtypes = [C.c_void_p, fp, fp]
        L.orc_get_init.argtypes = [C.c_void_p]
        L.orc_get_timing.argtypes = [C.c_void_p, dp]
        L.orc_flow_once.argtypes = [C.c_void_p, fp, fp, ip, dp, fp, ip, ip, fp, C.c_int]
        L.orc_cubic_step.argtypes = [C.c_float] * 5; L.orc_cubic_step.restype = C.c_float
        L.orc_exp_sek3.argtypes = [fp, fp, C.c_float, fp, fp]
        L.orc_dist_se3.argtypes = [fp, fp]; L.orc_dist_se3.restype = C.c_float
        L.orc_hessian_regularize.argtypes = [fp, C.c_int, dp]
        L.orc_radius_search.argtypes = [fp, C.c_int, fp, C.c_float, ip, fp, C.c_int, C.c_int]
        L.orc_pair_values.argtypes = [C.POINTER(Params), C.c_float, C.c_int, fp, fp, fp, fp, fp]
        L.orc_libm_f32.argtypes = [C.c_int, C.c_int, fp, fp]
        L.orc_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_ulonglong]
        L.orc_cubic_step_f32eig.argtypes = [C.c_float] * 5; L.orc_cubic_step_f32eig.restype = C.c_float
        L.orc_dist_se3_f32logm.argtypes = [fp, fp]; L.orc_dist_se3_f32logm.restype = C.c_float
        L.orc_test_eigenvalues.argtypes = [C.c_int, dp, dp, dp, C.c_int]
        L.orc_test_logm.argtypes = [C.c_int, dp, dp, C.c_int]
        _libs[flavor] = L
    return _libs[flavor]


def ref_lib():
    """oracle/_ref/libref_nanoflann.so (the reference's own nanoflann), or None."""
    if not os.path.exists(REF_LIB_PATH):
        return None
    L = C.CDLL(REF_LIB_PATH)
    fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
    L.ref_radius_search.argtypes = [fp, C.c_int, fp, C.c_int, C.c_float, ip, ip, fp, C.c_int]
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def default_params() -> Params:
    p = Params(); lib().orc_default_params(C.byref(p)); return p


class OracleCvo:
    """The reference's `cvo::cvo` (cvo.hpp:82-282) with pcd_generator output handed in."""

    def __init__(self, params: Params | None = None, search=SEARCH_BRUTE, threads=1, flavor="parity", variant=0, shuffle_seed=0):
        self.L = lib(flavor)
        self.params = params or default_params()
        self.h = C.c_void_p(self.L.orc_create(C.byref(self.params)))
        self.L.orc_set_exec(self.h, search, threads)
        if variant:
            self.L.orc_set_variant(self.h, int(variant), int(shuffle_seed))

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h); self.h = None
        except Exception:
            pass

    def set_exec(self, search, threads):
        self.L.orc_set_exec(self.h, search, threads)

    def set_pcd(self, xyz, feat):
        x, xp = _f(xyz); f, fp_ = _f(feat)
        assert x.shape[1] == 3 and f.shape[0] == 5 and f.shape[1] == x.shape[0]
        return self.L.orc_set_pcd(self.h, xp, fp_, x.shape[0])

    def align(self, trace_cap=0):
        if trace_cap:
            rows = (TraceRow * trace_cap)(); n = C.c_int(0)
            rc = self.L.orc_align(self.h, rows, trace_cap, C.byref(n))
            tr = [dict(omega=np.array(r.omega[:], np.float32), v=np.array(r.v[:], np.float32), nnz=r.nnz,
                       BCDE=np.array([r.B, r.C, r.D, r.E]), step=r.step, ell=r.ell, dist=r.dist) for r in rows[: n.value]]
            return rc, tr
        return self.L.orc_align(self.h, None, 0, None), None

    def match(self, xyz, feat):
        x, xp = _f(xyz); f, fp_ = _f(feat)
        out = np.zeros(12, np.float64)
        rc = self.L.orc_match(self.h, xp, fp_, x.shape[0], out.ctypes.data_as(C.POINTER(C.c_double)))
        return rc, out.reshape(3, 4)

    def update_fixed_pcd(self): self.L.orc_update_fixed_pcd(self.h)
    def update_previous_pcd(self): self.L.orc_update_previous_pcd(self.h)

    def reset_keyframe(self, odom):
        o, op = _f(np.asarray(odom).reshape(12)); self.L.orc_reset_keyframe(self.h, op)

    def reset_transform(self, odom):
        o, op = _f(np.asarray(odom).reshape(12)); self.L.orc_reset_transform(self.h, op)

    def reset_initial(self, odom):
        o, op = _f(np.asarray(odom).reshape(12)); out = np.zeros(12, np.float32)
        self.L.orc_reset_initial(self.h, op, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out.reshape(3, 4)

    def function_inner_product(self, slot_a, tran_a, slot_b):
        r = InnP()
        tp = None
        if tran_a is not None:
            t, tp = _f(np.asarray(tran_a).reshape(12))
        rc = self.L.orc_function_inner_product(self.h, slot_a, tp, slot_b, C.byref(r))
        return rc, (r.value, r.num, r.num_e)

    def se3_hessian(self, slot_a, tran_a, slot_b):
        H = np.zeros(36); Hraw = np.zeros(36); inl = C.c_int(0)
        tp = None
        if tran_a is not None:
            t, tp = _f(np.asarray(tran_a).reshape(12))
        dp = C.POINTER(C.c_double)
        rc = self.L.orc_se3_hessian(self.h, slot_a, tp, slot_b, H.ctypes.data_as(dp), C.byref(inl), Hraw.ctypes.data_as(dp))
        return rc, H.reshape(6, 6), inl.value, Hraw.reshape(6, 6)

    def compute_innerproduct(self, tran):
        pre, post, fx, mv = InnP(), InnP(), InnP(), InnP()
        H = np.zeros(36); inl = C.c_int(0); cos = C.c_float(0)
        t, tp = _f(np.asarray(tran).reshape(12))
        rc = self.L.orc_compute_innerproduct(self.h, C.byref(pre), C.byref(post), H.ctypes.data_as(C.POINTER(C.c_double)), tp,
                                             C.byref(inl), C.byref(fx), C.byref(mv), C.byref(cos))
        tup = lambda r: (r.value, r.num, r.num_e)
        return rc, dict(inn_pre=tup(pre), inn_post=tup(post), post_hessian=H.reshape(6, 6), inliers=inl.value,
                        inn_fixed_pcd=tup(fx), inn_moving_pcd=tup(mv), cos_angle=cos.value)

    def compute_innerproduct_lc(self, prior_tran, lc_prior_tran, lc_prior_tran_2, lc_tran):
        prior, lcp, lcpre, lcpost, fx, mv = (InnP() for _ in range(6))
        H = np.zeros(36); i1 = C.c_int(0); i2 = C.c_int(0); cos = C.c_float(0)
        keep = [_f(np.asarray(t).reshape(12)) for t in (prior_tran, lc_prior_tran, lc_prior_tran_2, lc_tran)]
        rc = self.L.orc_compute_innerproduct_lc(self.h, C.byref(prior), C.byref(lcp), C.byref(lcpre), C.byref(lcpost),
                                                H.ctypes.data_as(C.POINTER(C.c_double)), keep[0][1], keep[1][1], keep[2][1], keep[3][1],
                                                C.byref(i1), C.byref(i2), C.byref(fx), C.byref(mv), C.byref(cos))
        tup = lambda r: (r.value, r.num, r.num_e)
        return rc, dict(inn_prior=tup(prior), inn_lc_prior=tup(lcp), inn_lc_pre=tup(lcpre), inn_lc_post=tup(lcpost),
                        post_hessian=H.reshape(6, 6), inliers_svd=i1.value, inliers_pnpransac=i2.value,
                        inn_fixed_pcd=tup(fx), inn_moving_pcd=tup(mv), cos_angle=cos.value)

    def get_state(self):
        R = np.zeros(9, np.float32); T = np.zeros(3, np.float32); tf = np.zeros(12, np.float32)
        ell = C.c_float(0); it = C.c_int(0); nnz = C.c_int(0); nf = C.c_int(0); nm = C.c_int(0)
        fp = C.POINTER(C.c_float)
        self.L.orc_get_state(self.h, R.ctypes.data_as(fp), T.ctypes.data_as(fp), C.byref(ell), tf.ctypes.data_as(fp),
                             C.byref(it), C.byref(nnz), C.byref(nf), C.byref(nm))
        return dict(R=R.reshape(3, 3), T=T, ell=ell.value, transform=tf.reshape(3, 4), iter=it.value, A_nonzero=nnz.value,
                    num_fixed=nf.value, num_moving=nm.value)

    def get_timing(self):
        t = np.zeros(4); self.L.orc_get_timing(self.h, t.ctypes.data_as(C.POINTER(C.c_double)))
        return dict(kdtree_build=t[0], search_and_kernel=t[1], csr_assembly=t[2], sparse_sweeps=t[3])

    def set_state(self, R, T, ell):
        r, rp = _f(np.asarray(R).reshape(9)); t, tp = _f(np.asarray(T).reshape(3))
        self.L.orc_set_state(self.h, rp, tp, float(ell))

    def flow_once(self, want_csr=False, csr_cap=0):
        om = np.zeros(3, np.float32); v = np.zeros(3, np.float32); nnz = C.c_int(0); bcde = np.zeros(4); step = C.c_float(0)
        fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int)
        st = self.get_state()
        if want_csr:
            rowptr = np.zeros(st["num_fixed"] + 1, np.int32); col = np.zeros(csr_cap, np.int32); val = np.zeros(csr_cap, np.float32)
            rc = self.L.orc_flow_once(self.h, om.ctypes.data_as(fp), v.ctypes.data_as(fp), C.byref(nnz),
                                      bcde.ctypes.data_as(C.POINTER(C.c_double)), C.byref(step),
                                      rowptr.ctypes.data_as(ip), col.ctypes.data_as(ip), val.ctypes.data_as(fp), csr_cap)
            return rc, dict(omega=om, v=v, nnz=nnz.value, BCDE=bcde, step=step.value, rowptr=rowptr, col=col[: nnz.value], val=val[: nnz.value])
        rc = self.L.orc_flow_once(self.h, om.ctypes.data_as(fp), v.ctypes.data_as(fp), C.byref(nnz),
                                  bcde.ctypes.data_as(C.POINTER(C.c_double)), C.byref(step), None, None, None, 0)
        return rc, dict(omega=om, v=v, nnz=nnz.value, BCDE=bcde, step=step.value)


def adaptive_default_params() -> AdaptiveParams:
    p = AdaptiveParams(); lib().orc_adaptive_default_params(C.byref(p)); return p


def adaptive_align(fixed_xyz, fixed_feat, moving_xyz, moving_feat, params: AdaptiveParams | None = None, R=None, T=None, trace_cap=0,
                   search=SEARCH_KDTREE, threads=1):
    """acvo::align (adaptive_cvo.cpp:490-555) from a fresh object: dict(transform, R, T, ell, iter, trace)."""
    L = lib(); L.orc_adaptive_align.restype = C.c_int
    p = params or adaptive_default_params()
    fx, fxp = _f(fixed_xyz); ff, ffp = _f(fixed_feat); mx, mxp = _f(moving_xyz); mf, mfp = _f(moving_feat)
    Rb = np.ascontiguousarray(np.eye(3) if R is None else R, np.float32).reshape(9).copy(); Tb = np.ascontiguousarray(np.zeros(3) if T is None else T, np.float32).copy()
    tf = np.zeros(12, np.float32); ell = C.c_float(0); it = C.c_int(-1); n = C.c_int(0)
    rows = (AdaptiveRow * max(1, trace_cap))()
    fp = C.POINTER(C.c_float)
    rc = L.orc_adaptive_align(C.byref(p), fxp, ffp, fx.shape[0], mxp, mfp, mx.shape[0], Rb.ctypes.data_as(fp), Tb.ctypes.data_as(fp), C.byref(ell),
                              tf.ctypes.data_as(fp), C.byref(it), rows, trace_cap, C.byref(n), search, threads)
    tr = [dict(omega=np.array(r.omega[:], np.float32), v=np.array(r.v[:], np.float32), dl=r.dl, ell=r.ell, step=r.step,
               nnz_xy=r.nnz_xy, nnz_xx=r.nnz_xx, nnz_yy=r.nnz_yy) for r in rows[: n.value]]
    return rc, dict(transform=tf.reshape(3, 4), R=Rb.reshape(3, 3), T=Tb, ell=ell.value, iter=it.value, trace=tr)


def pair_values(d2, d2c, ell, params: Params | None = None):
    """cvo.cpp:166-175 on given squared distances: (a, k, ck) float32 arrays (a = 0 for a non-member)."""
    d, dp_ = _f(d2); c, cp = _f(d2c)
    assert d.shape == c.shape and d.ndim == 1
    a = np.zeros_like(d); k = np.zeros_like(d); ck = np.zeros_like(d)
    fp = C.POINTER(C.c_float)
    p = params or default_params()
    lib().orc_pair_values(C.byref(p), float(ell), d.shape[0], dp_, cp, a.ctypes.data_as(fp), k.ctypes.data_as(fp), ck.ctypes.data_as(fp))
    return a, k, ck


def libm_f32(kind: str, x):
    """glibc sinf / cosf / logf element by element (kind 'sin', 'cos', 'log'); 'sin_cr', 'cos_cr': the correctly rounded floats of the oracle's Exp_SEK3."""
    a, ap = _f(x); out = np.zeros_like(a)
    lib().orc_libm_f32({"sin": 0, "cos": 1, "log": 2, "sin_cr": 3, "cos_cr": 4, "log_cr": 5}[kind], a.size, ap, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def cubic_step(c3, c2, c1, c0, min_step=0.2):
    return float(lib().orc_cubic_step(c3, c2, c1, c0, min_step))


def exp_sek3(omega, v, dt):
    o, op = _f(omega); vv, vp = _f(v); dR = np.zeros(9, np.float32); dT = np.zeros(3, np.float32)
    fp = C.POINTER(C.c_float)
    lib().orc_exp_sek3(op, vp, float(dt), dR.ctypes.data_as(fp), dT.ctypes.data_as(fp))
    return dR.reshape(3, 3), dT


def dist_se3(dR, dT):
    r, rp = _f(np.asarray(dR).reshape(9)); t, tp = _f(dT)
    return float(lib().orc_dist_se3(rp, tp))


def cubic_step_f32eig(c3, c2, c1, c0, min_step=0.2):
    return float(lib().orc_cubic_step_f32eig(c3, c2, c1, c0, min_step))


def dist_se3_f32logm(dR, dT):
    r, rp = _f(np.asarray(dR).reshape(9)); t, tp = _f(dT)
    return float(lib().orc_dist_se3_f32logm(rp, tp))


def test_eigenvalues(A, use_f32=False):
    A = np.ascontiguousarray(A, np.float64); n = A.shape[0]
    re = np.zeros(n); im = np.zeros(n); dp = C.POINTER(C.c_double)
    rc = lib().orc_test_eigenvalues(n, A.ctypes.data_as(dp), re.ctypes.data_as(dp), im.ctypes.data_as(dp), int(use_f32))
    return rc, re + 1j * im


def test_logm(A, use_f32=False):
    A = np.ascontiguousarray(A, np.float64); n = A.shape[0]
    out = np.zeros((n, n)); dp = C.POINTER(C.c_double)
    rc = lib().orc_test_logm(n, A.ctypes.data_as(dp), out.ctypes.data_as(dp), int(use_f32))
    return rc, out


def hessian_regularize(H, inliers):
    h, hp = _f(np.asarray(H).reshape(36)); out = np.zeros(36)
    lib().orc_hessian_regularize(hp, int(inliers), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(6, 6)


def radius_search(cloud_xyz, query, r2, use_kdtree, cap=4096):
    c, cp = _f(cloud_xyz); q, qp = _f(query)
    idx = np.zeros(cap, np.int32); d2 = np.zeros(cap, np.float32)
    n = lib().orc_radius_search(cp, c.shape[0], qp, float(r2), idx.ctypes.data_as(C.POINTER(C.c_int)),
                                d2.ctypes.data_as(C.POINTER(C.c_float)), cap, int(use_kdtree))
    return idx[:n], d2[:n]


# ----------------------------------------------------------------------------- point-cloud generator (oracle/pcd_oracle.cpp)
class Camera(C.Structure):
    _fields_ = [("scaling_factor", C.c_float), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


def glibc_rand_bytes(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint8)
    lib().orc_glibc_rand_bytes(C.c_uint(seed), out.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_long(n))
    return out


def pcd_generate(bgr8, depth16, camera, num_want: int = 3000, cap: int = 20000, debug: bool = False):
    """pcd_generator::create_pointcloud(1, ...) of the reference on one frame (cvo.cpp:355-366).
    camera = (scaling_factor, fx, fy, cx, cy).  Returns dict(xyz (n,3), feat (5,n), px (n,2)[, gray, map, ths, info])."""
    bgr = np.ascontiguousarray(bgr8, np.uint8); dep = np.ascontiguousarray(depth16, np.uint16)
    h, w = dep.shape
    assert bgr.shape == (h, w, 3)
    cam = Camera(*[float(v) for v in camera])
    xyz = np.zeros((cap, 3), np.float32); feat = np.zeros((5, cap), np.float32); px = np.zeros((cap, 2), np.uint16)
    gray = np.zeros((h, w), np.uint8); mp = np.zeros((h, w), np.float32); ths = np.zeros((h // 32, w // 32), np.float32); info = np.zeros(4, np.int32)
    L = lib()
    L.orc_pcd_generate.restype = C.c_int
    n = L.orc_pcd_generate(bgr.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h), C.byref(cam), C.c_int(num_want),
                           xyz.ctypes.data_as(C.c_void_p), feat.ctypes.data_as(C.c_void_p), px.ctypes.data_as(C.c_void_p), C.c_int(cap),
                           gray.ctypes.data_as(C.c_void_p), mp.ctypes.data_as(C.c_void_p), ths.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise ValueError(f"cap {cap} too small: {-n} points")
    out = dict(xyz=xyz[:n].copy(), feat=feat[:, :n].copy(), px=px[:n].copy(), n=n)
    if debug:
        out.update(gray=gray, map=mp, ths=ths, info=info)
    return out
