"""ctypes binding of libcvo_hip.so, mirroring the reference's `cvo::cvo` interface
(thirdparty/cvo/include/cvo.hpp:216-276): same method names, argument meaning and
error behaviour, with the pcd_generator output (positions + features) handed in
where the reference takes the RGB / depth images.

Every call goes through the C ABI of include/cvo_hip.h.  There is no fallback: a
missing library raises at load time, a missing gfx950 device raises CvoError on the
first call.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CVO_OK, CVO_ERR_NOT_INITIALIZED, CVO_ERR_EMPTY_CLOUD, CVO_ERR_HIP, CVO_ERR_INVALID, CVO_ERR_NO_DEVICE, CVO_ERR_TIMEOUT, CVO_ERR_PADDING, CVO_ERR_RANK_FAILED = range(9)
SLOT_FIXED, SLOT_MOVING, SLOT_PREVIOUS = 0, 1, 2
RESULT_FLOATS = 16


class CvoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcvo_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("ell", C.c_float), ("sigma", C.c_float), ("sp_thres", C.c_float), ("c", C.c_float), ("d", C.c_float),
                ("c_ell", C.c_float), ("c_sigma", C.c_float), ("max_iter", C.c_int), ("min_step", C.c_float),
                ("eps", C.c_float), ("eps_2", C.c_float)]


class InnP(C.Structure):
    _fields_ = [("value", C.c_float), ("num", C.c_int), ("num_e", C.c_int)]


class TraceRow(C.Structure):
    _fields_ = [("omega", C.c_float * 3), ("v", C.c_float * 3), ("nnz", C.c_int), ("candidates", C.c_int),
                ("B", C.c_double), ("C", C.c_double), ("D", C.c_double), ("E", C.c_double),
                ("step", C.c_float), ("ell", C.c_float), ("dist", C.c_float), ("pad_", C.c_int)]


class PairResult(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("R", C.c_float * 9), ("T", C.c_float * 3), ("ell", C.c_float),
                ("iter", C.c_int), ("A_nonzero", C.c_int), ("iterations_run", C.c_int), ("status", C.c_int),
                ("rebuilds", C.c_int), ("dense_fallbacks", C.c_int)]


class Camera(C.Structure):
    """cvo::camera_info (data_type.h:33-39)."""
    _fields_ = [("scaling_factor", C.c_float), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class LcScores(C.Structure):
    _fields_ = [("inn_prior", InnP), ("inn_lc_prior", InnP), ("inn_pre", InnP), ("inn_post", InnP), ("inn_fixed_pcd", InnP),
                ("inn_moving_pcd", InnP), ("post_hessian", C.c_double * 36), ("inliers_svd", C.c_int), ("inliers_pnpransac", C.c_int),
                ("cos_angle", C.c_float), ("accept", C.c_int)]


class AdaptiveParams(C.Structure):      # cvo_adaptive_params (adaptive_cvo.cpp:27-46)
    _fields_ = [("ell_init", C.c_float), ("ell_min", C.c_float), ("ell_max", C.c_float), ("dl_step", C.c_float), ("sigma", C.c_float),
                ("sp_thres", C.c_float), ("c", C.c_float), ("d", C.c_float), ("c_ell", C.c_float), ("c_sigma", C.c_float),
                ("max_iter", C.c_int), ("min_step", C.c_float), ("eps", C.c_float), ("eps_2", C.c_float)]


class AdaptiveRow(C.Structure):
    _fields_ = [("omega", C.c_float * 3), ("v", C.c_float * 3), ("dl", C.c_float), ("ell", C.c_float), ("step", C.c_float),
                ("nnz_xy", C.c_int), ("nnz_xx", C.c_int), ("nnz_yy", C.c_int)]


class TrackScores(C.Structure):
    _fields_ = [("inn_pre", InnP), ("inn_post", InnP), ("inn_fixed_pcd", InnP), ("inn_moving_pcd", InnP), ("post_hessian", C.c_double * 36),
                ("inliers", C.c_int), ("cos_angle", C.c_float)]


# every symbol include/cvo_hip.h declares (tests check the .so exports exactly these)
ABI_SYMBOLS = [
    "cvo_last_error", "cvo_device_count", "cvo_default_params", "cvo_create", "cvo_destroy", "cvo_set_pcd", "cvo_align",
    "cvo_align_traced", "cvo_match_odometry", "cvo_match_keyframe", "cvo_function_inner_product", "cvo_se3_hessian",
    "cvo_compute_innerproduct", "cvo_compute_innerproduct_lc", "cvo_update_fixed_pcd", "cvo_update_previous_pcd",
    "cvo_reset_keyframe", "cvo_reset_transform", "cvo_reset_initial", "cvo_get_fixed_and_moving_number",
    "cvo_get_iteration_number", "cvo_get_A_nonzero", "cvo_get_transform", "cvo_get_prev_accum_transform", "cvo_get_init",
    "cvo_get_first_frame", "cvo_set_first_frame", "cvo_get_state", "cvo_set_state", "cvo_set_workgroups",
    "cvo_batch_create", "cvo_batch_destroy", "cvo_batch_set_pair", "cvo_batch_set_state", "cvo_batch_set_workgroups",
    "cvo_batch_reset_states", "cvo_batch_align_async", "cvo_batch_wait", "cvo_batch_last_launch",
    "cvo_batch_results_to_device", "cvo_batch_last_phase_seconds", "cvo_batch_compute_innerproduct_lc",
    "cvo_set_pcd_images", "cvo_shared_cloud_count", "cvo_stage_next_frame", "cvo_staged_frame_count", "cvo_queued_score_count", "cvo_set_num_want", "cvo_match_odometry_images", "cvo_match_keyframe_images", "cvo_get_cloud", "cvo_get_selected_points",
    "cvo_batch_enqueue_innerproduct", "cvo_batch_innerproduct_results", "cvo_batch_compute_innerproduct",
    "cvo_selftest_cubic_step", "cvo_selftest_exp_sek3", "cvo_selftest_dist_se3", "cvo_selftest_libm", "cvo_selftest_pair_values",
    "cvo_function_inner_product_clouds", "cvo_se3_hessian_clouds", "cvo_batch_set_max_workgroups", "cvo_batch_set_adoption", "cvo_batch_last_adoptions", "cvo_batch_last_adoption_retractions",
    "cvo_adaptive_default_params", "cvo_adaptive_align",
    "cvo_shard_range", "cvo_comm_unique_id", "cvo_comm_create", "cvo_comm_create_all", "cvo_host_register", "cvo_host_unregister", "cvo_comm_destroy", "cvo_comm_info", "cvo_comm_set_gather_stream", "cvo_comm_library_path", "cvo_batch_gather_results",
    "cvo_gather_results", "cvo_multi_create", "cvo_multi_destroy", "cvo_multi_batch", "cvo_multi_align_async", "cvo_multi_wait",
    "cvo_batch_set_pairs", "cvo_batch_result_records", "cvo_shard_block", "cvo_batch_gather_results_padded", "cvo_batch_padded_records",
    "cvo_compact_records", "cvo_gather_results_padded", "cvo_multi_align_async_v", "cvo_batch_done", "cvo_batch_set_tail_scores", "cvo_batch_last_tail_answers", "cvo_batch_last_pair_seconds", "cvo_batch_last_pair_spans", "cvo_batch_last_tail_seconds", "cvo_set_tail_scores", "cvo_batch_last_cull_masks", "cvo_batch_last_nonzeros",
]

_lib = None


def lib_path() -> str:
    # CVO_HIP_LIB: experiment builds of the same ABI (scripts/gpu_*.sh); products leave it unset
    return os.environ.get("CVO_HIP_LIB") or os.path.join(HERE, "libcvo_hip.so")


def load_library():
    """Load libcvo_hip.so (built in-tree by cvo_slam_amd/build.py).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: build it with `python -m cvo_slam_amd.build` (hipcc, gfx950). "
                                "There is no CPU fallback.")
    L = C.CDLL(path)
    if os.environ.get("CVO_HIP_LIB"):
        # an experiment build or an OLDER build of the ABI (regression checks against last round's library): entry points it lacks are bound to
        # a stand-in that fails when called.  The product library (no CVO_HIP_LIB) must export everything: tests/test_capi_symbols.py.
        class _Missing:
            def __init__(self, name): self.name = name; self.argtypes = None; self.restype = None
            def __call__(self, *a): raise CvoError(4, f"{self.name} is not exported by {path}")
        class _Tolerant(C.CDLL):
            def __getattr__(self, name):
                try:
                    return super().__getattr__(name)
                except AttributeError:
                    if not name.startswith("cvo_"): raise
                    m = _Missing(name); self.__dict__[name] = m; return m
        L = _Tolerant(path)
    fp = C.POINTER(C.c_float); dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int); vp = C.c_void_p
    L.cvo_last_error.restype = C.c_char_p
    L.cvo_default_params.argtypes = [C.POINTER(Params)]
    L.cvo_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(vp)]
    L.cvo_destroy.argtypes = [vp]
    L.cvo_set_pcd.argtypes = [vp, fp, fp, C.c_int]
    L.cvo_align.argtypes = [vp]
    L.cvo_align_traced.argtypes = [vp, C.POINTER(TraceRow), C.c_int, ip]
    L.cvo_match_odometry.argtypes = [vp, fp, fp, C.c_int, dp]
    L.cvo_match_keyframe.argtypes = [vp, fp, fp, C.c_int, dp]
    L.cvo_function_inner_product.argtypes = [vp, C.c_int, fp, C.c_int, C.POINTER(InnP)]
    L.cvo_se3_hessian.argtypes = [vp, C.c_int, fp, C.c_int, dp, ip]
    L.cvo_compute_innerproduct.argtypes = [vp, C.POINTER(InnP), C.POINTER(InnP), dp, fp, ip, C.POINTER(InnP), C.POINTER(InnP), fp]
    L.cvo_compute_innerproduct_lc.argtypes = [vp] + [C.POINTER(InnP)] * 4 + [dp, fp, fp, fp, fp, ip, ip, C.POINTER(InnP), C.POINTER(InnP), fp]
    L.cvo_update_fixed_pcd.argtypes = [vp]
    L.cvo_update_previous_pcd.argtypes = [vp]
    L.cvo_reset_keyframe.argtypes = [vp, fp]
    L.cvo_reset_transform.argtypes = [vp, fp]
    L.cvo_reset_initial.argtypes = [vp, fp, fp]
    L.cvo_get_fixed_and_moving_number.argtypes = [vp, ip, ip]
    L.cvo_get_iteration_number.argtypes = [vp, ip]
    L.cvo_get_A_nonzero.argtypes = [vp, ip]
    L.cvo_get_transform.argtypes = [vp, fp]
    L.cvo_get_prev_accum_transform.argtypes = [vp, fp, fp]
    L.cvo_get_init.argtypes = [vp, ip]
    L.cvo_get_first_frame.argtypes = [vp, ip]
    L.cvo_set_first_frame.argtypes = [vp, C.c_int]
    L.cvo_get_state.argtypes = [vp, fp, fp, fp]
    L.cvo_set_state.argtypes = [vp, fp, fp, C.c_float]
    L.cvo_set_workgroups.argtypes = [vp, C.c_int]
    L.cvo_batch_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.POINTER(vp)]
    L.cvo_batch_destroy.argtypes = [vp]
    L.cvo_batch_set_pair.argtypes = [vp, C.c_int, fp, fp, C.c_int, fp, fp, C.c_int]
    L.cvo_batch_set_state.argtypes = [vp, C.c_int, fp, fp, C.c_float]
    L.cvo_batch_set_workgroups.argtypes = [vp, C.c_int]
    L.cvo_batch_reset_states.argtypes = [vp]
    L.cvo_batch_align_async.argtypes = [vp, C.c_int, vp]
    L.cvo_batch_wait.argtypes = [vp, C.POINTER(PairResult), C.c_int]
    L.cvo_batch_last_launch.argtypes = [vp, fp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.cvo_batch_results_to_device.argtypes = [vp, vp, C.c_int, vp]
    L.cvo_batch_last_phase_seconds.argtypes = [vp, dp]
    L.cvo_batch_compute_innerproduct_lc.argtypes = [vp, C.c_int, fp, fp, fp, C.POINTER(LcScores)]
    L.cvo_set_pcd_images.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(Camera)]
    L.cvo_set_num_want.argtypes = [vp, C.c_int]
    L.cvo_shared_cloud_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.cvo_stage_next_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(Camera)]
    L.cvo_staged_frame_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.cvo_queued_score_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.cvo_match_odometry_images.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(Camera), dp]
    L.cvo_match_keyframe_images.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(Camera), dp]
    L.cvo_get_cloud.argtypes = [vp, C.c_int, fp, fp, C.c_int, ip]
    L.cvo_get_selected_points.argtypes = [vp, C.c_int, vp, C.c_int, ip]
    L.cvo_batch_enqueue_innerproduct.argtypes = [vp, C.c_int]
    L.cvo_batch_innerproduct_results.argtypes = [vp, C.c_int, C.POINTER(TrackScores)]
    L.cvo_batch_compute_innerproduct.argtypes = [vp, C.c_int, C.POINTER(TrackScores)]
    for name in ("cvo_selftest_cubic_step", "cvo_selftest_exp_sek3", "cvo_selftest_dist_se3", "cvo_selftest_libm"):
        getattr(L, name).argtypes = [C.c_int, C.c_int, fp, fp]
    L.cvo_selftest_pair_values.argtypes = [C.c_int, C.POINTER(Params), C.c_float, C.c_int, fp, fp, fp]
    L.cvo_function_inner_product_clouds.argtypes = [vp, fp, fp, C.c_int, fp, fp, C.c_int, C.POINTER(InnP)]
    L.cvo_se3_hessian_clouds.argtypes = [vp, fp, fp, C.c_int, fp, fp, C.c_int, dp, ip]
    L.cvo_batch_set_max_workgroups.argtypes = [vp, C.c_int]
    L.cvo_batch_set_adoption.argtypes = [vp, C.c_int]
    L.cvo_batch_last_adoptions.argtypes = [vp, C.POINTER(C.c_int)]
    L.cvo_batch_last_adoption_retractions.argtypes = [vp, C.POINTER(C.c_int)]
    L.cvo_adaptive_default_params.argtypes = [C.POINTER(AdaptiveParams)]
    L.cvo_adaptive_align.argtypes = [C.c_int, C.POINTER(AdaptiveParams), fp, fp, C.c_int, fp, fp, C.c_int, fp, fp, fp, fp, ip, C.POINTER(AdaptiveRow), C.c_int, ip]
    L.cvo_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
    L.cvo_comm_unique_id.argtypes = [C.c_char_p]
    L.cvo_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.cvo_comm_create_all.argtypes = [ip, C.c_int, C.POINTER(vp)]
    L.cvo_comm_destroy.argtypes = [vp]
    L.cvo_comm_info.argtypes = [vp, ip, ip]
    L.cvo_host_register.argtypes = [vp, C.c_size_t]
    L.cvo_host_unregister.argtypes = [vp]
    L.cvo_comm_set_gather_stream.argtypes = [vp, C.c_int]
    L.cvo_comm_library_path.argtypes = [C.c_char_p, C.c_int]
    L.cvo_batch_gather_results.argtypes = [vp, vp, C.c_int, vp]
    L.cvo_gather_results.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp)]
    L.cvo_multi_create.argtypes = [C.POINTER(Params), ip, C.c_int, C.c_int, C.POINTER(vp)]
    L.cvo_multi_destroy.argtypes = [vp]
    L.cvo_multi_batch.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.cvo_multi_align_async.argtypes = [vp, C.c_int]
    L.cvo_multi_wait.argtypes = [vp, C.c_int, fp]
    pp = C.POINTER(fp)
    L.cvo_batch_set_pairs.argtypes = [vp, C.c_int, C.c_int, pp, pp, ip, pp, pp, ip]
    L.cvo_batch_result_records.argtypes = [vp, C.POINTER(vp)]
    L.cvo_shard_block.argtypes = [C.c_int, C.c_int]
    L.cvo_batch_gather_results_padded.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.cvo_batch_padded_records.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.cvo_compact_records.argtypes = [fp, C.c_int, C.c_int, fp, ip]
    L.cvo_gather_results_padded.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, ip, C.c_int, ip, C.POINTER(vp)]
    L.cvo_multi_align_async_v.argtypes = [vp, ip]
    L.cvo_batch_done.argtypes = [vp, ip]
    L.cvo_batch_set_tail_scores.argtypes = [vp, C.c_int]
    L.cvo_batch_last_tail_answers.argtypes = [vp, C.c_int, ip]
    L.cvo_batch_last_pair_seconds.argtypes = [vp, C.c_int, dp]
    L.cvo_set_tail_scores.argtypes = [vp, C.c_int]
    L.cvo_batch_last_tail_seconds.argtypes = [vp, dp]
    L.cvo_batch_last_pair_spans.argtypes = [vp, C.c_int, dp, dp, C.POINTER(C.c_int)]
    L.cvo_batch_last_cull_masks.argtypes = [vp, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.cvo_batch_last_nonzeros.argtypes = [vp, C.POINTER(C.c_longlong)]
    _lib = L
    return L


def adaptive_default_params() -> AdaptiveParams:
    p = AdaptiveParams(); _check(load_library().cvo_adaptive_default_params(C.byref(p))); return p


def adaptive_align(fixed_xyz, fixed_feat, moving_xyz, moving_feat, params: AdaptiveParams | None = None, R=None, T=None, trace_cap: int = 0, device: int = 0):
    """acvo::align (adaptive_cvo.cpp:490-555) on the GPU from a fresh object: dict(transform, R, T, ell, iter, trace)."""
    L = load_library()
    p = params or adaptive_default_params()
    fx, fxp, ff, ffp = _cloud_args(fixed_xyz, fixed_feat); mx, mxp, mf, mfp = _cloud_args(moving_xyz, moving_feat)
    Rb = np.ascontiguousarray(np.eye(3) if R is None else R, np.float32).reshape(9).copy(); Tb = np.ascontiguousarray(np.zeros(3) if T is None else T, np.float32).copy()
    tf = np.zeros(12, np.float32); ell = C.c_float(0); it = C.c_int(-1); n = C.c_int(0)
    rows = (AdaptiveRow * max(1, trace_cap))()
    fp = C.POINTER(C.c_float)
    _check(L.cvo_adaptive_align(device, C.byref(p), fxp, ffp, fx.shape[0], mxp, mfp, mx.shape[0], Rb.ctypes.data_as(fp), Tb.ctypes.data_as(fp), C.byref(ell),
                                tf.ctypes.data_as(fp), C.byref(it), rows if trace_cap else None, trace_cap, C.byref(n)))
    tr = [dict(omega=np.array(r.omega[:], np.float32), v=np.array(r.v[:], np.float32), dl=r.dl, ell=r.ell, step=r.step,
               nnz_xy=r.nnz_xy, nnz_xx=r.nnz_xx, nnz_yy=r.nnz_yy) for r in rows[: n.value]]
    return dict(transform=tf.reshape(3, 4), R=Rb.reshape(3, 3), T=Tb, ell=ell.value, iter=it.value, trace=tr)


def selftest_cubic_step(coef_minstep, device: int = 0):
    """cubic_step on the device for n x {c3, c2, c1, c0, min_step} (cvo.cpp:76-92,317-333)."""
    a = np.ascontiguousarray(coef_minstep, np.float32).reshape(-1, 5); out = np.zeros(a.shape[0], np.float32)
    fp = C.POINTER(C.c_float)
    _check(load_library().cvo_selftest_cubic_step(device, a.shape[0], a.ctypes.data_as(fp), out.ctypes.data_as(fp)))
    return out


def selftest_exp_sek3(omega_v_dt, device: int = 0):
    """Exp_SEK3 on the device for n x {omega, v, dt} (LieGroup.cpp:159-186): (n,3,3) dR and (n,3) dT."""
    a = np.ascontiguousarray(omega_v_dt, np.float32).reshape(-1, 7); out = np.zeros((a.shape[0], 12), np.float32)
    fp = C.POINTER(C.c_float)
    _check(load_library().cvo_selftest_exp_sek3(device, a.shape[0], a.ctypes.data_as(fp), out.ctypes.data_as(fp)))
    return out[:, :9].reshape(-1, 3, 3), out[:, 9:]


def selftest_dist_se3(dR_dT, device: int = 0):
    """dist_se3 on the device for n x {dR row-major, dT} (cvo.cpp:94-104)."""
    a = np.ascontiguousarray(dR_dT, np.float32).reshape(-1, 12); out = np.zeros(a.shape[0], np.float32)
    fp = C.POINTER(C.c_float)
    _check(load_library().cvo_selftest_dist_se3(device, a.shape[0], a.ctypes.data_as(fp), out.ctypes.data_as(fp)))
    return out


def selftest_libm(x, device: int = 0):
    """The device's float routines element by element: (n, 6) = OCML sinf, cosf, logf, then the correctly rounded sine, cosine (exp_sek3) and logarithm (gates)."""
    a = np.ascontiguousarray(x, np.float32).reshape(-1); out = np.zeros((a.shape[0], 6), np.float32)
    fp = C.POINTER(C.c_float)
    _check(load_library().cvo_selftest_libm(device, a.shape[0], a.ctypes.data_as(fp), out.ctypes.data_as(fp)))
    return out


def selftest_pair_values(y_g, ell: float, params: "Params | None" = None, device: int = 0):
    """The pair arithmetic of se_kernel (cvo.cpp:166-175) on the device by the align kernel's four routes, for n x {y[3], g[5]} against a fixed
    point at the origin with zero features: (a (n, 4), d2 (n,), d2c (n,))."""
    a = np.ascontiguousarray(y_g, np.float32).reshape(-1, 8); out = np.zeros((a.shape[0], 4), np.float32); aux = np.zeros((a.shape[0], 2), np.float32)
    fp = C.POINTER(C.c_float)
    _check(load_library().cvo_selftest_pair_values(device, C.byref(params) if params is not None else None, float(ell), a.shape[0], a.ctypes.data_as(fp),
                                                   out.ctypes.data_as(fp), aux.ctypes.data_as(fp)))
    return out, aux[:, 0], aux[:, 1]


def _check(rc: int):
    if rc != CVO_OK:
        raise CvoError(rc, load_library().cvo_last_error().decode())


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _tran(t):
    if t is None:
        return None, None
    return _f(np.asarray(t, dtype=np.float32).reshape(12))


def default_params() -> Params:
    p = Params(); _check(load_library().cvo_default_params(C.byref(p))); return p


def device_count() -> int:
    return int(load_library().cvo_device_count())


def _cloud_args(xyz, feat):
    x, xp = _f(xyz); f, fpt = _f(feat)
    if x.ndim != 2 or x.shape[1] != 3 or f.shape != (5, x.shape[0]):
        raise ValueError("cloud must be xyz (n,3) and feat (5,n)")
    return x, xp, f, fpt


class Cvo:
    """One `cvo::cvo` object (cvo.hpp:82-282) living on a gfx950 device."""

    def __init__(self, params: Params | None = None, device: int = 0):
        self.L = load_library()
        self.params = params or default_params()
        self.h = C.c_void_p()
        _check(self.L.cvo_create(C.byref(self.params), device, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.cvo_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- cvo.cpp:345-386 (cloud handed in instead of RGB/depth)
    def set_pcd(self, xyz, feat):
        x, xp, f, fpt = _cloud_args(xyz, feat)
        _check(self.L.cvo_set_pcd(self.h, xp, fpt, x.shape[0]))

    # -- cvo.cpp:345-386 with the images, as the reference's signature has it (pcd_generator on the GPU)
    @staticmethod
    def _images(bgr8, depth16):
        bgr = np.ascontiguousarray(bgr8, np.uint8); dep = np.ascontiguousarray(depth16, np.uint16)
        h, w = dep.shape
        if bgr.shape != (h, w, 3):
            raise ValueError("bgr8 must be (h, w, 3) for a (h, w) depth image")
        return bgr, dep, w, h

    def set_pcd_images(self, bgr8, depth16, camera):
        """camera = (scaling_factor, fx, fy, cx, cy)."""
        bgr, dep, w, h = self._images(bgr8, depth16); cam = Camera(*[float(v) for v in camera])
        _check(self.L.cvo_set_pcd_images(self.h, bgr.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p), w, h, C.byref(cam)))

    def stage_next_frame(self, bgr8, depth16, camera):
        """cvo_stage_next_frame: start generating this frame's cloud now; a later set_pcd_images / match_*_images with the same images takes it.
        The arrays are kept alive (and must not be written) until then."""
        bgr, dep, w, h = self._images(bgr8, depth16); cam = Camera(*[float(v) for v in camera])
        self._staged = (bgr, dep)                                    # the library reads them from its worker thread
        _check(self.L.cvo_stage_next_frame(self.h, bgr.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p), w, h, C.byref(cam)))

    def staged_frame_count(self) -> int:
        """cvo_staged_frame_count: clouds this object took from a generation started ahead of time"""
        n = C.c_int(0); _check(self.L.cvo_staged_frame_count(self.h, C.byref(n))); return n.value

    def set_num_want(self, num_want: int):
        _check(self.L.cvo_set_num_want(self.h, int(num_want)))

    def match_keyframe_images(self, bgr8, depth16, camera):
        bgr, dep, w, h = self._images(bgr8, depth16); cam = Camera(*[float(v) for v in camera]); out = np.zeros(12, np.float64)
        _check(self.L.cvo_match_keyframe_images(self.h, bgr.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p), w, h, C.byref(cam),
                                                out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.reshape(3, 4)

    def match_odometry_images(self, bgr8, depth16, camera):
        bgr, dep, w, h = self._images(bgr8, depth16); cam = Camera(*[float(v) for v in camera]); out = np.zeros(12, np.float64)
        _check(self.L.cvo_match_odometry_images(self.h, bgr.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p), w, h, C.byref(cam),
                                                out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.reshape(3, 4)

    def get_cloud(self, slot: int):
        n = C.c_int(0)
        _check(self.L.cvo_get_cloud(self.h, slot, None, None, 0, C.byref(n)))
        xyz = np.zeros((n.value, 3), np.float32); feat = np.zeros((5, n.value), np.float32)
        if n.value:
            _check(self.L.cvo_get_cloud(self.h, slot, xyz.ctypes.data_as(C.POINTER(C.c_float)), feat.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)))
        return xyz, feat

    def get_selected_points(self, slot: int):
        n = C.c_int(0)
        _check(self.L.cvo_get_selected_points(self.h, slot, None, 0, C.byref(n)))
        px = np.zeros((n.value, 2), np.uint16)
        if n.value:
            _check(self.L.cvo_get_selected_points(self.h, slot, px.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return px

    # -- cvo.cpp:763-821
    def align(self, trace_cap: int = 0):
        if trace_cap:
            rows = (TraceRow * trace_cap)(); n = C.c_int(0)
            _check(self.L.cvo_align_traced(self.h, rows, trace_cap, C.byref(n)))
            return [dict(omega=np.array(r.omega[:], np.float32), v=np.array(r.v[:], np.float32), nnz=r.nnz,
                         candidates=r.candidates, BCDE=np.array([r.B, r.C, r.D, r.E]), step=r.step, ell=r.ell, dist=r.dist)
                    for r in rows[: min(n.value, trace_cap)]]
        _check(self.L.cvo_align(self.h))
        return None

    # -- cvo.cpp:461-473 / 563-576.  "cvo not initialized !" -> CvoError(code 1), output untouched
    def match_odometry(self, xyz, feat):
        x, xp, f, fpt = _cloud_args(xyz, feat); out = np.zeros(12, np.float64)
        _check(self.L.cvo_match_odometry(self.h, xp, fpt, x.shape[0], out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.reshape(3, 4)

    def match_keyframe(self, xyz, feat):
        x, xp, f, fpt = _cloud_args(xyz, feat); out = np.zeros(12, np.float64)
        _check(self.L.cvo_match_keyframe(self.h, xp, fpt, x.shape[0], out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.reshape(3, 4)

    # -- cvo.cpp:388-459
    def function_inner_product(self, slot_a, tran_a, slot_b):
        r = InnP(); t, tp = _tran(tran_a)
        _check(self.L.cvo_function_inner_product(self.h, slot_a, tp, slot_b, C.byref(r)))
        return (r.value, r.num, r.num_e)

    # -- cvo.cpp:620-759
    def se3_hessian(self, slot_a, tran_a, slot_b, inliers: int = 0):
        H = np.zeros(36); inl = C.c_int(inliers); t, tp = _tran(tran_a)
        _check(self.L.cvo_se3_hessian(self.h, slot_a, tp, slot_b, H.ctypes.data_as(C.POINTER(C.c_double)), C.byref(inl)))
        return H.reshape(6, 6), inl.value

    # -- the same two on clouds handed in directly, as the reference's members take them (cvo.hpp:222, 260)
    def function_inner_product_clouds(self, xyz_a, feat_a, xyz_b, feat_b):
        a, ap, fa, fap = _cloud_args(xyz_a, feat_a); b, bp, fb, fbp = _cloud_args(xyz_b, feat_b); r = InnP()
        _check(self.L.cvo_function_inner_product_clouds(self.h, ap, fap, a.shape[0], bp, fbp, b.shape[0], C.byref(r)))
        return (r.value, r.num, r.num_e)

    def se3_hessian_clouds(self, xyz_a, feat_a, xyz_b, feat_b, inliers: int = 0):
        a, ap, fa, fap = _cloud_args(xyz_a, feat_a); b, bp, fb, fbp = _cloud_args(xyz_b, feat_b)
        H = np.zeros(36); inl = C.c_int(inliers)
        _check(self.L.cvo_se3_hessian_clouds(self.h, ap, fap, a.shape[0], bp, fbp, b.shape[0], H.ctypes.data_as(C.POINTER(C.c_double)), C.byref(inl)))
        return H.reshape(6, 6), inl.value

    # -- cvo.cpp:475-503
    def compute_innerproduct(self, tran):
        pre, post, fx, mv = InnP(), InnP(), InnP(), InnP()
        H = np.zeros(36); inl = C.c_int(0); cos = C.c_float(0); t, tp = _tran(tran)
        _check(self.L.cvo_compute_innerproduct(self.h, C.byref(pre), C.byref(post), H.ctypes.data_as(C.POINTER(C.c_double)), tp,
                                               C.byref(inl), C.byref(fx), C.byref(mv), C.byref(cos)))
        tup = lambda r: (r.value, r.num, r.num_e)
        return dict(inn_pre=tup(pre), inn_post=tup(post), post_hessian=H.reshape(6, 6), inliers=inl.value,
                    inn_fixed_pcd=tup(fx), inn_moving_pcd=tup(mv), cos_angle=cos.value)

    # -- cvo.cpp:505-561
    def compute_innerproduct_lc(self, prior_tran, lc_prior_tran, lc_prior_tran_2, lc_tran):
        prior, lcp, lcpre, lcpost, fx, mv = (InnP() for _ in range(6))
        H = np.zeros(36); i1 = C.c_int(0); i2 = C.c_int(0); cos = C.c_float(0)
        keep = [_tran(t) for t in (prior_tran, lc_prior_tran, lc_prior_tran_2, lc_tran)]
        _check(self.L.cvo_compute_innerproduct_lc(self.h, C.byref(prior), C.byref(lcp), C.byref(lcpre), C.byref(lcpost),
                                                  H.ctypes.data_as(C.POINTER(C.c_double)), keep[0][1], keep[1][1], keep[2][1], keep[3][1],
                                                  C.byref(i1), C.byref(i2), C.byref(fx), C.byref(mv), C.byref(cos)))
        tup = lambda r: (r.value, r.num, r.num_e)
        return dict(inn_prior=tup(prior), inn_lc_prior=tup(lcp), inn_lc_pre=tup(lcpre), inn_lc_post=tup(lcpost),
                    post_hessian=H.reshape(6, 6), inliers_svd=i1.value, inliers_pnpransac=i2.value,
                    inn_fixed_pcd=tup(fx), inn_moving_pcd=tup(mv), cos_angle=cos.value)

    # -- cvo.cpp:578-618
    def update_fixed_pcd(self): _check(self.L.cvo_update_fixed_pcd(self.h))
    def update_previous_pcd(self): _check(self.L.cvo_update_previous_pcd(self.h))

    def reset_keyframe(self, odometry):
        t, tp = _tran(odometry); _check(self.L.cvo_reset_keyframe(self.h, tp))

    def reset_transform(self, odometry):
        t, tp = _tran(odometry); _check(self.L.cvo_reset_transform(self.h, tp))

    def reset_initial(self, odometry):
        t, tp = _tran(odometry); out = np.zeros(12, np.float32)
        _check(self.L.cvo_reset_initial(self.h, tp, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out.reshape(3, 4)

    # -- getters, cvo.hpp:268-270 + public members cvo.hpp:139-144
    def get_fixed_and_moving_number(self):
        a = C.c_int(0); b = C.c_int(0); _check(self.L.cvo_get_fixed_and_moving_number(self.h, C.byref(a), C.byref(b))); return a.value, b.value

    def get_iteration_number(self):
        a = C.c_int(0); _check(self.L.cvo_get_iteration_number(self.h, C.byref(a))); return a.value

    def get_A_nonzero(self):
        a = C.c_int(0); _check(self.L.cvo_get_A_nonzero(self.h, C.byref(a))); return a.value

    @property
    def transform(self):
        out = np.zeros(12, np.float32); _check(self.L.cvo_get_transform(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))); return out.reshape(3, 4)

    @property
    def init(self):
        a = C.c_int(0); _check(self.L.cvo_get_init(self.h, C.byref(a))); return bool(a.value)

    @property
    def first_frame(self):
        a = C.c_int(0); _check(self.L.cvo_get_first_frame(self.h, C.byref(a))); return bool(a.value)

    @first_frame.setter
    def first_frame(self, v):
        _check(self.L.cvo_set_first_frame(self.h, int(bool(v))))

    def prev_accum_transform(self):
        a = np.zeros(12, np.float32); b = np.zeros(12, np.float32); fp = C.POINTER(C.c_float)
        _check(self.L.cvo_get_prev_accum_transform(self.h, a.ctypes.data_as(fp), b.ctypes.data_as(fp)))
        return a.reshape(3, 4), b.reshape(3, 4)

    def get_state(self):
        R = np.zeros(9, np.float32); T = np.zeros(3, np.float32); ell = C.c_float(0); fp = C.POINTER(C.c_float)
        _check(self.L.cvo_get_state(self.h, R.ctypes.data_as(fp), T.ctypes.data_as(fp), C.byref(ell)))
        return dict(R=R.reshape(3, 3), T=T, ell=ell.value)

    def set_state(self, R, T, ell):
        r, rp = _f(np.asarray(R).reshape(9)); t, tp = _f(np.asarray(T).reshape(3))
        _check(self.L.cvo_set_state(self.h, rp, tp, float(ell)))

    def set_workgroups(self, g: int):
        _check(self.L.cvo_set_workgroups(self.h, int(g)))

    def set_tail_scores(self, on):
        """cvo_set_tail_scores: the alignment queues the tracker's score block behind itself and compute_innerproduct(the result) only collects.
        False / 0 never, True / 1 every alignment, 2 (the handle's default) when the previous alignment was followed by that question"""
        _check(self.L.cvo_set_tail_scores(self.h, int(on)))

    def queued_score_count(self) -> int:
        """cvo_queued_score_count: score blocks answered by what an alignment of this object had queued behind itself"""
        n = C.c_int(0); _check(self.L.cvo_queued_score_count(self.h, C.byref(n))); return n.value

    def shared_cloud_count(self) -> int:
        """cvo_shared_cloud_count: clouds this object took from the thread's previous generation (same images) instead of generating them"""
        n = C.c_int(0); _check(self.L.cvo_shared_cloud_count(self.h, C.byref(n))); return n.value


RESULT_FLOATS = 16      # CVO_RESULT_FLOATS
COMM_ID_BYTES = 128     # CVO_COMM_ID_BYTES


def shard_range(n_pairs: int, rank: int, world: int) -> range:
    """cvo_shard_range: the contiguous block of global pair indices `rank` owns."""
    first = C.c_int(0); count = C.c_int(0)
    _check(load_library().cvo_shard_range(n_pairs, rank, world, C.byref(first), C.byref(count)))
    return range(first.value, first.value + count.value)


def shard_block(n_pairs: int, world: int) -> int:
    """cvo_shard_block: records every rank contributes to a gather (its own, then padding)."""
    return int(load_library().cvo_shard_block(n_pairs, world))


def compact_records(gathered, n_pairs: int, world: int):
    """cvo_compact_records: rank-major gathered blocks -> (n_pairs, 16) records in global pair order, first non-zero status."""
    g = np.ascontiguousarray(gathered, np.float32)
    assert g.size == world * shard_block(n_pairs, world) * RESULT_FLOATS, g.shape
    out = np.zeros((n_pairs, RESULT_FLOATS), np.float32); err = C.c_int(0); fp = C.POINTER(C.c_float)
    _check(load_library().cvo_compact_records(g.ctypes.data_as(fp), n_pairs, world, out.ctypes.data_as(fp), C.byref(err)))
    return out, err.value


def host_register(array):
    """cvo_host_register: pin + map a numpy array's memory; clouds handed over from inside it are read in place (no staging copy)."""
    a = np.asarray(array)
    assert a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"]
    _check(load_library().cvo_host_register(C.c_void_p(a.ctypes.data), a.nbytes))


def host_unregister(array):
    _check(load_library().cvo_host_unregister(C.c_void_p(np.asarray(array).ctypes.data)))


def comm_library_path() -> str:
    """cvo_comm_library_path: the file the RCCL bound by the C ABI was loaded from."""
    buf = C.create_string_buffer(1024)
    _check(load_library().cvo_comm_library_path(buf, 1024))
    return buf.value.decode()


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().cvo_comm_unique_id(buf))
    return buf.raw


class CvoComm:
    """One rank's RCCL communicator (one process per GPU): rank 0 makes the id, the launcher broadcasts it."""

    def __init__(self, unique_id: bytes, n_ranks: int, rank: int, device: int = 0):
        self.L = load_library(); self.h = C.c_void_p(); self.n_ranks = n_ranks; self.rank = rank
        assert len(unique_id) == COMM_ID_BYTES
        _check(self.L.cvo_comm_create(unique_id, n_ranks, rank, device, C.byref(self.h)))

    def set_gather_stream(self, on: bool):
        """cvo_comm_set_gather_stream: every gather of this communicator on one stream of its own (fallback mode, include/cvo_hip.h)."""
        _check(self.L.cvo_comm_set_gather_stream(self.h, 1 if on else 0))

    def info(self):
        """(ranks, rank) as the RCCL communicator reports them (ncclCommCount / ncclCommUserRank)."""
        n = C.c_int(0); r = C.c_int(0)
        _check(self.L.cvo_comm_info(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.cvo_comm_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CvoMulti:
    """Single process, several GPUs: one batch per device, contiguous block sharding, one RCCL all-gather of the result records
    enqueued behind every device's align launch (cvo_multi_*)."""

    def __init__(self, devices, max_pairs_per_device: int, params: Params | None = None):
        self.L = load_library(); self.params = params or default_params()
        self.devices = list(devices); self.max_pairs = max_pairs_per_device
        dev = (C.c_int * len(self.devices))(*self.devices)
        self.h = C.c_void_p()
        _check(self.L.cvo_multi_create(C.byref(self.params), dev, len(self.devices), max_pairs_per_device, C.byref(self.h)))

    def batch(self, i: int) -> "CvoBatch":
        h = C.c_void_p(); _check(self.L.cvo_multi_batch(self.h, i, C.byref(h)))
        return CvoBatch._borrow(h, self.max_pairs, self.params)

    def align_async(self, n: int):
        self._n = n
        _check(self.L.cvo_multi_align_async(self.h, n))

    def align_async_v(self, n_pairs):
        """n_pairs[i] pairs on device i; every device contributes max(n_pairs) records (padding behind its own)."""
        arr = (C.c_int * len(self.devices))(*[int(v) for v in n_pairs]); self._n = max(int(v) for v in n_pairs)
        _check(self.L.cvo_multi_align_async_v(self.h, arr))

    def wait(self, from_device: int = 0):
        out = np.zeros((len(self.devices) * self._n, RESULT_FLOATS), np.float32)
        _check(self.L.cvo_multi_wait(self.h, from_device, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.cvo_multi_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CvoBatch:
    """Independent frame pairs aligned in one persistent launch (the
    keyframe<->keyframe batch of keyframe_graph.cpp:622-731; BASELINE configs 3-4)."""

    def __init__(self, max_pairs: int, params: Params | None = None, device: int = 0):
        self.L = load_library()
        self.params = params or default_params()
        self.max_pairs = max_pairs
        self.h = C.c_void_p()
        self.owned = True
        _check(self.L.cvo_batch_create(C.byref(self.params), device, max_pairs, C.byref(self.h)))

    @classmethod
    def _borrow(cls, handle, max_pairs, params):
        """A batch owned by a CvoMulti: same methods, never destroyed from here."""
        b = cls.__new__(cls)
        b.L = load_library(); b.params = params; b.max_pairs = max_pairs; b.h = handle; b.owned = False
        return b

    def gather_results(self, comm: "CvoComm", n: int, recv_device_ptr: int):
        """ONE ncclAllGather of the first n result records (the align kernel wrote them), enqueued behind the last launch on its
        stream.  n must be the same on every rank; see gather_results_padded."""
        _check(self.L.cvo_batch_gather_results(self.h, comm.h, n, C.c_void_p(recv_device_ptr)))

    def gather_results_padded(self, comm: "CvoComm", n_valid: int, n_block: int, recv_device_ptr: int, launch_status: int = 0):
        """Every rank sends n_block records: its n_valid own, then padding (status CVO_ERR_PADDING); launch_status != 0: this rank's
        launch failed, all its records carry that code -- the rank still enters the collective."""
        _check(self.L.cvo_batch_gather_results_padded(self.h, comm.h, n_valid, n_block, launch_status, C.c_void_p(recv_device_ptr)))

    def padded_records(self, n_valid: int, n_block: int, launch_status: int = 0) -> int:
        """Device address of the block this rank would send (for launchers that run the collective themselves)."""
        p = C.c_void_p()
        _check(self.L.cvo_batch_padded_records(self.h, n_valid, n_block, launch_status, C.byref(p)))
        return int(p.value)

    def result_records(self) -> int:
        """Device address of the record table the align kernel writes (n x 16 floats), valid once the launch's stream has drained."""
        p = C.c_void_p()
        _check(self.L.cvo_batch_result_records(self.h, C.byref(p)))
        return int(p.value)

    def close(self):
        if getattr(self, "h", None) and self.h.value and getattr(self, "owned", True):
            self.L.cvo_batch_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pair(self, p, fixed_xyz, fixed_feat, moving_xyz, moving_feat):
        fx, fxp, ff, ffp = _cloud_args(fixed_xyz, fixed_feat)
        mx, mxp, mf, mfp = _cloud_args(moving_xyz, moving_feat)
        _check(self.L.cvo_batch_set_pair(self.h, p, fxp, ffp, fx.shape[0], mxp, mfp, mx.shape[0]))

    @staticmethod
    def prepare_pairs(pairs):
        """pairs: sequence of (fixed_xyz, fixed_feat, moving_xyz, moving_feat).  Returns the pointer tables cvo_batch_set_pairs takes
        (and keeps the arrays alive), so that a loop handing the same host buffers over every step pays for the hand-over only."""
        fp = C.POINTER(C.c_float)
        keep = [tuple(_cloud_args(fx, ff)[0::2] + _cloud_args(mx, mf)[0::2]) for fx, ff, mx, mf in pairs]
        n = len(keep)
        tab = lambda k: (fp * n)(*[q[k].ctypes.data_as(fp) for q in keep])
        return dict(n=n, keep=keep, fx=tab(0), ff=tab(1), mx=tab(2), mf=tab(3),
                    nf=(C.c_int * n)(*[q[0].shape[0] for q in keep]), nm=(C.c_int * n)(*[q[2].shape[0] for q in keep]))

    def set_pairs(self, prepared, first: int = 0):
        """cvo_batch_set_pairs: one hand-over for all pairs of `prepared` (prepare_pairs), or of a sequence of cloud tuples."""
        pr = prepared if isinstance(prepared, dict) else self.prepare_pairs(prepared)
        _check(self.L.cvo_batch_set_pairs(self.h, first, pr["n"], pr["fx"], pr["ff"], pr["nf"], pr["mx"], pr["mf"], pr["nm"]))

    def set_state(self, p, R, T, ell):
        r, rp = _f(np.asarray(R).reshape(9)); t, tp = _f(np.asarray(T).reshape(3))
        _check(self.L.cvo_batch_set_state(self.h, p, rp, tp, float(ell)))

    def set_workgroups(self, g: int):
        _check(self.L.cvo_batch_set_workgroups(self.h, int(g)))

    def set_max_workgroups(self, n: int):
        _check(self.L.cvo_batch_set_max_workgroups(self.h, int(n)))

    def set_adoption(self, on: bool):
        """finished workgroups help with the pairs of their launch that still run (cvo_hip.h: cvo_batch_set_adoption)"""
        _check(self.L.cvo_batch_set_adoption(self.h, int(bool(on))))

    def set_tail_scores(self, on: bool):
        """the tracker's score block answered by the align launch itself (cvo_hip.h: cvo_batch_set_tail_scores)"""
        _check(self.L.cvo_batch_set_tail_scores(self.h, int(bool(on))))

    def last_pair_seconds(self, n: int):
        out = np.zeros(n); _check(self.L.cvo_batch_last_pair_seconds(self.h, n, out.ctypes.data_as(C.POINTER(C.c_double)))); return out

    def last_pair_spans(self, n: int):
        """(start, end) of every pair of the last launch in seconds of the device's 100 MHz clock, and the iteration a helper joined at (0 = none)"""
        t0 = np.zeros(n); t1 = np.zeros(n); j = (C.c_int * n)()
        _check(self.L.cvo_batch_last_pair_spans(self.h, n, t0.ctypes.data_as(C.POINTER(C.c_double)), t1.ctypes.data_as(C.POINTER(C.c_double)), j))
        return t0, t1, np.array(list(j))

    def last_tail_seconds(self):
        out = np.zeros(4); _check(self.L.cvo_batch_last_tail_seconds(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))); return out

    def last_cull_masks(self, n: int):
        m = (C.c_ulonglong * n)(); q = (C.c_ulonglong * n)(); _check(self.L.cvo_batch_last_cull_masks(self.h, n, m, q)); return [int(x) for x in m], [int(x) for x in q]

    def last_tail_answers(self, n: int):
        m = (C.c_int * n)()
        _check(self.L.cvo_batch_last_tail_answers(self.h, n, m))
        return list(m)

    def last_adoptions(self) -> int:
        n = C.c_int(0)
        _check(self.L.cvo_batch_last_adoptions(self.h, C.byref(n)))
        return int(n.value)

    def last_adoption_retractions(self) -> int:
        n = C.c_int(0)
        _check(self.L.cvo_batch_last_adoption_retractions(self.h, C.byref(n)))
        return int(n.value)

    def reset_states(self):
        _check(self.L.cvo_batch_reset_states(self.h))

    def align_async(self, n_pairs: int, stream: int | None = None):
        _check(self.L.cvo_batch_align_async(self.h, n_pairs, C.c_void_p(stream) if stream else None))

    def wait(self, n: int = 0):
        if n <= 0:
            _check(self.L.cvo_batch_wait(self.h, None, 0)); return []
        res = (PairResult * n)()
        _check(self.L.cvo_batch_wait(self.h, res, n))
        return [dict(transform=np.array(r.transform[:], np.float32).reshape(3, 4), R=np.array(r.R[:], np.float32).reshape(3, 3),
                     T=np.array(r.T[:], np.float32), ell=r.ell, iter=r.iter, A_nonzero=r.A_nonzero,
                     iterations_run=r.iterations_run, status=r.status, rebuilds=r.rebuilds, dense_fallbacks=r.dense_fallbacks) for r in res]

    def done(self) -> bool:
        """cvo_batch_done: has the last launch completed?  Never blocks."""
        d = C.c_int(0)
        _check(self.L.cvo_batch_done(self.h, C.byref(d)))
        return bool(d.value)

    def align(self, n_pairs: int):
        self.align_async(n_pairs)
        return self.wait(n_pairs)

    def last_launch(self):
        ms = C.c_float(0); it = C.c_longlong(0); ca = C.c_longlong(0)
        _check(self.L.cvo_batch_last_launch(self.h, C.byref(ms), C.byref(it), C.byref(ca)))
        nz = C.c_longlong(0)
        _check(self.L.cvo_batch_last_nonzeros(self.h, C.byref(nz)))
        return dict(kernel_ms=ms.value, iterations_total=it.value, candidates_total=ca.value, nonzeros_total=nz.value)

    def last_phase_seconds(self):
        out = np.zeros(10); _check(self.L.cvo_batch_last_phase_seconds(self.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return dict(zip(("lists", "candidates", "cand_reduce", "linesearch", "cand_exchange", "epilogue", "lists_cull", "cand_prologue", "lists_sort", "cand_rows"), out.tolist()))

    # -- keyframe_graph.cpp:704-717 for every aligned pair, one launch
    def compute_innerproduct_lc(self, prior_tran, lc_prior_tran, lc_prior_tran_2):
        """prior_tran / lc_prior_tran / lc_prior_tran_2: (n, 3, 4) Affine3f each.  Returns one dict per pair with the
        fields of `compute_innerproduct_lc` (cvo.cpp:505-561) plus `accept` (the reference's rule)."""
        pt = np.ascontiguousarray(prior_tran, np.float32).reshape(-1, 12); n = pt.shape[0]
        lp = np.ascontiguousarray(lc_prior_tran, np.float32).reshape(n, 12); l2 = np.ascontiguousarray(lc_prior_tran_2, np.float32).reshape(n, 12)
        out = (LcScores * n)()
        f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        _check(self.L.cvo_batch_compute_innerproduct_lc(self.h, n, f(pt), f(lp), f(l2), out))
        tup = lambda r: (r.value, r.num, r.num_e)
        return [dict(inn_prior=tup(o.inn_prior), inn_lc_prior=tup(o.inn_lc_prior), inn_lc_pre=tup(o.inn_pre), inn_lc_post=tup(o.inn_post),
                     inn_fixed_pcd=tup(o.inn_fixed_pcd), inn_moving_pcd=tup(o.inn_moving_pcd),
                     post_hessian=np.array(o.post_hessian[:]).reshape(6, 6), inliers_svd=o.inliers_svd, inliers_pnpransac=o.inliers_pnpransac,
                     cos_angle=o.cos_angle, accept=bool(o.accept)) for o in out]

    # -- local_tracker.cpp:240-251 for every aligned pair, one launch queued behind the align launch
    def enqueue_innerproduct(self, n: int):
        _check(self.L.cvo_batch_enqueue_innerproduct(self.h, n))

    def innerproduct_results_raw(self, n: int):
        """cvo_batch_innerproduct_results into a ctypes array of TrackScores: the C call alone, without the per-pair Python objects below."""
        out = (TrackScores * n)()
        _check(self.L.cvo_batch_innerproduct_results(self.h, n, out))
        return out

    def innerproduct_results(self, n: int):
        """One dict per pair with the fields of `compute_innerproduct` (cvo.cpp:475-503), tran = the pair's own align() result."""
        out = self.innerproduct_results_raw(n)
        tup = lambda r: (r.value, r.num, r.num_e)
        return [dict(inn_pre=tup(o.inn_pre), inn_post=tup(o.inn_post), inn_fixed_pcd=tup(o.inn_fixed_pcd), inn_moving_pcd=tup(o.inn_moving_pcd),
                     post_hessian=np.array(o.post_hessian[:]).reshape(6, 6), inliers=o.inliers, cos_angle=o.cos_angle) for o in out]

    def compute_innerproduct(self, n: int):
        self.enqueue_innerproduct(n)
        return self.innerproduct_results(n)

    def results_to_device(self, dst_device_ptr: int, n: int, stream: int | None = None):
        _check(self.L.cvo_batch_results_to_device(self.h, C.c_void_p(dst_device_ptr), n, C.c_void_p(stream) if stream else None))
