/* ============================================================================
 * include/cvo_hip.h -- C ABI of libcvo_hip.so: the MI355X (gfx950) implementation
 * of CVO-SLAM's per-frame-pair CVO alignment hot path.
 *
 * The reference has no FFI layer: the C++ class `cvo::cvo`
 * (thirdparty/cvo/include/cvo.hpp:82-282) IS the boundary that local_tracker /
 * keyframe_graph link against.  Each entry point below replaces one member of
 * that class (cited per function); a header-only `cvo::cvo` adaptor with the
 * reference's exact signatures forwards to them.  A point cloud
 * enters either as plain arrays (cvo_set_pcd: the reference's pcd_generator stays
 * on the host side of the adaptor) or as the RGB and depth images themselves
 * (cvo_set_pcd_images: the generator runs on the GPU, the cloud never leaves HBM).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are HOST pointers unless the
 *     name says `_device`.
 *   - point cloud = n x 3 f32 positions, AoS, 12-byte stride (cloud_t,
 *     data_type.h:30) + 5 channel-major f32 arrays of n (the column-major
 *     Eigen::Matrix<float,Dynamic,5> `features`, data_type.h:75).
 *   - rigid transforms = 3x4 row-major [R | t] (top rows of Eigen::Affine3f).
 *   - every call returns a status; the reference's functions are `void` and
 *     print-and-return on "not initialized" (cvo.cpp:463-466), which the adaptor
 *     reproduces from CVO_ERR_NOT_INITIALIZED.
 *   - a handle is used by one thread at a time; distinct handles are independent
 *     (own HIP stream, no globals), like distinct cvo::cvo objects.
 * ========================================================================== */
#ifndef CVO_HIP_H
#define CVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    CVO_OK = 0,
    CVO_ERR_NOT_INITIALIZED = 1,  /* cvo.cpp:463-466 / 565-568 */
    CVO_ERR_EMPTY_CLOUD = 2,      /* reference: KD adaptor assert / UB (Q8) */
    CVO_ERR_HIP = 3,              /* HIP runtime error, message via cvo_last_error() */
    CVO_ERR_INVALID = 4,          /* bad argument */
    CVO_ERR_NO_DEVICE = 5,        /* no gfx950 device: the library never falls back to the CPU */
    CVO_ERR_TIMEOUT = 6,          /* in-kernel inter-workgroup wait gave up */
    CVO_ERR_PADDING = 7,          /* status field of a gathered result record that stands for no pair (shard padding); never returned by a call */
    CVO_ERR_RANK_FAILED = 8       /* status field of the records a rank contributes to a gather when it could not prepare its own block (the rank's
                                     call returned the cause); never returned by a call */
};

enum { CVO_SLOT_FIXED = 0, CVO_SLOT_MOVING = 1, CVO_SLOT_PREVIOUS = 2 };   /* cvo.hpp:91-94 */

/* Hyper-parameters the reference hard-codes in the ctor (cvo.cpp:35-51). */
typedef struct cvo_params {
    float ell;        /* 0.15  initial kernel length-scale        cvo.cpp:35 */
    float sigma;      /* 0.1                                      cvo.cpp:36 */
    float sp_thres;   /* 8e-3  sparsification threshold           cvo.cpp:37 */
    float c;          /* 7.0   so(3) inner-product scale          cvo.cpp:38 */
    float d;          /* 7.0   R^3 inner-product scale            cvo.cpp:39 */
    float c_ell;      /* 200   colour kernel length-scale         cvo.cpp:41 */
    float c_sigma;    /* 1                                        cvo.cpp:42 */
    int   max_iter;   /* 2000                                     cvo.cpp:48 */
    float min_step;   /* 0.2                                      cvo.cpp:49 */
    float eps;        /* 5e-5  stop A                             cvo.cpp:50 */
    float eps_2;      /* 1e-5  stop B                             cvo.cpp:51 */
} cvo_params;

/* inn_p, cvo.hpp:52-80 */
typedef struct cvo_inn_p { float value; int num; int num_e; } cvo_inn_p;

/* one align() iteration as the kernel saw it (parity/diagnostics; optional) */
typedef struct cvo_trace_row {
    float  omega[3];
    float  v[3];
    int    nnz;
    int    candidates;   /* pairs that passed the conservative cull (>= nnz) */
    double B, C, D, E;
    float  step;
    float  ell;
    float  dist;         /* dist_se3 of the applied update, -1 if stop A fired first */
    int    pad_;
} cvo_trace_row;

typedef struct cvo_handle_s* cvo_handle;

const char* cvo_last_error(void);                 /* thread-local message of the last failing call */
int cvo_device_count(void);                       /* number of visible gfx950 devices */
int cvo_default_params(cvo_params* p);            /* ctor constants, cvo.cpp:35-51 */

/* ---- object lifetime:  cvo::cvo(const string& calib_file) / ~cvo()  cvo.cpp:18-74
 * (the calib file only feeds pcd_generator, which stays on the adaptor side) */
int cvo_create(const cvo_params* p /* NULL = defaults */, int device, cvo_handle* out);
int cvo_destroy(cvo_handle h);

/* ---- set_pcd(RGB, depth)  cvo.cpp:345-386, with the selected cloud handed in.
 * First call fills FIXED and sets `init`; later calls replace MOVING. */
int cvo_set_pcd(cvo_handle h, const float* xyz, const float* feat, int n);

/* ---- align()  cvo.cpp:763-821.  Runs entirely on the device from the handle's
 * R, T, ell (warm start, Q1/Q2) and leaves R, T, ell, transform, iter, A_nonzero
 * updated.  trace may be NULL. */
int cvo_align(cvo_handle h);
int cvo_align_traced(cvo_handle h, cvo_trace_row* trace, int trace_cap, int* trace_len);

/* ---- set_pcd(RGB_img, dep_img) with the reference's pcd_generator on the GPU  cvo.cpp:345-386
 * (SURVEY 8f next-1): gray image, 3-level gradient pyramid (pcd_generator.cpp:50-143), DSO pixel selection
 * (thirdparty/PixelSelector2.cpp:34-436, num_want points, srand(3141592) sub-sampling pattern), back-projection
 * and (B,G,R,dx,dy) features (pcd_generator.cpp:456-499, 590-612), written straight into the HBM cloud the
 * alignment reads.  bgr8: height x width x 3 bytes (cv::Mat CV_8UC3, row stride 3*width); depth16: height x
 * width uint16 (0 = invalid); cam = cvo::camera_info (data_type.h:33-39, read from the calib file by the ctor).
 * Same slot semantics as cvo_set_pcd.
 * The tracker hands one frame to two objects (cvo_odometry->match_odometry(frame), cvo_keyframe->match_keyframe(frame), local_tracker.cpp:356, 415) and the
 * selector is deterministic, so the two MOVING clouds are the same cloud: a handle that is given, on the same host thread and device, byte for byte the images
 * (and camera, num_want) the thread's previous generation was given takes a device copy of that cloud instead of generating it again (the staged images are kept
 * for the compare; a different frame differs within its first bytes).  Same bits either way (tests/test_gpu_pcd.py); CVO_HIP_SHARE_CLOUDS=0 generates always.
 * cvo_shared_cloud_count: how many of this handle's clouds were taken that way. */
typedef struct cvo_camera { float scaling_factor, fx, fy, cx, cy; } cvo_camera;
int cvo_set_pcd_images(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height,
                       const cvo_camera* cam);
int cvo_shared_cloud_count(cvo_handle h, int* count);
/* Start the generation of a frame's cloud AHEAD of the cvo_set_pcd_images that will ask for it.  The reference handles a frame strictly in sequence
 * (local_tracker.cpp:356 -> 375 -> 407 -> 415 -> 431) and every frame starts by waiting for its generator (cvo.cpp:348-366); frame t + 1's images do not depend on
 * frame t's alignments, so a caller that has them (run_SLAM.cpp:70-87 loads the next image right after this one) hands them over here, e.g. between the odometry
 * block and the keyframe block of frame t, and goes on: a worker thread with a stream and scratch of its own runs the generator beside frame t's keyframe
 * alignment (eight workgroups of 256 compute units), and frame t + 1's cvo_set_pcd_images -- on this host thread and device, given byte for byte these images,
 * this camera and the handle's num_want -- takes the finished cloud (waiting for it if need be; then the frame's second object takes its copy as it always did).
 * Returns at once.  The two images must stay valid and unchanged until that cvo_set_pcd_images (or the next cvo_stage_next_frame) has returned; one frame is
 * staged per host thread, a frame never asked for is dropped.  Nothing changes in what any call computes: same cloud bits (tests/test_gpu_pcd.py).
 * cvo_staged_frame_count: how many of this handle's clouds were taken from a staged generation. */
int cvo_stage_next_frame(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera* cam);
int cvo_staged_frame_count(cvo_handle h, int* count);
/* The tracker calls compute_innerproduct(tran = the transform match_* has just returned) behind every alignment (local_tracker.cpp:356-375, 415-431;
 * cvo.cpp:475-503).  An alignment of this handle can start that score block itself -- the score kernel is queued behind the align kernel with the transform
 * and ell taken from the pair's device-resident state, cvo_align returns as soon as the alignment is in, and cvo_compute_innerproduct only collects when it is
 * asked for exactly that transform on the same clouds at the same ell; any other request runs the score kernel as before.  Same numbers either way
 * (tests/test_gpu_tail_scores.py).  on = 0: never; 1: every alignment; 2 (the default): an alignment does when the handle's PREVIOUS alignment was followed by
 * exactly that question -- the tracker's two objects from their second frame on, a loop-closure object (cvo_compute_innerproduct_lc) never, and an alignment nobody
 * scores queues nothing after the first miss.  CVO_HIP_HANDLE_TAIL=kernel: the batches' way instead (the align launch's own tail, cvo_batch_set_tail_scores),
 * slower for one pair alone on its cooperating workgroups. */
int cvo_set_tail_scores(cvo_handle h, int on);
int cvo_queued_score_count(cvo_handle h, int* count);   /* score blocks of this handle that were answered by what an alignment had queued */
/* pcd_generator::num_want (3000, pcd_generator.cpp:22) for this handle's later cvo_set_pcd_images calls */
int cvo_set_num_want(cvo_handle h, int num_want);
/* match_odometry / match_keyframe taking the images, exactly as the reference's signatures do */
int cvo_match_odometry_images(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height,
                              const cvo_camera* cam, double transform_out[12]);
int cvo_match_keyframe_images(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height,
                              const cvo_camera* cam, double transform_out[12]);
/* a slot's cloud back in the reference layout (xyz: n x 3, feat: 5 channel-major arrays of n); *n = points, cap = room in
 * the arrays (in points); nothing is written if cap < n */
int cvo_get_cloud(cvo_handle h, int slot, float* xyz, float* feat, int cap, int* n);
/* get_fixed_frame_selected_points / get_moving_frame_selected_points  cvo.hpp:272-276: pixel (x, y) of every point of a
 * cloud made by cvo_set_pcd_images (n x 2 uint16); *n = 0 for clouds handed in by cvo_set_pcd */
int cvo_get_selected_points(cvo_handle h, int slot, unsigned short* px, int cap, int* n);

/* ---- match_odometry / match_keyframe  cvo.cpp:461-473, 563-576:
 * set_pcd + align; transform_out = 3x4 row-major double (Affine3d). */
int cvo_match_odometry(cvo_handle h, const float* xyz, const float* feat, int n, double transform_out[12]);
int cvo_match_keyframe(cvo_handle h, const float* xyz, const float* feat, int n, double transform_out[12]);

/* ---- function_inner_product(cloud_a, cloud_b)  cvo.cpp:388-459 and
 * se3_Hessian(cloud_a, cloud_b, inliers)  cvo.cpp:620-759, on the handle's slots;
 * tran_a (may be NULL) is applied to slot a's positions first, as
 * compute_innerproduct does at cvo.cpp:485-487.  Both use the handle's CURRENT ell. */
int cvo_function_inner_product(cvo_handle h, int slot_a, const float* tran_a, int slot_b, cvo_inn_p* out);
int cvo_se3_hessian(cvo_handle h, int slot_a, const float* tran_a, int slot_b, double H[36], int* inliers /* in/out, accumulated */);

/* The same two members as the reference declares them -- function_inner_product(point_cloud* cloud_a, point_cloud* cloud_b)
 * cvo.hpp:222 and se3_Hessian(point_cloud* cloud_a, point_cloud* cloud_b, int& inliers) cvo.hpp:260 -- on clouds the caller
 * holds in host memory (reference layout: n x 3 positions, 5 channel-major feature arrays); the handle's CURRENT ell applies
 * (cvo.cpp:395, 626) and *inliers accumulates (cvo.cpp:708).  Both clouds are staged in scratch buffers of the handle. */
int cvo_function_inner_product_clouds(cvo_handle h, const float* xyz_a, const float* feat_a, int n_a,
                                      const float* xyz_b, const float* feat_b, int n_b, cvo_inn_p* out);
int cvo_se3_hessian_clouds(cvo_handle h, const float* xyz_a, const float* feat_a, int n_a,
                           const float* xyz_b, const float* feat_b, int n_b, double H[36], int* inliers /* in/out */);

/* ---- compute_innerproduct  cvo.cpp:475-503 */
int cvo_compute_innerproduct(cvo_handle h, cvo_inn_p* inn_pre, cvo_inn_p* inn_post, double post_hessian[36],
                             const float tran[12], int* inliers, cvo_inn_p* inn_fixed_pcd,
                             cvo_inn_p* inn_moving_pcd, float* cos_angle);
/* ---- compute_innerproduct_lc  cvo.cpp:505-561 */
int cvo_compute_innerproduct_lc(cvo_handle h, cvo_inn_p* inn_prior, cvo_inn_p* inn_lc_prior, cvo_inn_p* inn_lc_pre,
                                cvo_inn_p* inn_lc_post, double post_hessian[36], const float prior_tran[12],
                                const float lc_prior_tran[12], const float lc_prior_tran_2[12],
                                const float lc_tran[12], int* inliers_svd, int* inliers_pnpransac,
                                cvo_inn_p* inn_fixed_pcd, cvo_inn_p* inn_moving_pcd, float* cos_angle);

/* ---- cloud-slot state machine  cvo.cpp:578-618 */
int cvo_update_fixed_pcd(cvo_handle h);                                   /* cvo.cpp:578-582 */
int cvo_update_previous_pcd(cvo_handle h);                                /* cvo.cpp:584-589 */
int cvo_reset_keyframe(cvo_handle h, const float odometry[12]);           /* cvo.cpp:591-604 */
int cvo_reset_transform(cvo_handle h, const float odometry[12]);          /* cvo.cpp:606-609 */
int cvo_reset_initial(cvo_handle h, const float odometry[12], float init_inverse_out[12]);   /* cvo.cpp:611-618 */

/* ---- getters / public members  cvo.hpp:139-144, 268-270 */
int cvo_get_fixed_and_moving_number(cvo_handle h, int* fixed_num, int* moving_num);
int cvo_get_iteration_number(cvo_handle h, int* iteration);
int cvo_get_A_nonzero(cvo_handle h, int* nonzero);
int cvo_get_transform(cvo_handle h, float transform[12]);
int cvo_get_prev_accum_transform(cvo_handle h, float prev_transform[12], float accum_transform[12]);
int cvo_get_init(cvo_handle h, int* init);
int cvo_get_first_frame(cvo_handle h, int* first_frame);
int cvo_set_first_frame(cvo_handle h, int first_frame);
/* R, T (row-major 3x3, 3) and ell are private in the reference but are carried
 * state between calls (Q1, Q2); exposed so callers/tests can pin them. */
int cvo_get_state(cvo_handle h, float R[9], float T[3], float* ell);
int cvo_set_state(cvo_handle h, const float R[9], const float T[3], float ell);
/* number of workgroups that cooperate on this handle's alignment (latency knob; 0 = auto) */
int cvo_set_workgroups(cvo_handle h, int workgroups_per_pair);

/* ---- adaptive-ell variant of the alignment: acvo::align, thirdparty/cvo/src/adaptive_cvo.cpp:490-555, with its
 * own constants (adaptive_cvo.cpp:27-46).  Per iteration the kernel matrices Axy, Axx and Ayy at the current ell give the length-scale
 * gradient dl (:154-272); ell moves by dl_step*dl inside [ell_min, ell_max], ell_max shrinking by 0.7 whenever it is hit (:538-545).
 * The reference's first loop never fills `sum_diff_yy_2` (:218-226 against :246-262): the rows of Ayy below num_fixed add nothing to
 * dl, only those from num_fixed on do; reproduced as is.  The reference does not build that file and has no caller for it
 * (thirdparty/cvo/CMakeLists.txt:66,77-81); its constants presume colour features scaled to [0,1] (c_ell = 0.5), which the shipped
 * generator does not produce (Q7).  A dense, untuned path: one workgroup per call.  Fresh-object semantics as after acvo::set_pcd
 * (ell = ell_init, ell_max as given, :476-477); R, T in: the start pose, out: the final one; transform_out = [R^T | -R^T T];
 * *iter = k at the break (left alone when max_iter is hit).  Clouds as for cvo_set_pcd.  trace may be NULL. */
typedef struct cvo_adaptive_params {
    float ell_init, ell_min, ell_max, dl_step;      /* 0.1, 0.0391, 0.15, 0.3        adaptive_cvo.cpp:27-32 */
    float sigma, sp_thres, c, d, c_ell, c_sigma;    /* 0.1, 8.315e-3, 7, 7, 0.5, 1   adaptive_cvo.cpp:35-42 (c_sp_thres = sp_thres) */
    int   max_iter; float min_step, eps, eps_2;     /* 2000, 0.2, 5e-5, 1e-5         adaptive_cvo.cpp:44-47 */
} cvo_adaptive_params;
typedef struct cvo_adaptive_row { float omega[3], v[3], dl, ell, step; int nnz_xy, nnz_xx, nnz_yy; } cvo_adaptive_row;
int cvo_adaptive_default_params(cvo_adaptive_params* p);
int cvo_adaptive_align(int device, const cvo_adaptive_params* p /* NULL = defaults */, const float* fixed_xyz, const float* fixed_feat, int n_fixed,
                       const float* moving_xyz, const float* moving_feat, int n_moving, float R_inout[9], float T_inout[3], float* ell_out,
                       float transform_out[12], int* iter, cvo_adaptive_row* trace, int trace_cap, int* trace_len);

/* ---- device self-test of the scalar closed forms the align kernel's epilogue runs once per iteration.  Each call
 * evaluates n cases on the device, one lane per case, with the very device functions the kernel calls:
 *   cubic_step: poly_solver + root selection + clamp, cvo.cpp:76-92,317-333   in: n x {c3, c2, c1, c0, min_step}  out: n steps
 *   exp_sek3:   Exp_SEK3 (K = 1) incl. the theta < 1e-6 branch, LieGroup.cpp:159-186   in: n x {omega[3], v[3], dt}
 *               out: n x {dR[9] row-major, dT[3]}
 *   dist_se3:   || logm([dR dT; 0 1]) ||_F, cvo.cpp:94-104   in: n x {dR[9], dT[3]}   out: n distances
 * Host pointers.  For known-answer tests that do not involve the CPU oracle (tests/test_gpu_closed_forms.py). */
int cvo_selftest_cubic_step(int device, int n, const float* coef_minstep, float* step_out);
int cvo_selftest_exp_sek3(int device, int n, const float* omega_v_dt, float* dR_dT_out);
int cvo_selftest_dist_se3(int device, int n, const float* dR_dT, float* dist_out);
/*   libm:       the device's float routines element by element: OCML's sinf, cosf, logf (logf: the gates, cvo.cpp:125-126) and the correctly rounded
 *               float sine and cosine Exp_SEK3 is evaluated with (LieGroup.cpp:174-175; cvo_math.hpp: sin_f32_cr, cos_f32_cr) and the correctly rounded logarithm of the
 *               gates (log_f32_cr)     in: n floats   out: n x {sinf, cosf, logf, sin_f32_cr, cos_f32_cr, log_f32_cr}
 *   pair_values: the pair arithmetic of se_kernel (cvo.cpp:166-175) by the four routes the align kernel has for it, for n pairs {fixed point at the
 *               origin with zero features; moving point y[3] with features g[5]} at length-scale `ell`:   in: n x {y0, y1, y2, g0, g1, g2, g3, g4}
 *               out: n x {a by the branchy full evaluation (dense fallback), a with the colour factor made once per list entry (lists outside the
 *               polynomial's range), a by the branch-free 12-term chain, a by the degree-7 polynomial with its rounding guard (the steady walk)} --
 *               a = 0 for a pair that is not a member of A; d2_d2c_out (may be NULL): n x {d2, d2c} as the device formed them.
 *               tests/test_gpu_pair_values.py: all four bit-equal to the oracle's (float)(s2*exp(-d2/(2.0*l*l))) sequence on >= 1e7 samples. */
int cvo_selftest_libm(int device, int n, const float* x, float* out6);
int cvo_selftest_pair_values(int device, const cvo_params* params /* NULL = defaults */, float ell, int n, const float* y_g, float* a_out, float* d2_d2c_out);

/* ======================= batched alignment (independent frame pairs) ===========
 * keyframe<->keyframe loop-closure candidates (keyframe_graph.cpp:622-731) and
 * offline batches are independent cvo::cvo objects; a batch runs all of them in
 * one persistent launch, `workgroups_per_pair` workgroups each. */
typedef struct cvo_batch_s* cvo_batch;

typedef struct cvo_pair_result {       /* what match_keyframe + the getters return, per pair */
    float transform[12];               /* cvo::transform after align(), cvo.cpp:817 */
    float R[9];
    float T[3];
    float ell;                         /* ell left behind (Q1) */
    int   iter;                        /* get_iteration_number (Q4: value of k at the break; max_iter if none) */
    int   A_nonzero;                   /* get_A_nonzero (Q5) */
    int   iterations_run;              /* loop trips executed = iter+1 on a break */
    int   status;                      /* CVO_OK / CVO_ERR_* for this pair */
    int   rebuilds;                    /* dense O(N*M) culls executed (candidate lists are reused between them) */
    int   dense_fallbacks;             /* culls whose candidates overflowed the lists (slow per-row path taken) */
} cvo_pair_result;

int cvo_batch_create(const cvo_params* p, int device, int max_pairs, cvo_batch* out);
int cvo_batch_destroy(cvo_batch b);
/* upload pair p (host arrays, reference layout); sets R=I, T=0, ell=params.ell (fresh-object semantics) */
int cvo_batch_set_pair(cvo_batch b, int p, const float* fixed_xyz, const float* fixed_feat, int n_fixed,
                       const float* moving_xyz, const float* moving_feat, int n_moving);
/* the same for pairs first .. first+count-1 in one hand-over (arrays of `count` pointers / sizes): the clouds' arrays are copied as
 * they are into one pinned staging block, ONE host-to-device copy brings them over and ONE kernel builds the device layout -- the
 * way to hand a whole batch over per step (64 pairs of ~3 k points = 12.6 MB).  The host arrays may be reused when the call returns. */
/* Caller-registered host memory.  cvo_host_register pins a range of the caller's memory and maps it into the devices' address space (hipHostRegister); clouds that
 * cvo_batch_set_pair(s) is handed from INSIDE a registered range (both arrays of the cloud) are not copied at all: the align launch that builds their device layout
 * reads them over PCIe where they lie.  For them -- and only for them -- the arrays must stay valid and unchanged until that launch has been waited for (cvo_batch_wait);
 * everything else about the calls is unchanged, and clouds outside registered ranges are staged as before (cvo.cpp:345-386 copies its inputs too).  Register once, e.g.
 * the pool a frame grabber or dataset reader fills; cvo_host_unregister before the memory is freed. */
int cvo_host_register(void* ptr, size_t bytes);
int cvo_host_unregister(void* ptr);
int cvo_batch_set_pairs(cvo_batch b, int first, int count, const float* const* fixed_xyz, const float* const* fixed_feat, const int* n_fixed,
                        const float* const* moving_xyz, const float* const* moving_feat, const int* n_moving);
/* warm start / carried ell for pair p (reset_initial + Q1) */
int cvo_batch_set_state(cvo_batch b, int p, const float R[9], const float T[3], float ell);
int cvo_batch_set_workgroups(cvo_batch b, int workgroups_per_pair /* 0 = auto: fill the CUs */);
/* cap on the workgroups one launch of this batch occupies (0 = no cap: up to the whole device).  Launches that are meant to run side
 * by side (several batches in flight on their own streams) each take a share; a launch with fewer pair slots than pairs hands its pairs
 * to the slots dynamically, so a slot is never idle behind the longest alignment. */
int cvo_batch_set_max_workgroups(cvo_batch b, int max_workgroups);
/* Adoption (off by default): in launches with one workgroup and one slot per pair, a workgroup that has finished its pair and finds
 * nothing queued on the device offers its help to a pair of the launch that still runs; from the next iteration on that pair runs on one
 * more workgroup (a pair can grow to four).  "Nothing queued" counts every align and score launch this library has submitted on the
 * device in this process, whatever its kind (not other processes, not other libraries' kernels).  Shortens the tail of a job whose
 * alignments take different numbers of iterations (33 ... 150); the results are those of any other workgroup count.
 * A pair only counts on a helper that has CONFIRMED the acceptance of its offer; when no confirmation comes within 50 us the owner takes the
 * acceptance back and carries on with the workgroups it has -- a helper that disappears cannot turn a healthy pair into CVO_ERR_TIMEOUT.
 * cvo_batch_last_adoptions: pairs of the last launch that were helped; cvo_batch_last_adoption_retractions: acceptances taken back. */
int cvo_batch_set_adoption(cvo_batch b, int on);
int cvo_batch_last_adoptions(cvo_batch b, int* pairs_helped);
int cvo_batch_last_adoption_retractions(cvo_batch b, int* retractions);
/* restore every pair's (R,T,ell) to what set_pair/set_state last gave it (bench loops re-run the same inputs) */
int cvo_batch_reset_states(cvo_batch b);
/* enqueue one persistent launch aligning pairs [0, n_pairs) on `stream` (a hipStream_t, NULL = the batch's own); asynchronous */
int cvo_batch_align_async(cvo_batch b, int n_pairs, void* stream);
/* wait for the launch and fetch results (n entries) */
int cvo_batch_wait(cvo_batch b, cvo_pair_result* results, int n);
/* has the last launch completed?  (never blocks; a caller that keeps several batches in flight can reuse whichever is done first --
 * alignments take data-dependent numbers of iterations, the oldest launch is not always the first to finish) */
int cvo_batch_done(cvo_batch b, int* done);
/* device time of the last launch in ms (HIP events on the launch stream), total loop trips it executed */
int cvo_batch_last_launch(cvo_batch b, float* kernel_ms, long long* iterations_total, long long* candidates_total);
/* nonzeros of the kernel matrix A (cvo.cpp:166-175) summed over every executed iteration of every pair of the last launch: the work the
 * reference's arithmetic is defined on (bench.py prices the kernel's instructions per nonzero with it) */
int cvo_batch_last_nonzeros(cvo_batch b, long long* nonzeros_total);
/* where the last launch spent its time: seconds summed over pairs, as seen by workgroup 0 of each pair:
 * [0] transform + list upkeep (cull, sort, refine)   [1] candidate phase   [2] candidates: workgroup reduction (incl. waiting
 * for the slowest wave)   [3] line-search phase   [4] candidates: exchange between the pair's workgroups   [5] scalar epilogue
 * [6] inside [0]: dense culls   [7] candidates: prologue   [8] inside [0]: row sorts   [9] candidates: the row walk */
int cvo_batch_last_phase_seconds(cvo_batch b, double seconds[10]);
/* seconds the first workgroup of each pair of the last launch spent on it (diagnostics: alignments take 33 ... 150 iterations of very different cost) */
int cvo_batch_last_pair_seconds(cvo_batch b, int n, double* seconds);
/* the same as spans on the device's own 100 MHz clock (one counter per device: the pairs of launches that ran side by side lie on one time axis), and the iteration
 * at which a finished workgroup joined the pair (0 = none; joined_at may be NULL): the drain of a job, pair by pair (scripts/gpu_timeline.py) */
int cvo_batch_last_pair_spans(cvo_batch b, int n, double* start_s, double* end_s, int* joined_at);
/* where the score block in the tail of the last launch spent its time (cvo_batch_set_tail_scores): seconds summed over pairs, workgroup 0 of each:
 * [0] final transform + list walk (inn_post, Hessian)   [1] the cull for inn_pre   [2] its walk + the rest   [3] all of it */
int cvo_batch_last_tail_seconds(cvo_batch b, double seconds[4]);
/* diagnostics: bit min(k, 63) of masks[i] is set when iteration k of pair i began with a dense cull (the candidate lists had gone stale, or were not there yet);
 * predicted[i] (may be NULL): the culls that built their lists around extrapolated positions */
int cvo_batch_last_cull_masks(cvo_batch b, int n, unsigned long long* masks, unsigned long long* predicted);
/* The last launch's results as records of CVO_RESULT_FLOATS floats {transform[12], iter, A_nonzero, iterations_run, status}: the
 * payload of the cross-GPU RCCL gather.  The align kernel writes them itself when a pair ends (no pack kernel behind the
 * launch): cvo_batch_result_records hands out the DEVICE address of the record table (valid once the launch's stream has drained; it
 * may move when a later launch has more pairs), cvo_batch_results_to_device copies the first n into a caller-owned DEVICE buffer
 * on `stream` (NULL = the launch's stream). */
#define CVO_RESULT_FLOATS 16
int cvo_batch_results_to_device(cvo_batch b, void* dst_device, int n, void* stream);
int cvo_batch_result_records(cvo_batch b, const void** records_device);

/* ======================= multi-GPU: shard the pairs, gather the SE(3) records ===========
 * Frame pairs are independent (loop-closure candidates keyframe_graph.cpp:622-731, offline batches): pair p of P goes to
 * a contiguous block per rank (cvo_shard_range: block sizes differ by at most one -- the reference's batch source is <= 10 loop-closure
 * candidates, i.e. 2,2,1,1,1,1,1,1 over 8 GPUs), its clouds live only on the owning GPU, and the ONLY exchange is one RCCL
 * all-gather of the CVO_RESULT_FLOATS-float result records over xGMI.  RCCL (librccl.so.1) is loaded when the first
 * communicator is made; a process that never shards does not need it.
 *
 * One process per GPU (torch.distributed / MPI launchers): rank 0 calls cvo_comm_unique_id, the host framework
 * broadcasts the CVO_COMM_ID_BYTES bytes, every rank calls cvo_comm_create (ncclCommInitRank).  One process, several
 * GPUs: cvo_comm_create_all (ncclCommInitAll).
 *
 * The gather.  An all-gather needs the same count on every rank, so every rank contributes a block of n_block =
 * cvo_shard_block(P, n_ranks) records: its n_valid own ones (written by the align kernel) and, behind them, records that stand for
 * no pair (status CVO_ERR_PADDING, the rest zero).  cvo_batch_gather_results_padded enqueues ONE ncclAllGather of n_block *
 * CVO_RESULT_FLOATS floats behind the launch on its stream: no host synchronisation between align and gather; recv_device
 * (n_ranks * n_block records, rank-major) is valid when that stream has drained (cvo_batch_wait); cvo_compact_records turns the
 * gathered table (on the host) into the P records in global pair order and reports the first non-zero status.
 * RULE: every rank enters the collective exactly once per step, whatever happened before.  A rank whose cvo_batch_align_async
 * failed passes that code as launch_status (n_valid is then ignored): all its records carry the status, its peers see it in the
 * gathered table instead of waiting for the rank forever.  A rank with no pairs (n_valid = 0) just sends padding.  Everything
 * that can fail inside the call (argument checks, buffer growth, the padding kernel) happens before the collective is posted, and a
 * rank on which it does fail STILL posts the all-gather -- from a block of CVO_ERR_RANK_FAILED records the communicator has held since
 * cvo_comm_create (1024 records; grown on demand) -- and then returns the error: its peers find status 8 in the gathered table
 * (cvo_compact_records' first_error).  Only when even that block is unavailable does the call return without entering the collective;
 * cvo_last_error() then ends in "(collective NOT entered)" and the peers have to be told out of band.
 * cvo_batch_gather_results(b, c, n, recv) is the n_valid = n_block = n form: n MUST be the same on every rank. */
#define CVO_COMM_ID_BYTES 128
typedef struct cvo_comm_s* cvo_comm;
int cvo_shard_range(int n_pairs_total, int rank, int n_ranks, int* first, int* count);    /* contiguous block of rank */
int cvo_shard_block(int n_pairs_total, int n_ranks);                                      /* records per rank in a gather: ceil(P / n_ranks) */
int cvo_comm_unique_id(char id[CVO_COMM_ID_BYTES]);
int cvo_comm_create(const char id[CVO_COMM_ID_BYTES], int n_ranks, int rank, int device, cvo_comm* out);
int cvo_comm_create_all(const int* devices, int n_devices, cvo_comm* out /* n_devices handles */);
int cvo_comm_info(cvo_comm c, int* n_ranks, int* rank);                                    /* ncclCommCount / ncclCommUserRank of the communicator */
int cvo_comm_destroy(cvo_comm c);
/* ORDER INVARIANT of the gathers: a communicator carries ONE gather per step, and every rank posts its gathers in the same (step) order.  Several may be
 * outstanding at once, on different streams (each behind the align launch it belongs to: bench.py keeps eight steps in flight), and a rank may post step k's
 * gather from whichever of its batch objects holds step k -- what has to agree across the ranks is the ORDER of the calls on the communicator, not the batch
 * object or the stream.  A rank that skips a step, or posts two steps in the other order, pairs its collective with the wrong one of its peers'.
 * cvo_comm_set_gather_stream(c, 1) (or CVO_HIP_GATHER_STREAM=1 when the communicator is made) posts every gather of the communicator to ONE stream of the
 * communicator's own instead -- behind an event of the align launch, with the launch's stream continuing behind the gather -- for a RCCL build that does not
 * take collectives of one communicator from several streams at once; cvo_batch_wait still returns with every rank's records in place.
 * cvo_comm_library_path: the file the bound RCCL was loaded from (a torch process resolves librccl.so.1 to torch's bundled copy, a C++ caller to /opt/rocm's). */
int cvo_comm_set_gather_stream(cvo_comm c, int on);
int cvo_comm_library_path(char* out, int cap);
int cvo_batch_gather_results(cvo_batch b, cvo_comm c, int n, void* recv_device);
int cvo_batch_gather_results_padded(cvo_batch b, cvo_comm c, int n_valid, int n_block, int launch_status, void* recv_device);
/* the block a rank would send (padding / status records in place), for launchers that run their own collective: DEVICE address,
 * complete in the order of the launch's stream */
int cvo_batch_padded_records(cvo_batch b, int n_valid, int n_block, int launch_status, const void** send_device);
int cvo_compact_records(const float* gathered_host, int n_pairs_total, int n_ranks, float* out_host /* n_pairs_total records */, int* first_error);
/* single-process form: one batch per device, all n_devices all-gathers inside one RCCL group (everything that can fail is checked
 * for every device before the group is opened) */
int cvo_gather_results(cvo_batch* batches, cvo_comm* comms, int n_devices, int n, void* const* recv_device);
int cvo_gather_results_padded(cvo_batch* batches, cvo_comm* comms, int n_devices, const int* n_valid, int n_block, const int* launch_status /* NULL = all CVO_OK */,
                              void* const* recv_device);

/* Convenience object for the single-process case: n_devices batches (one per GPU, max_pairs_per_device each), their
 * communicators and gather buffers.  cvo_multi_batch hands out device i's batch for cvo_batch_set_pair & co;
 * cvo_multi_align_async launches every device's batch (n pairs each) and enqueues the gather behind it;
 * cvo_multi_wait drains the streams and copies the gathered records (n_devices*n, device-major) to the host from
 * device `from_device`'s copy (every device holds all of them). */
typedef struct cvo_multi_s* cvo_multi;
int cvo_multi_create(const cvo_params* p, const int* devices, int n_devices, int max_pairs_per_device, cvo_multi* out);
int cvo_multi_destroy(cvo_multi m);
int cvo_multi_batch(cvo_multi m, int i, cvo_batch* out);
int cvo_multi_align_async(cvo_multi m, int n);
/* n_pairs[i] pairs on device i (0 = none); every device contributes max(n_pairs) records to the gather, padding behind its own */
int cvo_multi_align_async_v(cvo_multi m, const int* n_pairs);
int cvo_multi_wait(cvo_multi m, int from_device, float* records_out /* n_devices * max(n_pairs) * CVO_RESULT_FLOATS */);

/* Loop-closure verification of the aligned pairs (keyframe_graph.cpp:704-717): per pair the
 * compute_innerproduct_lc block (cvo.cpp:505-561: 6 inner products + 2 Hessians, lc_tran = the pair's
 * own align() result, ell = what that align() left behind, Q1) and the reference's accept rule.  All
 * pairs' 8 evaluations are ONE launch.  prior_tran / lc_prior_tran / lc_prior_tran_2: n row-major 3x4
 * Affine3f each (keyframe_graph.cpp: prior, lc_prior, lc_prior_2).  Waits for the align launch first. */
typedef struct cvo_lc_scores {
    cvo_inn_p inn_prior, inn_lc_prior, inn_pre, inn_post, inn_fixed_pcd, inn_moving_pcd;   /* cvo.cpp:539-551 */
    double post_hessian[36];           /* cvo.cpp:555 */
    int    inliers_svd;                /* cvo.cpp:554-555 */
    int    inliers_pnpransac;          /* cvo.cpp:557-558 */
    float  cos_angle;                  /* cvo.cpp:552 */
    int    accept;                     /* keyframe_graph.cpp:711-712: inn_post > inn_pre, inn_lc_prior, inn_prior and cos_angle >= 0.1 */
} cvo_lc_scores;
int cvo_batch_compute_innerproduct_lc(cvo_batch b, int n, const float* prior_tran, const float* lc_prior_tran,
                                      const float* lc_prior_tran_2, cvo_lc_scores* out);

/* The tracker's score block (cvo::compute_innerproduct, cvo.cpp:475-503; caller local_tracker.cpp:240-251) for the
 * first n pairs of the last align launch: tran = each pair's own align() result, ell = what that align() left
 * behind (Q1), inliers counted from 0.  enqueue queues ONE launch behind the align launch on its stream (the
 * transforms are read from the device-resident states, the host does not wait); results waits for it and finishes
 * the sums (inn_p count rule, Hessian scaling and eigenvalue shift) on the host.  compute = both. */
typedef struct cvo_track_scores {
    cvo_inn_p inn_pre, inn_post, inn_fixed_pcd, inn_moving_pcd;    /* cvo.cpp:489-497 */
    double post_hessian[36];           /* cvo.cpp:500 */
    int    inliers;                    /* cvo.cpp:708 */
    float  cos_angle;                  /* cvo.cpp:498 */
} cvo_track_scores;
int cvo_batch_enqueue_innerproduct(cvo_batch b, int n);
/* The same block answered by the align launch itself (off by default): when a pair's workgroup has finished the alignment it computes
 * inn_post and the Hessian terms from its resident candidate lists (the cloud re-transformed with the FINAL transform, cvo.cpp:485-487),
 * inn_pre from one cull of the untransformed cloud, and takes fip(fixed, fixed) / fip(moving, moving) from the clouds' tables of cached
 * self inner products -- no score launch has to find room beside the persistent align workgroups.  cvo_batch_innerproduct_results then
 * returns these; anything a workgroup could not answer (lists stale for the final transform, a pair run by several workgroups, a cloud
 * whose self product has not been computed yet) is computed by the score kernel at that point. */
int cvo_batch_set_tail_scores(cvo_batch b, int on);
/* which requests the last launch's workgroups answered themselves, per pair: bit 0 inn_pre, 1 inn_post, 2 inn_fixed_pcd, 3 inn_moving_pcd,
 * 4 the Hessian (diagnostics / tests; call before cvo_batch_innerproduct_results) */
int cvo_batch_last_tail_answers(cvo_batch b, int n, int* masks);
int cvo_batch_innerproduct_results(cvo_batch b, int n, cvo_track_scores* out);
int cvo_batch_compute_innerproduct(cvo_batch b, int n, cvo_track_scores* out);

#ifdef __cplusplus
}
#endif
#endif /* CVO_HIP_H */
