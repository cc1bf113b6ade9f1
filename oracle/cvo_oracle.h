/* ============================================================================
 * oracle/cvo_oracle.h  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C interface of the CPU restatement ("oracle") of CVO-SLAM's per-frame-pair
 * CVO alignment hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (cvo_slam_amd/, include/)
 * never includes, links or calls anything in oracle/.
 *
 * Parity status: PARITY UNPINNED by the reference's own tests -- the reference
 * ships no tests, golden vectors or fixtures for this path (SURVEY.md section 4,
 * section 8c) and cvo.cpp cannot be built here (needs Eigen, OpenCV, legacy TBB,
 * Boost; none present).  What IS pinned against real reference code:
 *   - the exact radius-search semantics (strict `<`, float squared-L2 expression
 *     order, result ordering) against the reference's vendored nanoflann.hpp,
 *     compiled from where it lies into oracle/_ref/ (oracle/ref_nanoflann.cpp);
 *   - closed-form pieces (cubic roots, SE(3) exp / log norm, 6x6 eigen shift)
 *     against numpy/scipy known answers (tests/golden/).
 *
 * Everything follows thirdparty/cvo/src/cvo.cpp and thirdparty/cvo/src/LieGroup.cpp
 * of the reference; each function cites the file:line it restates.
 * ========================================================================== */
#ifndef CVO_ORACLE_H
#define CVO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* hyper-parameters, defaults = reference ctor constants (cvo.cpp:35-51) */
typedef struct orc_params {
    float ell;        /* initial kernel length-scale 0.15       cvo.cpp:35 */
    float sigma;      /* 0.1                                    cvo.cpp:36 */
    float sp_thres;   /* 8e-3                                   cvo.cpp:37 */
    float c;          /* 7.0                                    cvo.cpp:38 */
    float d;          /* 7.0                                    cvo.cpp:39 */
    float c_ell;      /* 200                                    cvo.cpp:41 */
    float c_sigma;    /* 1                                      cvo.cpp:42 */
    int   max_iter;   /* 2000                                   cvo.cpp:48 */
    float min_step;   /* 0.2                                    cvo.cpp:49 */
    float eps;        /* 5e-5                                   cvo.cpp:50 */
    float eps_2;      /* 1e-5                                   cvo.cpp:51 */
} orc_params;

/* inn_p (cvo.hpp:52-80) */
typedef struct orc_inn_p { float value; int num; int num_e; } orc_inn_p;

/* one row per align() iteration, for per-iteration parity */
typedef struct orc_trace_row {
    float  omega[3];
    float  v[3];
    int    nnz;
    double B, C, D, E;
    float  step;
    float  ell;       /* ell used by this iteration */
    float  dist;      /* dist_se3(dR,dT); -1 if stop A fired before the update */
} orc_trace_row;

typedef struct orc_cvo orc_cvo;   /* opaque: one reference `cvo::cvo` object */

enum { ORC_SEARCH_BRUTE = 0, ORC_SEARCH_KDTREE = 1 };
enum { ORC_SLOT_FIXED = 0, ORC_SLOT_MOVING = 1, ORC_SLOT_PREVIOUS = 2 };

void     orc_default_params(orc_params* p);
orc_cvo* orc_create(const orc_params* p);                 /* cvo::cvo ctor  cvo.cpp:18-71 */
void     orc_destroy(orc_cvo* o);
/* search mode + thread count of the row loops (the TBB parallel_for stand-in) */
void     orc_set_exec(orc_cvo* o, int search_mode, int threads);

/* set_pcd (cvo.cpp:345-386) with the pcd_generator output handed in directly:
 * xyz = n x 3 AoS (cloud_t, data_type.h:30), feat = 5 channel-major arrays of n
 * (Eigen col-major Matrix<float,Dynamic,5>, data_type.h:75). */
int  orc_set_pcd(orc_cvo* o, const float* xyz, const float* feat, int n);
int  orc_align(orc_cvo* o, orc_trace_row* trace, int trace_cap, int* trace_len);   /* cvo.cpp:763-821 */
int  orc_match(orc_cvo* o, const float* xyz, const float* feat, int n, double transform_out[12]); /* match_odometry / match_keyframe cvo.cpp:461-473,563-576 */

void orc_update_fixed_pcd(orc_cvo* o);                    /* cvo.cpp:578-582 */
void orc_update_previous_pcd(orc_cvo* o);                 /* cvo.cpp:584-589 */
void orc_reset_keyframe(orc_cvo* o, const float odom[12]);/* cvo.cpp:591-604 */
void orc_reset_transform(orc_cvo* o, const float odom[12]);/* cvo.cpp:606-609 */
void orc_reset_initial(orc_cvo* o, const float odom[12], float out[12]); /* cvo.cpp:611-618 */

/* function_inner_product / se3_Hessian on (slot a, optional 3x4 row-major
 * transform applied to a's positions) vs slot b.   cvo.cpp:388-459, 620-759 */
int  orc_function_inner_product(orc_cvo* o, int slot_a, const float* tran_a, int slot_b, orc_inn_p* out);
int  orc_se3_hessian(orc_cvo* o, int slot_a, const float* tran_a, int slot_b,
                     double H[36], int* inliers, double H_raw_f64[36]);
/* compute_innerproduct cvo.cpp:475-503 */
int  orc_compute_innerproduct(orc_cvo* o, orc_inn_p* pre, orc_inn_p* post, double H[36],
                              const float tran[12], int* inliers, orc_inn_p* inn_fixed,
                              orc_inn_p* inn_moving, float* cos_angle);
/* compute_innerproduct_lc cvo.cpp:505-561 */
int  orc_compute_innerproduct_lc(orc_cvo* o, orc_inn_p* prior, orc_inn_p* lc_prior, orc_inn_p* lc_pre,
                                 orc_inn_p* lc_post, double H[36], const float prior_tran[12],
                                 const float lc_prior_tran[12], const float lc_prior_tran_2[12],
                                 const float lc_tran[12], int* inliers_svd, int* inliers_pnp,
                                 orc_inn_p* inn_fixed, orc_inn_p* inn_moving, float* cos_angle);

/* state access (public members + getters, cvo.hpp:139-144,268-270) */
void orc_get_state(const orc_cvo* o, float R[9], float T[3], float* ell, float transform[12],
                   int* iter, int* A_nonzero, int* num_fixed, int* num_moving);
void orc_set_state(orc_cvo* o, const float R[9], const float T[3], float ell);
void orc_get_accum(const orc_cvo* o, float prev_transform[12], float accum_transform[12]);
int  orc_get_init(const orc_cvo* o);
/* seconds this object's align() calls spent in: [0] KD-tree build (serial, cvo.cpp:135-136)  [1] radius searches + kernel values
 * (row-parallel, cvo.cpp:139-180)  [2] CSR assembly (serial: setFromTriplets, cvo.cpp:182-183)  [3] the two sparse sweeps */
void orc_get_timing(const orc_cvo* o, double seconds[4]);

/* ---- reference-noise variants (tests/golden/noise_envelope.json, scripts/make_noise_envelope.py).
 * The reference is not bit-reproducible and computes three things differently from the base oracle;
 * each flag switches one of them to a restatement of what the reference does, so that the spread of
 * the final pose over the variants bounds how far the real reference may sit from the base oracle:
 *   ORC_VAR_SHUFFLE  cross-row f64 sums of omega, v (cvo.cpp:226-230) and B..E (cvo.cpp:309-314) added in
 *                    a seeded random row order (the reference: TBB workers under a spin mutex, any order)
 *   ORC_VAR_F32_ROOTS  step = smallest positive eigenvalue with imag()==0 of the f32 companion matrix
 *                    (cvo.cpp:76-92, 324-330: MatrixXf::eigenvalues()), Francis QR in f32 (ref_noise.hpp)
 *   ORC_VAR_F32_LOGM   dist_se3 = Frobenius norm of an f32 Schur + Pade matrix logarithm of the f32 4x4
 *                    (cvo.cpp:94-104: Matrix4f::log().norm()), ref_noise.hpp
 * A fourth source, FMA contraction / re-association by an optimising compiler (the reference is built
 * -O3 -march=native with icpc, CMakeLists.txt:13), is a second BUILD of this same source
 * (oracle/Makefile: libcvo_oracle_fast.so, -O3 -march=native -ffp-contract=fast). */
/* Two more, inside a row / a feature vector (Eigen 3.3.7 read from its sources' structure; DESIGN.md section 2 has the derivation):
 *   ORC_VAR_ROW_LAZY16      compute_flow's `1/c*Ai*cross_xy` (cvo.cpp:222-223) as Eigen 3.3.7 evaluates it: rows with fewer than 16
 *                           nonzeros go through the coefficient-based lazy product, whose left factor `1/c*Ai` is evaluated first (alpha
 *                           folded into every a_j, then a sequential sum); longer rows through gebp's scalar tail (sequential sum, alpha after)
 *   ORC_VAR_ROW_ALPHA_FIRST alpha folded into a_j in every row
 *   ORC_VAR_ROW_STRIDE4/8   the row sum as 4 / 8 strided partial sums + horizontal add (re-associating vectoriser, packet redux)
 *   ORC_VAR_FEAT_HADD       (f_a-f_b).squaredNorm(), f_a.dot(f_b) (cvo.cpp:169, :662) as predux(Packet4f) + tail with SSE3's hadd:
 *                           ((t0+t1)+(t2+t3))+t4 -- the vectorised redux of a fixed size 5; the base order is the un-vectorised one
 *   ORC_VAR_FEAT_MOVEHL     the same with the movehl/add_ss predux: ((t0+t2)+(t1+t3))+t4 */
enum { ORC_VAR_SHUFFLE = 1, ORC_VAR_F32_ROOTS = 2, ORC_VAR_F32_LOGM = 4,
       ORC_VAR_ROW_LAZY16 = 8, ORC_VAR_ROW_STRIDE4 = 16, ORC_VAR_ROW_STRIDE8 = 32, ORC_VAR_ROW_ALPHA_FIRST = 64,
       ORC_VAR_FEAT_HADD = 128, ORC_VAR_FEAT_MOVEHL = 256 };
void  orc_set_variant(orc_cvo* o, int flags, unsigned long long shuffle_seed);
float orc_cubic_step_f32eig(float c3, float c2, float c1, float c0, float min_step);
float orc_dist_se3_f32logm(const float dR[9], const float dT[3]);
/* ref_noise.hpp instantiated in double (use_f32 = 0) or float: checked against numpy/scipy */
int   orc_test_eigenvalues(int n, const double* A /* n x n row-major */, double* re, double* im, int use_f32);
int   orc_test_logm(int n, const double* A, double* out, int use_f32);

/* ---- adaptive-ell variant: acvo::align of thirdparty/cvo/src/adaptive_cvo.cpp:490-555 with its own constants
 * (adaptive_cvo.cpp:27-46).  Per iteration three kernel matrices at the current ell -- Axy, Axx (fixed against itself), Ayy (the
 * transformed moving cloud against itself) -- give the length-scale gradient dl (adaptive_cvo.cpp:154-272); ell moves by dl_step*dl
 * inside [ell_min, ell_max], ell_max shrinking by 0.7 whenever it is hit (:539-546).  The reference's compute_flow never fills
 * `sum_diff_yy_2` in its first loop (:218-226 against :246-262), so the rows of Ayy below num_fixed add nothing to dl and only the
 * rows from num_fixed on (present when the moving cloud is the larger one) do: reproduced as is.  The reference does not build this
 * file (thirdparty/cvo/CMakeLists.txt:66,77-81) and ships no caller; c_sp_thres equals sp_thres in its constants and one value serves both. */
typedef struct orc_adaptive_params {
    float ell_init, ell_min, ell_max, dl_step;      /* 0.1, 0.0391, 0.15, 0.3   adaptive_cvo.cpp:27-32 */
    float sigma, sp_thres, c, d, c_ell, c_sigma;    /* 0.1, 8.315e-3, 7, 7, 0.5, 1   :35-42 */
    int   max_iter; float min_step, eps, eps_2;     /* 2000, 0.2, 5e-5, 1e-5   :44-47 */
} orc_adaptive_params;
typedef struct orc_adaptive_row {   /* one iteration */
    float omega[3], v[3], dl, ell, step; int nnz_xy, nnz_xx, nnz_yy;
} orc_adaptive_row;
void orc_adaptive_default_params(orc_adaptive_params* p);
/* fresh-object semantics (set_pcd resets ell = ell_init, ell_max = 0.15, adaptive_cvo.cpp:476-477; R = I, T = 0 unless given).
 * transform_out = final [R^T | -R^T T]; *iter = k at the break (unchanged if max_iter is hit); returns 0, or 2 for an empty cloud. */
int orc_adaptive_align(const orc_adaptive_params* p, const float* fixed_xyz, const float* fixed_feat, int n_fixed,
                       const float* moving_xyz, const float* moving_feat, int n_moving, float R_inout[9], float T_inout[3],
                       float* ell_out, float transform_out[12], int* iter, orc_adaptive_row* trace, int trace_cap, int* trace_len,
                       int search_mode, int threads);

/* one iteration's pieces, exposed for kernel-level parity */
int  orc_flow_once(orc_cvo* o, float omega[3], float v[3], int* nnz, double BCDE[4], float* step,
                   int* csr_rowptr /* nf+1 or NULL */, int* csr_col /* cap or NULL */,
                   float* csr_val /* cap or NULL */, int csr_cap);

/* ---- the pair arithmetic alone (known-answer checks of the device's pair functions, tests/test_gpu_pair_values.py).
 * orc_pair_values: cvo.cpp:166-175 on caller-supplied squared distances -- out[i] = a if (d2, d2c) is a member of A at `ell`, else 0;
 *                  k_out / ck_out (may be NULL): the two kernel factors as floats, whatever the gates say.
 * orc_libm_f32:    glibc's sinf (kind 0), cosf (1), logf (2: what the gates call, cvo.cpp:125-126), and the correctly rounded float sine (3) and cosine (4) the
 *                  oracle's Exp_SEK3 uses (LieGroup.cpp:174-175: `sin(float)` of the reference's libm, which differs between libms in the last bit),
 *                  element by element, to compare the device's routines with on the arguments the kernel produces. */
void orc_pair_values(const orc_params* p, float ell, int n, const float* d2, const float* d2c, float* a_out, float* k_out, float* ck_out);
void orc_libm_f32(int kind, int n, const float* in, float* out);

/* closed-form pieces */
float orc_cubic_step(float c3, float c2, float c1, float c0, float min_step);   /* cvo.cpp:76-92,317-333 */
void  orc_exp_sek3(const float omega[3], const float v[3], float dt, float dR[9], float dT[3]); /* LieGroup.cpp:159-186 */
float orc_dist_se3(const float dR[9], const float dT[3]);                        /* cvo.cpp:94-104 */
void  orc_hessian_regularize(const float Hin[36], int inliers, double Hout[36]); /* cvo.cpp:726-758 */
/* exact radius search used by ORC_SEARCH_KDTREE, for cross-checking vs brute force and nanoflann */
int   orc_radius_search(const float* cloud_xyz, int n, const float* query, float radius_sq,
                        int* out_idx, float* out_d2, int cap, int use_kdtree);

#ifdef __cplusplus
}
#endif
#endif
