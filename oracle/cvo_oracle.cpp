/* ============================================================================
 * oracle/cvo_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the CVO alignment hot path of bexilin/CVO-SLAM
 * (thirdparty/cvo/src/cvo.cpp, thirdparty/cvo/src/LieGroup.cpp).  See
 * cvo_oracle.h for the parity status ("parity unpinned" by reference tests;
 * radius-search semantics pinned against the reference's nanoflann.hpp via
 * oracle/_ref; closed forms pinned against numpy/scipy).
 *
 * Precision rules restated from the source:
 *   - cloud positions / features / R / T / ell: f32.
 *   - gate thresholds: double expression, float log (std::log(float) is chosen by
 *     overload resolution under `using namespace std`, cvo.hpp:44), stored f32
 *     (cvo.cpp:125-126, 395-396, 626-627).
 *   - k, ck: double exp of a double argument, product in double, stored f32
 *     (cvo.cpp:172-173).
 *   - omega, v: f32 per-row sums, f64 across rows, cast to f32 (cvo.cpp:222-235).
 *   - B..E: f32 beta..epsil, double polynomial, f64 sums (cvo.cpp:288-305).
 *   - Hessian: f32 everywhere (cvo.cpp:622,707).
 * Association order of 3-term sums: Eigen 3.3 evaluates FIXED-size reductions
 * of length 3 as t0+(t1+t2) (redux_novec_unroller halves the range) and
 * DYNAMIC-size ones sequentially; nanoflann's L2 tail loop is sequential
 * (nanoflann.hpp:403-406).  The reference itself is not bit-reproducible
 * (TBB reduction order, cvo.cpp:226-230,309-314), so these choices sit inside
 * its own noise; they are fixed here so that the HIP path can be compared
 * against ONE deterministic answer.
 *
 * Build: -O2 -ffp-contract=off (no FMA contraction), see oracle/Makefile.
 * ========================================================================== */
#include "cvo_oracle.h"
#include "ref_noise.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

// log(float) of the reference is its libm's float logarithm (cvo.cpp:125-126, 395-396); libms differ in its last bit, so oracle, host and device take the
// correctly rounded float: the double routine rounded once (cvo_math.hpp: log_f32_cr).  For the reference's constants glibc's logf gives the same floats.
static inline float log_cr(float x) { return (float)std::log((double)x); }

namespace {

// ---------------------------------------------------------------- small algebra
// 3x3 matrices are row-major float[9]; sums of three follow Eigen's fixed-size
// unrolled reduction t0 + (t1 + t2).
inline float sum3_fixed(float t0, float t1, float t2) { return t0 + (t1 + t2); }
inline float dot3_fixed(const float* a, const float* b) { return sum3_fixed(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline float dot3_seq(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

inline void mat3_mul(const float* A, const float* B, float* C) {   // C = A*B (C must not alias)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = sum3_fixed(A[i * 3 + 0] * B[0 * 3 + j], A[i * 3 + 1] * B[1 * 3 + j], A[i * 3 + 2] * B[2 * 3 + j]);
}
inline void mat3_vec(const float* A, const float* x, float* y) {   // y = A*x
    for (int i = 0; i < 3; ++i) y[i] = sum3_fixed(A[i * 3 + 0] * x[0], A[i * 3 + 1] * x[1], A[i * 3 + 2] * x[2]);
}
inline void cross3(const float* a, const float* b, float* c) {     // Eigen cross()
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
// skew, LieGroup.cpp:20-27
inline void skew3(const float* v, float* M) {
    M[0] = 0;     M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2];  M[4] = 0;     M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0];  M[8] = 0;
}
inline float norm3_fixed(const float* a) { return std::sqrt(sum3_fixed(a[0] * a[0], a[1] * a[1], a[2] * a[2])); }

// 3x4 row-major affine [L | t]
struct Aff { float m[12]; };
inline Aff aff_identity() { Aff a; std::memset(a.m, 0, sizeof(a.m)); a.m[0] = a.m[5] = a.m[10] = 1.f; return a; }
inline void aff_linear(const Aff& a, float* L) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L[r * 3 + c] = a.m[r * 4 + c]; }
inline void aff_apply(const float* m12, const float* p, float* out) {   // linear()*p + translation()
    for (int r = 0; r < 3; ++r)
        out[r] = sum3_fixed(m12[r * 4 + 0] * p[0], m12[r * 4 + 1] * p[1], m12[r * 4 + 2] * p[2]) + m12[r * 4 + 3];
}
// Affine3f * Affine3f  (4x4 product restricted to the top 3 rows; the 4-term sum of
// the translation column is a fixed-size-4 reduction (t0+t1)+(t2+t3) with t3 = a_t*1)
inline Aff aff_mul(const Aff& a, const Aff& b) {
    Aff c;
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k)
            c.m[r * 4 + k] = (a.m[r * 4 + 0] * b.m[0 * 4 + k] + a.m[r * 4 + 1] * b.m[1 * 4 + k]) + (a.m[r * 4 + 2] * b.m[2 * 4 + k] + a.m[r * 4 + 3] * 0.f);
        c.m[r * 4 + 3] = (a.m[r * 4 + 0] * b.m[0 * 4 + 3] + a.m[r * 4 + 1] * b.m[1 * 4 + 3]) + (a.m[r * 4 + 2] * b.m[2 * 4 + 3] + a.m[r * 4 + 3] * 1.f);
    }
    return c;
}
// Affine3f::inverse() (Affine mode: general 3x3 inverse by cofactors, then -Linv*t)
inline Aff aff_inverse(const Aff& a) {
    float L[9]; aff_linear(a, L);
    float c00 = L[4] * L[8] - L[5] * L[7], c01 = L[5] * L[6] - L[3] * L[8], c02 = L[3] * L[7] - L[4] * L[6];
    float det = sum3_fixed(L[0] * c00, L[1] * c01, L[2] * c02);
    float id = 1.f / det;
    float Li[9];
    Li[0] = c00 * id; Li[1] = (L[2] * L[7] - L[1] * L[8]) * id; Li[2] = (L[1] * L[5] - L[2] * L[4]) * id;
    Li[3] = c01 * id; Li[4] = (L[0] * L[8] - L[2] * L[6]) * id; Li[5] = (L[2] * L[3] - L[0] * L[5]) * id;
    Li[6] = c02 * id; Li[7] = (L[1] * L[6] - L[0] * L[7]) * id; Li[8] = (L[0] * L[4] - L[1] * L[3]) * id;
    Aff r;
    float t[3] = {a.m[3], a.m[7], a.m[11]}, nt[3];
    mat3_vec(Li, t, nt);
    for (int i = 0; i < 3; ++i) { for (int k = 0; k < 3; ++k) r.m[i * 4 + k] = Li[i * 3 + k]; r.m[i * 4 + 3] = -nt[i]; }
    return r;
}
// Affine3f::rotation(): orthogonal polar factor U*V^T of the linear part (Eigen:
// computeRotationScaling via JacobiSVD).  Restated as the Newton polar iteration
// X <- (X + X^-T)/2 in double, which converges to the same factor.
inline void polar_rotation(const float* L, float* Rout) {
    double X[9]; for (int i = 0; i < 9; ++i) X[i] = L[i];
    for (int it = 0; it < 32; ++it) {
        double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
        double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
        double inv[9];
        inv[0] = c00 / det; inv[1] = (X[2] * X[7] - X[1] * X[8]) / det; inv[2] = (X[1] * X[5] - X[2] * X[4]) / det;
        inv[3] = c01 / det; inv[4] = (X[0] * X[8] - X[2] * X[6]) / det; inv[5] = (X[2] * X[3] - X[0] * X[5]) / det;
        inv[6] = c02 / det; inv[7] = (X[1] * X[6] - X[0] * X[7]) / det; inv[8] = (X[0] * X[4] - X[1] * X[3]) / det;
        double delta = 0, Y[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            Y[r * 3 + c] = 0.5 * (X[r * 3 + c] + inv[c * 3 + r]);
            delta = std::max(delta, std::fabs(Y[r * 3 + c] - X[r * 3 + c]));
        }
        std::memcpy(X, Y, sizeof(X));
        if (delta < 1e-15) break;
    }
    for (int i = 0; i < 9; ++i) Rout[i] = (float)X[i];
}

// ---------------------------------------------------------------- clouds + search
struct Cloud {
    int n = 0;
    std::vector<float> xyz;    // n*3 AoS (cloud_t)
    std::vector<float> feat;   // 5*n channel-major (Eigen col-major N x 5)
};

// squared L2 exactly as nanoflann's L2_Adaptor::evalMetric for dim 3 (tail loop only):
// result = 0; result += d0*d0; result += d1*d1; result += d2*d2   (nanoflann.hpp:403-406)
inline float d2_nanoflann(const float* q, const float* p) {
    float result = 0.f;
    const float e0 = q[0] - p[0]; result += e0 * e0;
    const float e1 = q[1] - p[1]; result += e1 * e1;
    const float e2 = q[2] - p[2]; result += e2 * e2;
    return result;
}

// Exact KD-tree (leaf size 10, like cvo.cpp:135) used for the timed CPU baseline.
// Pruning is conservative in float arithmetic: a subtree is skipped only when the
// squared distance to its splitting plane is >= radius, and every point beyond the
// plane has d2 >= that value because float subtraction, multiplication and the
// addition of non-negative terms are all monotone.
struct KdTree {
    struct Node { int left, right; int lo, hi; int dim; float split; };   // children <0 => leaf over idx[lo,hi)
    const float* pts = nullptr;
    int n = 0;
    std::vector<int> idx;
    std::vector<Node> nodes;

    void build(const float* p, int count) {
        pts = p; n = count;
        idx.resize(n); std::iota(idx.begin(), idx.end(), 0);
        nodes.clear(); nodes.reserve(2 * (n / 5 + 1));
        if (n > 0) build_rec(0, n);
    }
    int build_rec(int lo, int hi) {
        int id = (int)nodes.size();
        nodes.push_back(Node{-1, -1, lo, hi, 0, 0.f});
        if (hi - lo <= 10) return id;
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int k = lo; k < hi; ++k)
            for (int d = 0; d < 3; ++d) { float v = pts[idx[k] * 3 + d]; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); }
        int dim = 0; float ext = mx[0] - mn[0];
        for (int d = 1; d < 3; ++d) if (mx[d] - mn[d] > ext) { ext = mx[d] - mn[d]; dim = d; }
        if (!(ext > 0.f)) return id;                       // all points identical: keep as leaf
        int mid = (lo + hi) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                         [&](int a, int b) { return pts[a * 3 + dim] < pts[b * 3 + dim]; });
        float split = pts[idx[mid] * 3 + dim];
        int l = build_rec(lo, mid);
        int r = build_rec(mid, hi);
        nodes[id].left = l; nodes[id].right = r; nodes[id].dim = dim; nodes[id].split = split;
        return id;
    }
    template <class F> void radius(const float* q, float r2, F&& emit) const {
        if (n == 0) return;
        int stack[64]; int sp = 0; stack[sp++] = 0;
        while (sp) {
            const Node& nd = nodes[stack[--sp]];
            if (nd.left < 0) {
                for (int k = nd.lo; k < nd.hi; ++k) {
                    int j = idx[k];
                    float d2 = d2_nanoflann(q, pts + j * 3);
                    if (d2 < r2) emit(j, d2);                // strict <, nanoflann.hpp:249-253
                }
                continue;
            }
            float diff = q[nd.dim] - nd.split;
            float plane = diff * diff;
            // left holds coords <= split, right holds coords >= split
            if (diff <= 0.f) { stack[sp++] = nd.left;  if (plane < r2) stack[sp++] = nd.right; }
            else             { stack[sp++] = nd.right; if (plane < r2) stack[sp++] = nd.left; }
        }
    }
};

struct Match { int j; float d2; };

// all j with d2(q, cloud_j) < r2.  order_by_index=true gives CSR (column) order
// (what Eigen::setFromTriplets leaves, cvo.cpp:182); false gives nanoflann's
// default sorted-by-distance order (nanoflann.hpp:1285-1286, IndexDist_Sorter).
inline void radius_matches(const float* cloud, int n, const KdTree* tree, const float* q, float r2,
                           bool order_by_index, std::vector<Match>& out) {
    out.clear();
    if (tree) {
        tree->radius(q, r2, [&](int j, float d2) { out.push_back(Match{j, d2}); });
    } else {
        for (int j = 0; j < n; ++j) { float d2 = d2_nanoflann(q, cloud + j * 3); if (d2 < r2) out.push_back(Match{j, d2}); }
    }
    if (order_by_index) std::sort(out.begin(), out.end(), [](const Match& a, const Match& b) { return a.j < b.j; });
    else std::stable_sort(out.begin(), out.end(), [](const Match& a, const Match& b) { return a.d2 < b.d2; });
}

// (f_a - f_b).squaredNorm() and f_a.dot(f_b) on fixed Matrix<float,5,1> (cvo.cpp:169, :662): an Eigen redux of length 5.
// Eigen 3.3.7, Core/Redux.h: the packet type of a fixed size 5 is Packet4f (find_best_packet halves Packet8f until the size divides or
// the half is itself), the traversal LinearVectorized with complete unrolling: predux(packet of t0..t3) + t4.  predux<Packet4f> is two
// _mm_hadd_ps with SSE3 on, (t0+t1)+(t2+t3), and _mm_movehl_ps + _mm_add_ss without, (t0+t2)+(t1+t3) (arch/SSE/PacketMath.h).  A build
// without vectorisation (EIGEN_DONT_VECTORIZE, or a target without SSE) takes redux_novec_unroller's binary split, (t0+t1)+(t2+(t3+t4)):
// the BASE order here and in the HIP kernels.  order: 0 base, 1 hadd (ORC_VAR_FEAT_HADD), 2 movehl (ORC_VAR_FEAT_MOVEHL).
// With features as the reference's generator makes them (8-bit B, G, R and half-integer gradients, pcd_generator.cpp:601-609) every term
// is a multiple of 1/4 below 2^18 and every partial sum is exact in f32: all three orders give the same bits.  The bench's synthetic
// pairs carry gradients of a float gray image, and there the order shows (tests/test_oracle_noise.py, tests/golden/noise_envelope.json).
inline float redux5(const float (&t)[5], int order) {
    if (order == 1) return ((t[0] + t[1]) + (t[2] + t[3])) + t[4];
    if (order == 2) return ((t[0] + t[2]) + (t[1] + t[3])) + t[4];
    return (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
}
inline float feat_d2(const float* fa, int na, int i, const float* fb, int nb, int j, int order = 0) {
    float t[5];
    for (int c = 0; c < 5; ++c) { float e = fa[c * na + i] - fb[c * nb + j]; t[c] = e * e; }
    return redux5(t, order);
}
inline float feat_dot(const float* fa, int na, int i, const float* fb, int nb, int j, int order = 0) {
    float t[5];
    for (int c = 0; c < 5; ++c) t[c] = fa[c * na + i] * fb[c * nb + j];
    return redux5(t, order);
}

}  // namespace

// =============================================================================
struct orc_cvo {
    orc_params p;
    // members of cvo::cvo (cvo.hpp:87-144)
    std::unique_ptr<Cloud> fixed, moving, previous;
    bool pre_pc_init = false, init = false, first_frame = true;
    int num_fixed = 0, num_moving = 0;
    float ell;
    float R[9], T[3];
    Aff transform, prev_transform, accum_transform;
    float omega[3] = {0, 0, 0}, v[3] = {0, 0, 0}, step = 0;
    int iter = 0;          // Q4: the reference leaves this uninitialised / stale when MAX_ITER is hit
    int A_nonzero = 0;
    std::vector<float> cloud_y;                    // transformed moving positions (cvo.cpp:377)
    std::vector<int> A_rowptr, A_col; std::vector<float> A_val;   // Eigen::SparseMatrix<float,RowMajor>
    double last_BCDE[4] = {0, 0, 0, 0};
    int search_mode = ORC_SEARCH_BRUTE, threads = 1;
    double t_sec[4] = {0, 0, 0, 0};                // where align() spent its time: KD-tree build | radius searches + kernel values (row-parallel) | CSR assembly (serial) | flow + step-size sweeps
    int variant = 0;                               // ORC_VAR_* (reference-noise variants, cvo_oracle.h)
    unsigned long long shuffle_seed = 0, shuffle_calls = 0;
};

namespace {

// update_tf, cvo.cpp:106-110: transform = [R^T, -R^T*T]
void update_tf(orc_cvo* o) {
    float Rt[9], nRt[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Rt[r * 3 + c] = o->R[c * 3 + r]; nRt[r * 3 + c] = -o->R[c * 3 + r]; }
    float t[3]; mat3_vec(nRt, o->T, t);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) o->transform.m[r * 4 + c] = Rt[r * 3 + c]; o->transform.m[r * 4 + 3] = t[r]; }
}

// transform_pcd, cvo.cpp:336-341 (always from the UNtransformed moving positions)
void transform_pcd(orc_cvo* o) {
    const Cloud& m = *o->moving;
    o->cloud_y.resize((size_t)m.n * 3);
    for (int j = 0; j < m.n; ++j) aff_apply(o->transform.m, &m.xyz[(size_t)j * 3], &o->cloud_y[(size_t)j * 3]);
}

// se_kernel, cvo.cpp:122-184 (and adaptive_cvo.cpp:92-147, which takes the two clouds as arguments): sparse kernel matrix of
// cloud a (positions a_xyz, features of cloud A) against cloud b, as CSR with ascending columns.
void se_kernel_clouds(orc_cvo* o, const float* a_xyz, const Cloud& A, const float* b_xyz, const Cloud& B, float l, float s2,
                      std::vector<int>& rowptr, std::vector<int>& colv, std::vector<float>& valv) {
    const orc_params& P = o->p;
    const int N = A.n, M = B.n;
    const int feat_order = (o->variant & ORC_VAR_FEAT_HADD) ? 1 : ((o->variant & ORC_VAR_FEAT_MOVEHL) ? 2 : 0);
    // float d2_thres = -2.0*l*l*log(sp_thres/s2);            cvo.cpp:125
    const float d2_thres = (float)(-2.0 * l * l * (double)log_cr(P.sp_thres / s2));
    // float d2_c_thres = -2.0*c_ell*c_ell*log(sp_thres/c_sigma/c_sigma);   cvo.cpp:126
    const float d2_c_thres = (float)(-2.0 * P.c_ell * P.c_ell * (double)log_cr(P.sp_thres / P.c_sigma / P.c_sigma));

    const auto tk0 = std::chrono::steady_clock::now();
    KdTree tree; const KdTree* tp = nullptr;
    if (o->search_mode == ORC_SEARCH_KDTREE) { tree.build(b_xyz, M); tp = &tree; }   // rebuilt every call, cvo.cpp:135-136
    const auto tk1 = std::chrono::steady_clock::now();

    std::vector<std::vector<int>> cols(N); std::vector<std::vector<float>> vals(N);
#pragma omp parallel num_threads(o->threads)
    {
        std::vector<Match> ms;
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < N; ++i) {
            radius_matches(b_xyz, M, tp, &a_xyz[(size_t)i * 3], d2_thres, true, ms);
            for (const Match& mt : ms) {
                const float d2 = mt.d2;
                if (d2 < d2_thres) {                                              // cvo.cpp:166
                    const float d2_color = feat_d2(A.feat.data(), N, i, B.feat.data(), M, mt.j, feat_order);   // cvo.cpp:169
                    if (d2_color < d2_c_thres) {                                  // cvo.cpp:171
                        const float k = (float)(s2 * std::exp(-d2 / (2.0 * l * l)));                         // cvo.cpp:172
                        const float ck = (float)(P.c_sigma * P.c_sigma * std::exp(-d2_color / (2.0 * P.c_ell * P.c_ell)));   // cvo.cpp:173
                        const float a = ck * k;                                   // cvo.cpp:174
                        if (a > P.sp_thres) { cols[i].push_back(mt.j); vals[i].push_back(a); }   // cvo.cpp:175
                    }
                }
            }
        }
    }
    const auto tk2 = std::chrono::steady_clock::now();
    // A.setFromTriplets + makeCompressed (cvo.cpp:182-183): CSR, columns ascending
    rowptr.assign(N + 1, 0);
    for (int i = 0; i < N; ++i) rowptr[i + 1] = rowptr[i] + (int)cols[i].size();
    colv.resize(rowptr[N]); valv.resize(rowptr[N]);
    for (int i = 0; i < N; ++i) {
        std::copy(cols[i].begin(), cols[i].end(), colv.begin() + rowptr[i]);
        std::copy(vals[i].begin(), vals[i].end(), valv.begin() + rowptr[i]);
    }
    const auto tk3 = std::chrono::steady_clock::now();
    o->t_sec[0] += std::chrono::duration<double>(tk1 - tk0).count(); o->t_sec[1] += std::chrono::duration<double>(tk2 - tk1).count();
    o->t_sec[2] += std::chrono::duration<double>(tk3 - tk2).count();
}
void se_kernel(orc_cvo* o, float l, float s2) {
    se_kernel_clouds(o, o->fixed->xyz.data(), *o->fixed, o->cloud_y.data(), *o->moving, l, s2, o->A_rowptr, o->A_col, o->A_val);
}

struct SweepTimer {   // the two sparse sweeps (compute_flow after se_kernel, compute_step_size)
    orc_cvo* o; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit SweepTimer(orc_cvo* p) : o(p) {}
    ~SweepTimer() { o->t_sec[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
// One row of compute_flow: the f32 values of (1/c*Ai*cross_xy) and (1/d*Ai*diff_yx) (cvo.cpp:213-223), Ai = the row's nonzeros in column
// order.  What Eigen 3.3.7 does with `scalar * MatrixXf(1,n) * MatrixXf(n,3)` (GeneralMatrixMatrix.h, generic_product_impl<..., GemmProduct>::evalTo):
//   * n + 1 + 3 < 20 (EIGEN_GEMM_TO_COEFFBASED_THRESHOLD), i.e. rows of fewer than 16 nonzeros: the coefficient-based lazy product.  Its
//     evaluator nests the left factor through nested_eval<Lhs, Dynamic>, which EVALUATES `1/c*Ai` into a temporary (the scalar is folded into
//     every a_j first), and each of the three coefficients is (lhs.row(0).transpose().cwiseProduct(rhs.col(k))).sum(): a row block of a
//     column-major matrix has no packet access, so the redux is DefaultTraversal -- sequential in j;
//   * 16 nonzeros and more: general_matrix_matrix_product with actualAlpha = 1/c taken out of the expression by blas_traits.  The result
//     has 3 columns, fewer than the kernel's nr = 4, and one row: gebp's "remaining columns, remaining rows" loop, C0 = sum_k A0*B_0 + C0
//     sequential in k, then res += alpha*C0 -- alpha AFTER the sum.  (The compiler may contract the multiply-add: the FMA build.)
// The base oracle and the HIP kernels: sequential, alpha after the sum, for every row.  Variants (ORC_VAR_ROW_*): LAZY16 = the rule above;
// ALPHA_FIRST = alpha folded into a_j in every row; STRIDE4 / STRIDE8 = the sum as 4 / 8 strided partial sums and a horizontal add
// ((p0+p1)+(p2+p3); 8: halves added first, like predux<Packet8f>) -- what a vectorising compiler allowed to re-associate (icpc's default
// fp-model) or an Eigen with a packet redux on this path would produce.
static void row_flow(const orc_cvo* o, int i, float inv_c, float inv_d, float (&rw)[3], float (&rv)[3]) {
    const Cloud& X = *o->fixed;
    const float* xi = &X.xyz[(size_t)i * 3];
    const int e0 = o->A_rowptr[i], e1 = o->A_rowptr[i + 1], n = e1 - e0;
    const bool alpha_first = (o->variant & ORC_VAR_ROW_ALPHA_FIRST) || ((o->variant & ORC_VAR_ROW_LAZY16) && n < 16);
    const int stride = (o->variant & ORC_VAR_ROW_STRIDE8) ? 8 : ((o->variant & ORC_VAR_ROW_STRIDE4) ? 4 : 1);
    float pw[8][3], pv[8][3];
    for (int q = 0; q < 8; ++q) for (int k = 0; k < 3; ++k) { pw[q][k] = 0.f; pv[q][k] = 0.f; }
    for (int e = e0; e < e1; ++e) {
        const float* yj = &o->cloud_y[(size_t)o->A_col[e] * 3];
        const float a = o->A_val[e];
        const float aw = alpha_first ? inv_c * a : a, av = alpha_first ? inv_d * a : a;
        float cr[3]; cross3(xi, yj, cr);                                          // cvo.cpp:216
        const int q = (e - e0) % stride;
        for (int k = 0; k < 3; ++k) { pw[q][k] += aw * cr[k]; pv[q][k] += av * (yj[k] - xi[k]); }   // cvo.cpp:217, 222-223
    }
    for (int k = 0; k < 3; ++k) {
        float sw = pw[0][k], sv = pv[0][k];
        if (stride == 4) { sw = (pw[0][k] + pw[1][k]) + (pw[2][k] + pw[3][k]); sv = (pv[0][k] + pv[1][k]) + (pv[2][k] + pv[3][k]); }
        if (stride == 8) {
            sw = ((pw[0][k] + pw[4][k]) + (pw[1][k] + pw[5][k])) + ((pw[2][k] + pw[6][k]) + (pw[3][k] + pw[7][k]));
            sv = ((pv[0][k] + pv[4][k]) + (pv[1][k] + pv[5][k])) + ((pv[2][k] + pv[6][k]) + (pv[3][k] + pv[7][k]));
        }
        rw[k] = alpha_first ? sw : inv_c * sw; rv[k] = alpha_first ? sv : inv_d * sv;
    }
}
void compute_flow(orc_cvo* o) {
    se_kernel(o, o->ell, o->p.sigma * o->p.sigma);                                // cvo.cpp:189
    SweepTimer sweep_timer(o);
    const Cloud& X = *o->fixed;
    const int N = X.n;
    const float inv_c = 1 / o->p.c, inv_d = 1 / o->p.d;                           // `1/c`, `1/d` are float
    double dw[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
    long nnz = 0;
    if (o->variant & ORC_VAR_SHUFFLE) {
        // the same per-row f32 sums, added across rows in a seeded random order (cvo.cpp:226-230 runs under a spin
        // mutex in whatever order the TBB workers arrive)
        std::vector<float> rw((size_t)N * 3), rv((size_t)N * 3);
        for (int i = 0; i < N; ++i) {
            float a3[3], b3[3]; row_flow(o, i, inv_c, inv_d, a3, b3);
            for (int k = 0; k < 3; ++k) { rw[(size_t)i * 3 + k] = a3[k]; rv[(size_t)i * 3 + k] = b3[k]; }
        }
        std::vector<int> perm(N); std::iota(perm.begin(), perm.end(), 0);
        std::mt19937_64 rng(o->shuffle_seed + 0x9E3779B97F4A7C15ull * (++o->shuffle_calls));
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int i : perm) for (int k = 0; k < 3; ++k) { dw[k] += (double)rw[(size_t)i * 3 + k]; dv[k] += (double)rv[(size_t)i * 3 + k]; }
        for (int k = 0; k < 3; ++k) { o->omega[k] = (float)dw[k]; o->v[k] = (float)dv[k]; }
        o->A_nonzero = o->A_rowptr[N];
        return;
    }
#pragma omp parallel num_threads(o->threads)
    {
        double lw[3] = {0, 0, 0}, lv[3] = {0, 0, 0}; long ln = 0;
#pragma omp for schedule(static)
        for (int i = 0; i < N; ++i) {
            float rw[3], rv[3]; row_flow(o, i, inv_c, inv_d, rw, rv);             // cvo.cpp:213-223
            for (int k = 0; k < 3; ++k) { lw[k] += (double)rw[k]; lv[k] += (double)rv[k]; }
            ln += o->A_rowptr[i + 1] - o->A_rowptr[i];
        }
#pragma omp critical
        { for (int k = 0; k < 3; ++k) { dw[k] += lw[k]; dv[k] += lv[k]; } nnz += ln; }   // spin_mutex block, cvo.cpp:226-230
    }
    for (int k = 0; k < 3; ++k) { o->omega[k] = (float)dw[k]; o->v[k] = (float)dv[k]; }   // cvo.cpp:234-235
    o->A_nonzero = (int)nnz;
}

}  // namespace

// poly_solver (cvo.cpp:76-92) + root selection (cvo.cpp:324-330) for the cubic
// c3 t^3 + c2 t^2 + c1 t + c0.  The reference forms the companion matrix of the
// monic polynomial with f32 coefficients -(coef/coef(0)) and takes f32 eigenvalues;
// only eigenvalues with imag()==0 qualify.  Restated in closed form on the same
// f32 monic coefficients (double arithmetic): real roots of a real cubic = the
// eigenvalues with zero imaginary part.  coef(0)==0 gives inf/NaN entries, no
// eigenvalue qualifies, and the caller falls back to min_step.
// Real roots of the monic cubic t^3 + a t^2 + b t + c (double).  One root by Newton
// from 0, falling back to a bracketed Newton iteration (always converges: f(-R) < 0 < f(R)
// for the Cauchy bound R), the other two from the deflated quadratic (deflation direction chosen
// by the root's size so no cancellation), each polished on the full cubic.  Plain
// Cardano loses the sign of the discriminant when the roots differ by many orders
// of magnitude.
static int cubic_real_roots(double a, double b, double c, double* roots) {
    // fast path: plain Newton from t = 0 (the step the line search wants is normally the small
    // root next to 0); accepted only if it converges, otherwise the bracketed iteration below.
    double x = 0.0;
    bool conv = false;
    for (int it = 0; it < 12; ++it) {
        const double f = ((x + a) * x + b) * x + c, df = (3.0 * x + 2.0 * a) * x + b;
        if (f == 0.0) { conv = true; break; }
        const double xn = x - f / df;
        if (!(df != 0.0) || !std::isfinite(xn)) break;
        if (std::fabs(xn - x) <= 1e-15 * std::fabs(xn)) { x = xn; conv = true; break; }
        x = xn;
    }
    if (!conv) {
        const double R = 1.0 + std::fmax(std::fabs(a), std::fmax(std::fabs(b), std::fabs(c)));
        double lo = -R, hi = R;
        x = 0.0;
        for (int it = 0; it < 200; ++it) {
            const double f = ((x + a) * x + b) * x + c, df = (3.0 * x + 2.0 * a) * x + b;
            if (f == 0.0) break;
            if (f < 0) lo = x; else hi = x;
            double xn = x - f / df;
            if (!(df != 0.0) || !(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
            if (std::fabs(xn - x) <= 1e-16 * std::fabs(xn) || xn == x) { x = xn; break; }
            x = xn;
        }
    }
    const double r = x;
    roots[0] = r;
    double p, q;                                   // t^2 + p t + q
    if (std::fabs(r) * std::fabs(r) * std::fabs(r) > std::fabs(c)) { q = -c / r; p = (q - b) / r; }   // large root: divide from the constant term up
    else { p = a + r; q = b + p * r; }                                              // small root: synthetic division from the top
    const double disc = p * p - 4.0 * q;
    if (!(disc >= 0.0)) return 1;
    const double s = -0.5 * (p + std::copysign(std::sqrt(disc), p));
    double r2 = s, r3 = (s != 0.0) ? q / s : 0.0;
    double* rr[2] = {&r2, &r3};
    for (int k = 0; k < 2; ++k) {
        double t = *rr[k];
        for (int it = 0; it < 2; ++it) {          // the deflated roots are already good to ~1e-15; two Newton steps on the full cubic
            const double f = ((t + a) * t + b) * t + c, df = (3.0 * t + 2.0 * a) * t + b;
            const double tn = t - f / df;
            if (!(df != 0.0) || !std::isfinite(tn)) break;
            t = tn;
        }
        *rr[k] = t;
    }
    roots[1] = r2; roots[2] = r3;
    return 3;
}

void orc_pair_values(const orc_params* p, float ell, int n, const float* d2v, const float* d2cv, float* a_out, float* k_out, float* ck_out) {
    const orc_params& P = *p;
    const float l = ell, s2 = P.sigma * P.sigma;                                        // se_kernel(ell, sigma*sigma), cvo.cpp:189
    const float d2_thres = (float)(-2.0 * l * l * (double)log_cr(P.sp_thres / s2));   // cvo.cpp:125
    const float d2_c_thres = (float)(-2.0 * P.c_ell * P.c_ell * (double)log_cr(P.sp_thres / P.c_sigma / P.c_sigma));   // cvo.cpp:126
    for (int i = 0; i < n; ++i) {
        const float d2 = d2v[i], d2_color = d2cv[i];
        const float k = (float)(s2 * std::exp(-d2 / (2.0 * l * l)));                    // cvo.cpp:172
        const float ck = (float)(P.c_sigma * P.c_sigma * std::exp(-d2_color / (2.0 * P.c_ell * P.c_ell)));   // cvo.cpp:173
        float a = 0.f;
        if (d2 < d2_thres && d2_color < d2_c_thres) {                                   // cvo.cpp:166, 171
            const float ak = ck * k;                                                    // cvo.cpp:174
            if (ak > P.sp_thres) a = ak;                                                // cvo.cpp:175
        }
        a_out[i] = a;
        if (k_out) k_out[i] = k;
        if (ck_out) ck_out[i] = ck;
    }
}
void orc_libm_f32(int kind, int n, const float* in, float* out) {
    for (int i = 0; i < n; ++i)
        out[i] = kind == 0 ? std::sin(in[i]) : kind == 1 ? std::cos(in[i]) : kind == 2 ? std::log(in[i])                    // the float overloads
               : kind == 3 ? (float)std::sin((double)in[i]) : kind == 4 ? (float)std::cos((double)in[i]) : log_cr(in[i]);      // the correctly rounded floats (Exp_SEK3, the gates)
}

extern "C" float orc_cubic_step(float c3, float c2, float c1, float c0, float min_step) {
    const float p1f = c2 / c3, p2f = c1 / c3, p3f = c0 / c3;                      // (coef/coef(0)).segment(1,3), cvo.cpp:86
    float best = std::numeric_limits<float>::max();
    if (std::isfinite(p1f) && std::isfinite(p2f) && std::isfinite(p3f)) {
        double roots[3];
        const int nr = cubic_real_roots(p1f, p2f, p3f, roots);
        for (int k = 0; k < nr; ++k) {
            const float tf = (float)roots[k];
            if (tf > 0 && tf < best) best = tf;                                   // cvo.cpp:326-327
        }
    }
    float step = (best == std::numeric_limits<float>::max()) ? min_step : best;   // cvo.cpp:330
    step = step > 0.8 ? (float)0.8 : step;                                        // cvo.cpp:333
    return step;
}

namespace {

// compute_step_size, cvo.cpp:239-334
void compute_step_size(orc_cvo* o) {
    SweepTimer sweep_timer(o);
    const Cloud& X = *o->fixed;
    const int N = X.n, M = o->moving->n;
    float Oh[9]; skew3(o->omega, Oh);                                             // cvo.cpp:241
    float O2[9], O3[9], O4[9], Ov[3], O2v[3], O3v[3];
    mat3_mul(Oh, Oh, O2); mat3_mul(O2, Oh, O3); mat3_mul(O3, Oh, O4);             // left-to-right products
    mat3_vec(Oh, o->v, Ov); mat3_vec(O2, o->v, O2v); mat3_vec(O3, o->v, O3v);

    std::vector<float> xiz((size_t)M * 3), xi2z((size_t)M * 3), xi3z((size_t)M * 3), xi4z((size_t)M * 3), nrm(M), mdot(M), econst(M);
#pragma omp parallel for num_threads(o->threads) schedule(static)
    for (int j = 0; j < M; ++j) {                                                 // cvo.cpp:252-264
        const float* y = &o->cloud_y[(size_t)j * 3];
        float t[3];
        cross3(o->omega, y, t);         for (int k = 0; k < 3; ++k) xiz[(size_t)j * 3 + k] = t[k] + o->v[k];
        mat3_vec(O2, y, t);             for (int k = 0; k < 3; ++k) xi2z[(size_t)j * 3 + k] = t[k] + Ov[k];
        mat3_vec(O3, y, t);             for (int k = 0; k < 3; ++k) xi3z[(size_t)j * 3 + k] = t[k] + O2v[k];
        mat3_vec(O4, y, t);             for (int k = 0; k < 3; ++k) xi4z[(size_t)j * 3 + k] = t[k] + O3v[k];
        const float* a = &xiz[(size_t)j * 3]; const float* b = &xi2z[(size_t)j * 3]; const float* c3 = &xi3z[(size_t)j * 3];
        nrm[j] = dot3_seq(a, a);                                                  // dynamic-row squaredNorm: sequential
        mdot[j] = -dot3_seq(a, b);
        econst[j] = dot3_seq(b, b) + 2 * dot3_seq(a, c3);
    }

    const float temp_coef = (float)(1 / (2.0 * o->ell * o->ell));                 // cvo.cpp:267
    const float s_beta = (float)(-2.0 * temp_coef);                               // double scalar cast to the expression scalar (Eigen 3.3)
    const float s_gamma = -temp_coef;
    const float s_delta = (float)(2.0 * temp_coef);
    double B = 0, C = 0, D = 0, E = 0;
    const bool shuffled = (o->variant & ORC_VAR_SHUFFLE) != 0;
    std::vector<double> rowBCDE(shuffled ? (size_t)N * 4 : 0);
#pragma omp parallel num_threads(shuffled ? 1 : o->threads)
    {
        double lB = 0, lC = 0, lD = 0, lE = 0;
#pragma omp for schedule(static)
        for (int i = 0; i < N; ++i) {                                             // cvo.cpp:275-315
            const float* xi = &X.xyz[(size_t)i * 3];
            double Bi = 0, Ci = 0, Di = 0, Ei = 0;
            for (int e = o->A_rowptr[i]; e < o->A_rowptr[i + 1]; ++e) {
                const int idx = o->A_col[e];
                const float* y = &o->cloud_y[(size_t)idx * 3];
                const float df[3] = {xi[0] - y[0], xi[1] - y[1], xi[2] - y[2]};   // cvo.cpp:286
                const float* z1 = &xiz[(size_t)idx * 3]; const float* z2 = &xi2z[(size_t)idx * 3];
                const float* z3 = &xi3z[(size_t)idx * 3]; const float* z4 = &xi4z[(size_t)idx * 3];
                // (s*row)*diff : fixed inner size 3 -> t0+(t1+t2)
                const float beta = sum3_fixed((s_beta * z1[0]) * df[0], (s_beta * z1[1]) * df[1], (s_beta * z1[2]) * df[2]);          // cvo.cpp:288
                const float gamma = s_gamma * (nrm[idx] + sum3_fixed((2.f * z2[0]) * df[0], (2.f * z2[1]) * df[1], (2.f * z2[2]) * df[2]));   // cvo.cpp:290-291
                const float delta = s_delta * (mdot[idx] + sum3_fixed((-z3[0]) * df[0], (-z3[1]) * df[1], (-z3[2]) * df[2]));         // cvo.cpp:293-294
                const float epsil = s_gamma * (econst[idx] + sum3_fixed((2.f * z4[0]) * df[0], (2.f * z4[1]) * df[1], (2.f * z4[2]) * df[2])); // cvo.cpp:296-297
                const float A_ij = o->A_val[e];
                Bi += double(A_ij * beta);                                                                                  // cvo.cpp:301
                Ci += double(A_ij * (gamma + beta * beta / 2.0));                                                          // cvo.cpp:302
                Di += double(A_ij * (delta + beta * gamma + beta * beta * beta / 6.0));                                    // cvo.cpp:303
                Ei += double(A_ij * (epsil + beta * delta + 1 / 2.0 * beta * beta * gamma                                  // cvo.cpp:304-305
                                     + 1 / 2.0 * gamma * gamma + 1 / 24.0 * beta * beta * beta * beta));
            }
            if (shuffled) { rowBCDE[(size_t)i * 4 + 0] = Bi; rowBCDE[(size_t)i * 4 + 1] = Ci; rowBCDE[(size_t)i * 4 + 2] = Di; rowBCDE[(size_t)i * 4 + 3] = Ei; }
            else { lB += Bi; lC += Ci; lD += Di; lE += Ei; }
        }
#pragma omp critical
        { B += lB; C += lC; D += lD; E += lE; }
    }
    if (shuffled) {                                                               // cvo.cpp:309-314 in a seeded random row order
        std::vector<int> perm(N); std::iota(perm.begin(), perm.end(), 0);
        std::mt19937_64 rng(o->shuffle_seed + 0xD1B54A32D192ED03ull * (++o->shuffle_calls));
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int i : perm) { B += rowBCDE[(size_t)i * 4 + 0]; C += rowBCDE[(size_t)i * 4 + 1]; D += rowBCDE[(size_t)i * 4 + 2]; E += rowBCDE[(size_t)i * 4 + 3]; }
    }
    o->last_BCDE[0] = B; o->last_BCDE[1] = C; o->last_BCDE[2] = D; o->last_BCDE[3] = E;
    // p_coef << 4.0*float(E), 3.0*float(D), 2.0*float(C), float(B);   cvo.cpp:318
    const float c3 = (float)(4.0 * float(E)), c2 = (float)(3.0 * float(D)), c1 = (float)(2.0 * float(C)), c0 = float(B);
    o->step = (o->variant & ORC_VAR_F32_ROOTS) ? orc_cubic_step_f32eig(c3, c2, c1, c0, o->p.min_step) : orc_cubic_step(c3, c2, c1, c0, o->p.min_step);
}

}  // namespace

static inline float sin_cr(float x) { return (float)std::sin((double)x); }
static inline float cos_cr(float x) { return (float)std::cos((double)x); }
// Exp_SEK3 for K=1, LieGroup.cpp:159-186 (incl. the theta<1e-6 branch: R=I, Jl=I)
extern "C" void orc_exp_sek3(const float omega[3], const float v[3], float dt, float dR[9], float dT[3]) {
    const float TOLERANCE = 1e-6f;                                                // LieGroup.cpp:18
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float Jl[9];
    const float theta = norm3_fixed(omega);
    if (theta < TOLERANCE) {
        std::memcpy(dR, I, sizeof(I)); std::memcpy(Jl, I, sizeof(I));             // LieGroup.cpp:168-170
    } else {
        float A[9]; skew3(omega, A);
        const float theta2 = theta * theta;
        // sin(float), cos(float) are the libm's float routines in the reference (LieGroup.cpp:174-175) -- Intel's under icpc, glibc's under gcc -- and
        // differ between libms in the last bit: glibc 2.35's sinf is not the correctly rounded value for 1 % of the arguments above 0.03 (none below
        // 1e-3; tests/test_gpu_pair_values.py measures it), the device's OCML sinf for another 1.4 %.  Oracle and device both take the correctly
        // rounded float: the double routine's value rounded once (sin_cr / cos_cr here, sin_f32_cr / cos_f32_cr in cvo_math.hpp).
        const float stheta = sin_cr(dt * theta);
        const float ctheta = cos_cr(dt * theta);
        const float oneMinusCosTheta2 = (1 - ctheta) / (theta2);
        float A2[9]; mat3_mul(A, A, A2);
        const float s1 = stheta / theta;
        const float s3 = (dt * theta - stheta) / (theta2 * theta);
        for (int i = 0; i < 9; ++i) {
            dR[i] = (I[i] + s1 * A[i]) + oneMinusCosTheta2 * A2[i];               // LieGroup.cpp:178
            Jl[i] = (dt * I[i] + oneMinusCosTheta2 * A[i]) + s3 * A2[i];          // LieGroup.cpp:179
        }
    }
    mat3_vec(Jl, v, dT);                                                          // LieGroup.cpp:183
}

// dist_se3, cvo.cpp:94-104: || logm([dR dT; 0 1]) ||_F.  Eigen's Matrix4f::log() is
// restated in closed form: logm = [phi^ , V(phi)^-1 dT ; 0 0], so the Frobenius norm
// is sqrt(2*theta^2 + |V^-1 dT|^2) with V^-1 = LeftJacobianInverse_SO3
// (LieGroup.cpp:61-69).  theta is taken from atan2(|vee(dR-dR^T)/2|, (tr-1)/2),
// which (unlike acos of the trace) keeps the small angles that decide the stop test.
extern "C" float orc_dist_se3(const float dR[9], const float dT[3]) {
    const double w[3] = {0.5 * ((double)dR[7] - dR[5]), 0.5 * ((double)dR[2] - dR[6]), 0.5 * ((double)dR[3] - dR[1])};
    const double s = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double c = 0.5 * ((double)dR[0] + dR[4] + dR[8] - 1.0);
    const double theta = std::atan2(s, c);
    double phi[3] = {w[0], w[1], w[2]};
    if (s > 1e-300) { const double f = theta / s; for (int k = 0; k < 3; ++k) phi[k] *= f; }
    // V^-1 = I - 0.5*Phi + coef*Phi^2
    double coef;
    if (theta < 1e-4) coef = 1.0 / 12.0 + theta * theta / 720.0;
    else coef = 1.0 / (theta * theta) - (1.0 + std::cos(theta)) / (2.0 * theta * std::sin(theta));
    const double t[3] = {dT[0], dT[1], dT[2]};
    const double pxt[3] = {phi[1] * t[2] - phi[2] * t[1], phi[2] * t[0] - phi[0] * t[2], phi[0] * t[1] - phi[1] * t[0]};
    const double ppxt[3] = {phi[1] * pxt[2] - phi[2] * pxt[1], phi[2] * pxt[0] - phi[0] * pxt[2], phi[0] * pxt[1] - phi[1] * pxt[0]};
    double u2 = 0;
    for (int k = 0; k < 3; ++k) { const double u = t[k] - 0.5 * pxt[k] + coef * ppxt[k]; u2 += u * u; }
    return (float)std::sqrt(2.0 * theta * theta + u2);
}

// ---- reference-noise variants (cvo_oracle.h) -------------------------------------------------
// poly_solver + root selection as the reference runs them: f32 companion matrix, f32 eigenvalues, imag()==0 (cvo.cpp:76-92,324-330)
extern "C" float orc_cubic_step_f32eig(float c3, float c2, float c1, float c0, float min_step) {
    refnoise::Mat<float, 4> M;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M.a[i][j] = 0.f;
    M.a[1][0] = 1.f; M.a[2][1] = 1.f;                                             // bottomLeftCorner = Identity, cvo.cpp:82-83
    M.a[0][0] = -(c2 / c3); M.a[0][1] = -(c1 / c3); M.a[0][2] = -(c0 / c3);       // M.row(0) = -(coef/coef(0)).segment(1,order), cvo.cpp:86
    float re[3], im[3];
    float best = std::numeric_limits<float>::max();
    if (refnoise::eigenvalues<float, 4>(M, 3, re, im)) {
        for (int k = 0; k < 3; ++k) if (re[k] > 0 && re[k] < best && im[k] == 0) best = re[k];   // cvo.cpp:325-327
    }
    float step = (best == std::numeric_limits<float>::max()) ? min_step : best;   // cvo.cpp:330
    step = step > 0.8 ? (float)0.8 : step;                                        // cvo.cpp:333
    return step;
}
// dist_se3 as the reference runs it: Matrix4f::log().norm() in f32 (cvo.cpp:94-104)
extern "C" float orc_dist_se3_f32logm(const float dR[9], const float dT[3]) {
    refnoise::Mat<float, 4> M = refnoise::Mat<float, 4>::identity(), Lg;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M.a[r][c] = dR[r * 3 + c]; M.a[r][3] = dT[r]; }
    if (!refnoise::logm<float, 4>(M, 4, Lg)) return std::numeric_limits<float>::quiet_NaN();
    float s = 0.f;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) s += Lg.a[r][c] * Lg.a[r][c];
    return std::sqrt(s);
}
extern "C" void orc_set_variant(orc_cvo* o, int flags, unsigned long long shuffle_seed) { o->variant = flags; o->shuffle_seed = shuffle_seed; o->shuffle_calls = 0; }
template <class S> static int test_eig(int n, const double* A, double* re, double* im) {
    refnoise::Mat<S, 4> M = refnoise::Mat<S, 4>::identity();
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M.a[i][j] = (S)A[i * n + j];
    S r[4], q[4];
    if (!refnoise::eigenvalues<S, 4>(M, n, r, q)) return 1;
    for (int i = 0; i < n; ++i) { re[i] = (double)r[i]; im[i] = (double)q[i]; }
    return 0;
}
template <class S> static int test_logm(int n, const double* A, double* out) {
    refnoise::Mat<S, 4> M = refnoise::Mat<S, 4>::identity(), Lg;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M.a[i][j] = (S)A[i * n + j];
    if (!refnoise::logm<S, 4>(M, n, Lg)) return 1;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) out[i * n + j] = (double)Lg.a[i][j];
    return 0;
}
extern "C" int orc_test_eigenvalues(int n, const double* A, double* re, double* im, int use_f32) {
    if (n < 1 || n > 4) return 2;
    return use_f32 ? test_eig<float>(n, A, re, im) : test_eig<double>(n, A, re, im);
}
extern "C" int orc_test_logm(int n, const double* A, double* out, int use_f32) {
    if (n < 1 || n > 4) return 2;
    return use_f32 ? test_logm<float>(n, A, out) : test_logm<double>(n, A, out);
}

// align, cvo.cpp:763-821
extern "C" int orc_align(orc_cvo* o, orc_trace_row* trace, int trace_cap, int* trace_len) {
    if (trace_len) *trace_len = 0;
    if (!o->fixed || !o->moving || o->fixed->n <= 0 || o->moving->n <= 0) return 2;   // Q8: reference asserts / UB on empty clouds
    for (int k = 0; k < o->p.max_iter; ++k) {
        update_tf(o);                                                             // cvo.cpp:770
        transform_pcd(o);                                                         // cvo.cpp:773
        compute_flow(o);                                                          // cvo.cpp:776
        compute_step_size(o);                                                     // cvo.cpp:779
        orc_trace_row* tr = (trace && k < trace_cap) ? &trace[k] : nullptr;
        if (tr) {
            for (int q = 0; q < 3; ++q) { tr->omega[q] = o->omega[q]; tr->v[q] = o->v[q]; }
            tr->nnz = o->A_nonzero; tr->B = o->last_BCDE[0]; tr->C = o->last_BCDE[1]; tr->D = o->last_BCDE[2]; tr->E = o->last_BCDE[3];
            tr->step = o->step; tr->ell = o->ell; tr->dist = -1.f;
            if (trace_len) *trace_len = k + 1;
        }
        if (norm3_fixed(o->omega) < o->p.eps && norm3_fixed(o->v) < o->p.eps) {    // cvo.cpp:782
            o->iter = k; break;
        }
        float dR[9], dT[3];
        orc_exp_sek3(o->omega, o->v, o->step, dR, dT);                            // cvo.cpp:793
        float RdT[3]; mat3_vec(o->R, dT, RdT);
        for (int q = 0; q < 3; ++q) o->T[q] = RdT[q] + o->T[q];                    // cvo.cpp:800
        float Rn[9]; mat3_mul(o->R, dR, Rn); std::memcpy(o->R, Rn, sizeof(Rn));   // cvo.cpp:801
        const float dist = (o->variant & ORC_VAR_F32_LOGM) ? orc_dist_se3_f32logm(dR, dT) : orc_dist_se3(dR, dT);
        if (tr) tr->dist = dist;
        if (dist < o->p.eps_2) { o->iter = k; break; }                            // cvo.cpp:804-808
        o->ell = (k > 2) ? (float)0.10 : o->ell;                                  // cvo.cpp:810-812
        o->ell = (k > 9) ? (float)0.06 : o->ell;
        o->ell = (k > 19) ? (float)0.03 : o->ell;
    }
    o->prev_transform = o->transform;                                             // cvo.cpp:815
    o->accum_transform = aff_mul(o->accum_transform, o->transform);               // cvo.cpp:816
    update_tf(o);                                                                 // cvo.cpp:817
    o->cloud_y.clear();                                                           // cvo.cpp:820
    return 0;
}

// ---- adaptive-ell variant: acvo::align, adaptive_cvo.cpp:490-555 (see cvo_oracle.h)
extern "C" void orc_adaptive_default_params(orc_adaptive_params* p) {
    p->ell_init = 0.1; p->ell_min = 0.0391; p->ell_max = 0.15; p->dl_step = 0.3;                 // adaptive_cvo.cpp:27-32
    p->sigma = 0.1; p->sp_thres = 8.315e-3; p->c = 7.0; p->d = 7.0; p->c_ell = 0.5; p->c_sigma = 1;   // :35-42
    p->max_iter = 2000; p->min_step = 2 * 1.0e-1; p->eps = 5 * 1.0e-5; p->eps_2 = 1.0e-5;        // :44-47
}
extern "C" int orc_adaptive_align(const orc_adaptive_params* ap, const float* fixed_xyz, const float* fixed_feat, int n_fixed,
                                  const float* moving_xyz, const float* moving_feat, int n_moving, float R_inout[9], float T_inout[3],
                                  float* ell_out, float transform_out[12], int* iter, orc_adaptive_row* trace, int trace_cap, int* trace_len,
                                  int search_mode, int threads) {
    if (trace_len) *trace_len = 0;
    if (n_fixed <= 0 || n_moving <= 0) return 2;
    orc_params bp; orc_default_params(&bp);
    bp.ell = ap->ell_init; bp.sigma = ap->sigma; bp.sp_thres = ap->sp_thres; bp.c = ap->c; bp.d = ap->d; bp.c_ell = ap->c_ell; bp.c_sigma = ap->c_sigma;
    bp.max_iter = ap->max_iter; bp.min_step = ap->min_step; bp.eps = ap->eps; bp.eps_2 = ap->eps_2;
    std::unique_ptr<orc_cvo> holder(orc_create(&bp));
    orc_cvo* o = holder.get();
    orc_set_exec(o, search_mode, threads);
    orc_set_pcd(o, fixed_xyz, fixed_feat, n_fixed); orc_set_pcd(o, moving_xyz, moving_feat, n_moving);
    std::memcpy(o->R, R_inout, sizeof(o->R)); std::memcpy(o->T, T_inout, sizeof(o->T));
    o->ell = ap->ell_init;                                                        // set_pcd: ell = ell_init; ell_max = 0.15   adaptive_cvo.cpp:476-477
    float ell_max = ap->ell_max;
    const Cloud& X = *o->fixed; const Cloud& Y = *o->moving;
    const int N = X.n, M = Y.n;
    const float s2 = ap->sigma * ap->sigma;
    std::vector<int> xx_rp, xx_c, yy_rp, yy_c; std::vector<float> xx_v, yy_v;
    for (int k = 0; k < ap->max_iter; ++k) {
        update_tf(o);                                                             // adaptive_cvo.cpp:497
        transform_pcd(o);                                                         // :500
        // compute_flow, adaptive_cvo.cpp:154-272
        se_kernel(o, o->ell, s2);                                                                                   // Axy  :156
        se_kernel_clouds(o, X.xyz.data(), X, X.xyz.data(), X, o->ell, s2, xx_rp, xx_c, xx_v);                       // Axx  :159
        se_kernel_clouds(o, o->cloud_y.data(), Y, o->cloud_y.data(), Y, o->ell, s2, yy_rp, yy_c, yy_v);             // Ayy  :160
        const float inv_c = 1 / ap->c, inv_d = 1 / ap->d;
        const float ell_3 = o->ell * o->ell * o->ell;                                                                // :172
        const float inv_l3 = 1 / ell_3;
        double dw[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, dl = 0;
        auto d2_seq = [](const float* a, const float* b) { const float e0 = a[0] - b[0], e1 = a[1] - b[1], e2 = a[2] - b[2]; return (e0 * e0 + e1 * e1) + e2 * e2; };
        for (int i = 0; i < N; ++i) {                                             // :175-240 (row order fixed here; the reference: TBB, any order)
            const float* xi = &X.xyz[(size_t)i * 3];
            float sw[3] = {0, 0, 0}, sv[3] = {0, 0, 0}, s_yx = 0.f, s_xx = 0.f;
            for (int e = o->A_rowptr[i]; e < o->A_rowptr[i + 1]; ++e) {
                const float* yj = &o->cloud_y[(size_t)o->A_col[e] * 3];
                const float a = o->A_val[e];
                float cr[3]; cross3(xi, yj, cr);                                  // :203
                for (int q = 0; q < 3; ++q) { sw[q] += a * cr[q]; sv[q] += a * (yj[q] - xi[q]); }   // :227-228
                s_yx += (inv_l3 * a) * d2_seq(yj, xi);                            // (1/ell_3*Ai*sum_diff_yx_2)(0,0)   :231, :205
            }
            for (int e = xx_rp[i]; e < xx_rp[i + 1]; ++e) s_xx += (inv_l3 * xx_v[e]) * d2_seq(&X.xyz[(size_t)xx_c[e] * 3], xi);   // :209-215, :234
            double partial_dl = 0;
            // :216-226: for i < num_moving the Ayy row is walked but sum_diff_yy_2 is never filled -> it adds 0 (cvo_oracle.h)
            partial_dl -= double(2 * s_yx);                                       // :231
            partial_dl += double(s_xx);                                           // :234
            for (int q = 0; q < 3; ++q) { dw[q] += (double)(inv_c * sw[q]); dv[q] += (double)(inv_d * sv[q]); }   // :227-228, :237-238
            dl += partial_dl;                                                     // :239
        }
        for (int i = N; i < M; ++i) {                                             // :243-266: the rows of Ayy beyond num_fixed, here with the squared norms
            const float* yi = &o->cloud_y[(size_t)i * 3];
            float s_yy = 0.f;
            for (int e = yy_rp[i]; e < yy_rp[i + 1]; ++e) s_yy += (inv_l3 * yy_v[e]) * d2_seq(&o->cloud_y[(size_t)yy_c[e] * 3], yi);
            dl += double(s_yy);
        }
        for (int q = 0; q < 3; ++q) { o->omega[q] = (float)dw[q]; o->v[q] = (float)dv[q]; }                         // :269-270
        const int nnz_xy = o->A_rowptr[N], nnz_xx = xx_rp[N], nnz_yy = yy_rp[M];
        const double dlf = dl / (nnz_xx + nnz_yy - 2 * nnz_xy);                   // :271 (dl and dl_step are double members, adaptive_cvo.hpp:76-77)
        o->A_nonzero = nnz_xy;
        compute_step_size(o);                                                     // :506, adaptive_cvo.cpp:275-365 = the base sequence
        if (trace && k < trace_cap) {
            orc_adaptive_row& tr = trace[k];
            for (int q = 0; q < 3; ++q) { tr.omega[q] = o->omega[q]; tr.v[q] = o->v[q]; }
            tr.dl = (float)dlf; tr.ell = o->ell; tr.step = o->step; tr.nnz_xy = nnz_xy; tr.nnz_xx = nnz_xx; tr.nnz_yy = nnz_yy;
            if (trace_len) *trace_len = k + 1;
        }
        const double nw = std::sqrt((double)o->omega[0] * o->omega[0] + (double)o->omega[1] * o->omega[1] + (double)o->omega[2] * o->omega[2]);
        const double nv = std::sqrt((double)o->v[0] * o->v[0] + (double)o->v[1] * o->v[1] + (double)o->v[2] * o->v[2]);
        if (nw < ap->eps && nv < ap->eps) { o->iter = k; break; }                 // :509 (norms in double there)
        float dR[9], dT[3];
        orc_exp_sek3(o->omega, o->v, o->step, dR, dT);                            // :520
        float RdT[3]; mat3_vec(o->R, dT, RdT);
        for (int q = 0; q < 3; ++q) o->T[q] = RdT[q] + o->T[q];                    // :527
        float Rn[9]; mat3_mul(o->R, dR, Rn); std::memcpy(o->R, Rn, sizeof(Rn));   // :528
        if (orc_dist_se3(dR, dT) < ap->eps_2) { o->iter = k; break; }             // :531
        o->ell = (float)(o->ell + (double)ap->dl_step * dlf);                     // :538 (double expression stored into the float member)
        if (o->ell >= ell_max) { o->ell = (float)(ell_max * 0.7); ell_max = (float)(ell_max * 0.7); }   // :541-544
        o->ell = (o->ell < ap->ell_min) ? ap->ell_min : o->ell;                   // :545
    }
    update_tf(o);                                                                 // :550
    std::memcpy(R_inout, o->R, sizeof(o->R)); std::memcpy(T_inout, o->T, sizeof(o->T));
    if (ell_out) *ell_out = o->ell;
    if (transform_out) std::memcpy(transform_out, o->transform.m, sizeof(float) * 12);
    if (iter) *iter = o->iter;
    return 0;
}

// function_inner_product, cvo.cpp:388-459.  `a_xyz` are the (possibly transformed)
// positions of cloud a; features come from the untransformed cloud.
static orc_inn_p fip_impl(orc_cvo* o, const std::vector<float>& a_xyz, const Cloud& a, const Cloud& b) {
    const orc_params& P = o->p;
    const float ell = o->ell, sigma = P.sigma;
    const float d2_thres = (float)(-2.0 * ell * ell * (double)log_cr(P.sp_thres / sigma / sigma));            // cvo.cpp:395
    const float d2_c_thres = (float)(-2.0 * P.c_ell * P.c_ell * (double)log_cr(P.sp_thres / P.c_sigma / P.c_sigma));   // cvo.cpp:396
    KdTree tree; const KdTree* tp = nullptr;
    if (o->search_mode == ORC_SEARCH_KDTREE) { tree.build(b.xyz.data(), b.n); tp = &tree; }
    const int feat_order = (o->variant & ORC_VAR_FEAT_HADD) ? 1 : ((o->variant & ORC_VAR_FEAT_MOVEHL) ? 2 : 0);
    double sum_A = 0, sum = 0, sum_e = 0;
#pragma omp parallel num_threads(o->threads)
    {
        std::vector<Match> ms; double lA = 0, ls = 0;
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < a.n; ++i) {
            radius_matches(b.xyz.data(), b.n, tp, &a_xyz[(size_t)i * 3], d2_thres, false, ms);
            for (const Match& mt : ms) {
                const float d2 = mt.d2;
                if (d2 < d2_thres) {
                    const float d2_color = feat_d2(a.feat.data(), a.n, i, b.feat.data(), b.n, mt.j, feat_order);
                    if (d2_color < d2_c_thres) {
                        const float k = (float)(sigma * sigma * std::exp(-d2 / (2.0 * ell * ell)));              // cvo.cpp:429
                        const float ck = (float)(P.c_sigma * P.c_sigma * std::exp(-d2_color / (2.0 * P.c_ell * P.c_ell)));
                        const float av = ck * k;
                        lA += av; ls += 1;                                        // cvo.cpp:432-435 (no a>sp_thres test: Q6)
                    }
                }
            }
        }
#pragma omp critical
        { sum_A += lA; sum += ls; }
    }
    if (sum == 0) sum = 1;                                                        // cvo.cpp:455-456
    orc_inn_p r; r.value = (float)sum_A; r.num = (int)sum; r.num_e = (int)sum_e;
    return r;
}

// eigenvalues of a symmetric 6x6 (cyclic Jacobi, double)
static void sym_eig6(const double* Hin, double* ev) {
    double A[36]; std::memcpy(A, Hin, sizeof(A));
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0; for (int i = 0; i < 6; ++i) for (int j = i + 1; j < 6; ++j) off += A[i * 6 + j] * A[i * 6 + j];
        if (off < 1e-300) break;
        for (int p = 0; p < 6; ++p) for (int q = p + 1; q < 6; ++q) {
            if (A[p * 6 + q] == 0.0) continue;
            const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.0 * A[p * 6 + q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 6; ++k) { const double akp = A[k * 6 + p], akq = A[k * 6 + q]; A[k * 6 + p] = c * akp - s * akq; A[k * 6 + q] = s * akp + c * akq; }
            for (int k = 0; k < 6; ++k) { const double apk = A[p * 6 + k], aqk = A[q * 6 + k]; A[p * 6 + k] = c * apk - s * aqk; A[q * 6 + k] = s * apk + c * aqk; }
        }
    }
    for (int i = 0; i < 6; ++i) ev[i] = A[i * 6 + i];
}

// cvo.cpp:726-758: Hessian *= -1e-5; shift by sum(1 - lambda_min|.|) until the
// eigenvalue of smallest magnitude has |lambda| >= 1; identity if no inliers.
extern "C" void orc_hessian_regularize(const float Hin[36], int inliers, double Hout[36]) {
    float H[36];
    if (inliers) {
        const float scale = (float)(-1.0 / 100000);
        for (int i = 0; i < 36; ++i) H[i] = Hin[i] * scale;
        double Hd[36], evd[6]; for (int i = 0; i < 36; ++i) Hd[i] = H[i];
        sym_eig6(Hd, evd);
        float ev[6]; for (int i = 0; i < 6; ++i) ev[i] = (float)evd[i];
        auto min_abs = [&]() { int mi = 0; for (int i = 1; i < 6; ++i) if (std::fabs(ev[i]) < std::fabs(ev[mi])) mi = i; return ev[mi]; };
        float sufficient_scale = 0.0f;
        float min_eigen = min_abs();
        int guard = 0;
        while (std::fabs(min_eigen) < 1.0 && guard++ < 1000) {
            sufficient_scale += (float)(1.0 - min_eigen);
            for (int i = 0; i < 6; ++i) ev[i] += (float)(1.0 - min_eigen) * 1.0f;
            min_eigen = min_abs();
        }
        for (int i = 0; i < 6; ++i) H[i * 6 + i] += sufficient_scale * 1.0f;
    } else {
        for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1.f : 0.f;
    }
    for (int i = 0; i < 36; ++i) Hout[i] = (double)H[i];
}

// se3_Hessian, cvo.cpp:620-759.  H_raw_f64 (optional) returns the un-regularised sum
// accumulated in double (an exact reference for the f32 sum the reference keeps).
static void hessian_impl(orc_cvo* o, const std::vector<float>& a_xyz, const Cloud& a, const Cloud& b,
                         double Hout[36], int* inliers_out, double* H_raw_f64) {
    const orc_params& P = o->p;
    const float ell = o->ell, sigma = P.sigma;
    const float d2_thres = (float)(-2.0 * ell * ell * (double)log_cr(P.sp_thres / sigma / sigma));
    const float d2_c_thres = (float)(-2.0 * P.c_ell * P.c_ell * (double)log_cr(P.sp_thres / P.c_sigma / P.c_sigma));
    KdTree tree; const KdTree* tp = nullptr;
    if (o->search_mode == ORC_SEARCH_KDTREE) { tree.build(b.xyz.data(), b.n); tp = &tree; }
    const int feat_order = (o->variant & ORC_VAR_FEAT_HADD) ? 1 : ((o->variant & ORC_VAR_FEAT_MOVEHL) ? 2 : 0);
    float H[36]; std::memset(H, 0, sizeof(H));
    double Hd[36]; std::memset(Hd, 0, sizeof(Hd));
    int inliers = *inliers_out;                       // the reference accumulates into the caller's variable
    const float il2 = 1 / (ell * ell);
    std::vector<Match> ms;
    for (int i = 0; i < a.n; ++i) {                   // serial: the f32 sum is order dependent, keep one order
        const float* pa = &a_xyz[(size_t)i * 3];
        radius_matches(b.xyz.data(), b.n, tp, pa, d2_thres, false, ms);
        for (const Match& mt : ms) {
            const float d2 = mt.d2;
            if (!(d2 < d2_thres)) continue;
            const float d2_color = feat_d2(a.feat.data(), a.n, i, b.feat.data(), b.n, mt.j, feat_order);
            if (!(d2_color < d2_c_thres)) continue;
            const float* pb = &b.xyz[(size_t)mt.j * 3];
            const float k = (float)(sigma * sigma * std::exp(-d2 / (2.0 * ell * ell)));                          // cvo.cpp:661
            const float cdot = feat_dot(a.feat.data(), a.n, i, b.feat.data(), b.n, mt.j, feat_order);                      // cvo.cpp:662
            float cr[3]; cross3(pa, pb, cr);
            float Bk[36];
            const float dot1 = pa[1] * pb[1] + pa[2] * pb[2], dot2 = pa[0] * pb[0] + pa[2] * pb[2], dot3 = pa[0] * pb[0] + pa[1] * pb[1];
            float A3[9], C3[9], D3[9];
            A3[0] = il2 * cr[0] * cr[0] - dot1;  A3[4] = il2 * cr[1] * cr[1] - dot2;  A3[8] = il2 * cr[2] * cr[2] - dot3;   // cvo.cpp:670-672
            A3[1] = A3[3] = (float)(il2 * cr[0] * cr[1] + 0.5 * (pa[0] * pb[1] + pa[1] * pb[0]));                         // cvo.cpp:673
            A3[2] = A3[6] = (float)(il2 * cr[0] * cr[2] + 0.5 * (pa[0] * pb[2] + pa[2] * pb[0]));
            A3[5] = A3[7] = (float)(il2 * cr[1] * cr[2] + 0.5 * (pa[1] * pb[2] + pa[2] * pb[1]));
            const float db[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};                                          // cvo.cpp:679
            C3[0] = il2 * cr[0] * db[0];  C3[4] = il2 * cr[1] * db[1];  C3[8] = il2 * cr[2] * db[2];                    // cvo.cpp:680-682
            C3[3] = pa[2] + il2 * db[1] * cr[0];   /* C(1,0) */  C3[6] = -pa[1] + il2 * db[2] * cr[0];  /* C(2,0) */
            C3[1] = -pa[2] + il2 * db[0] * cr[1];  /* C(0,1) */  C3[7] = pa[0] + il2 * db[2] * cr[1];   /* C(2,1) */
            C3[2] = pa[1] + il2 * db[0] * cr[2];   /* C(0,2) */  C3[5] = -pa[0] + il2 * db[1] * cr[2];  /* C(1,2) */
            D3[0] = il2 * db[0] * db[0] - 1;  D3[4] = il2 * db[1] * db[1] - 1;  D3[8] = il2 * db[2] * db[2] - 1;         // cvo.cpp:692-694
            D3[1] = D3[3] = il2 * db[0] * db[1];  D3[2] = D3[6] = il2 * db[0] * db[2];  D3[5] = D3[7] = il2 * db[1] * db[2];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {                                                   // cvo.cpp:700-704
                Bk[r * 6 + c] = A3[r * 3 + c];
                Bk[r * 6 + 3 + c] = C3[c * 3 + r];            // C^T
                Bk[(3 + r) * 6 + c] = C3[r * 3 + c];
                Bk[(3 + r) * 6 + 3 + c] = D3[r * 3 + c];
            }
            const float w = il2 * cdot * k;                                                                             // cvo.cpp:707
            for (int q = 0; q < 36; ++q) { H[q] += w * Bk[q]; Hd[q] += (double)w * (double)Bk[q]; }
            inliers++;
        }
    }
    *inliers_out = inliers;
    if (H_raw_f64) std::memcpy(H_raw_f64, Hd, sizeof(Hd));
    orc_hessian_regularize(H, inliers, Hout);
}

static const Cloud* slot_cloud(const orc_cvo* o, int slot) {
    switch (slot) { case ORC_SLOT_FIXED: return o->fixed.get(); case ORC_SLOT_MOVING: return o->moving.get(); case ORC_SLOT_PREVIOUS: return o->previous.get(); }
    return nullptr;
}
static std::vector<float> transformed_xyz(const Cloud& c, const float* tran) {
    std::vector<float> out(c.xyz);
    if (tran) for (int i = 0; i < c.n; ++i) aff_apply(tran, &c.xyz[(size_t)i * 3], &out[(size_t)i * 3]);   // cvo.cpp:485-487
    return out;
}

// =============================================================================
extern "C" {

void orc_default_params(orc_params* p) {
    p->ell = 0.15; p->sigma = 0.1; p->sp_thres = 8e-3; p->c = 7.0; p->d = 7.0; p->c_ell = 200; p->c_sigma = 1;
    p->max_iter = 2000; p->min_step = 2 * 1.0e-1; p->eps = 5 * 1.0e-5; p->eps_2 = 1.0e-5;
}

orc_cvo* orc_create(const orc_params* p) {
    orc_cvo* o = new orc_cvo();
    if (p) o->p = *p; else orc_default_params(&o->p);
    o->ell = o->p.ell;
    o->fixed.reset(new Cloud());                                                  // ptr_fixed_pcd(new point_cloud), cvo.cpp:30
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(o->R, I, sizeof(I)); o->T[0] = o->T[1] = o->T[2] = 0;              // cvo.cpp:66-67
    o->transform = o->prev_transform = o->accum_transform = aff_identity();       // cvo.cpp:68-70
    return o;
}
void orc_destroy(orc_cvo* o) { delete o; }
void orc_set_exec(orc_cvo* o, int search_mode, int threads) { o->search_mode = search_mode; o->threads = threads < 1 ? 1 : threads; }

int orc_set_pcd(orc_cvo* o, const float* xyz, const float* feat, int n) {
    auto fill = [&](Cloud& c) { c.n = n; c.xyz.assign(xyz, xyz + (size_t)n * 3); c.feat.assign(feat, feat + (size_t)n * 5); };
    if (!o->init) {                                                               // cvo.cpp:352-360
        if (!o->fixed) o->fixed.reset(new Cloud());
        fill(*o->fixed); o->init = true; return 0;
    }
    o->moving.reset(new Cloud()); fill(*o->moving);                               // cvo.cpp:362-366
    o->num_fixed = o->fixed ? o->fixed->n : 0; o->num_moving = o->moving->n;      // cvo.cpp:370-371
    o->A_nonzero = 0;                                                             // cvo.cpp:385
    return 0;
}

int orc_match(orc_cvo* o, const float* xyz, const float* feat, int n, double transform_out[12]) {
    if (!o->init) return 1;                                                       // "cvo not initialized !", cvo.cpp:463-466
    orc_set_pcd(o, xyz, feat, n);
    int rc = orc_align(o, nullptr, 0, nullptr);
    if (rc) return rc;
    for (int i = 0; i < 12; ++i) transform_out[i] = (double)o->transform.m[i];    // cvo.cpp:472
    return 0;
}

void orc_update_fixed_pcd(orc_cvo* o) { o->fixed = std::move(o->moving); }
void orc_update_previous_pcd(orc_cvo* o) { o->previous = std::move(o->moving); o->pre_pc_init = true; }
void orc_reset_transform(orc_cvo* o, const float odom[12]) { std::memcpy(o->transform.m, odom, sizeof(float) * 12); }
void orc_reset_keyframe(orc_cvo* o, const float odom[12]) {
    if (!o->pre_pc_init) { o->fixed = std::move(o->moving); }
    else { o->fixed = std::move(o->previous); orc_update_previous_pcd(o); }
    orc_reset_transform(o, odom);
}
void orc_reset_initial(orc_cvo* o, const float odom[12], float out[12]) {
    Aff od; std::memcpy(od.m, odom, sizeof(od.m));
    Aff init = aff_inverse(aff_mul(o->transform, od));                            // cvo.cpp:613
    float L[9]; aff_linear(init, L);
    polar_rotation(L, o->R);                                                      // cvo.cpp:614
    o->T[0] = init.m[3]; o->T[1] = init.m[7]; o->T[2] = init.m[11];                // cvo.cpp:615
    Aff back = aff_inverse(init);                                                 // cvo.cpp:617
    std::memcpy(out, back.m, sizeof(back.m));
}

int orc_function_inner_product(orc_cvo* o, int slot_a, const float* tran_a, int slot_b, orc_inn_p* out) {
    const Cloud* a = slot_cloud(o, slot_a); const Cloud* b = slot_cloud(o, slot_b);
    if (!a || !b || a->n <= 0 || b->n <= 0) return 2;
    *out = fip_impl(o, transformed_xyz(*a, tran_a), *a, *b);
    return 0;
}
int orc_se3_hessian(orc_cvo* o, int slot_a, const float* tran_a, int slot_b, double H[36], int* inliers, double H_raw_f64[36]) {
    const Cloud* a = slot_cloud(o, slot_a); const Cloud* b = slot_cloud(o, slot_b);
    if (!a || !b || a->n <= 0 || b->n <= 0) return 2;
    hessian_impl(o, transformed_xyz(*a, tran_a), *a, *b, H, inliers, H_raw_f64);
    return 0;
}

int orc_compute_innerproduct(orc_cvo* o, orc_inn_p* pre, orc_inn_p* post, double H[36], const float tran[12], int* inliers,
                             orc_inn_p* inn_fixed, orc_inn_p* inn_moving, float* cos_angle) {
    if (!o->moving || !o->fixed || o->moving->n <= 0 || o->fixed->n <= 0) return 2;
    const Cloud& m = *o->moving; const Cloud& f = *o->fixed;
    std::vector<float> tx = transformed_xyz(m, tran);
    *pre = fip_impl(o, m.xyz, m, f);                                              // cvo.cpp:489
    *post = fip_impl(o, tx, m, f);                                                // cvo.cpp:491
    *inn_fixed = fip_impl(o, f.xyz, f, f);                                        // cvo.cpp:496
    *inn_moving = fip_impl(o, m.xyz, m, m);                                       // cvo.cpp:497
    *cos_angle = post->value / (std::sqrt(inn_fixed->value) * std::sqrt(inn_moving->value));   // cvo.cpp:498
    hessian_impl(o, tx, m, f, H, inliers, nullptr);                               // cvo.cpp:500
    return 0;
}

int orc_compute_innerproduct_lc(orc_cvo* o, orc_inn_p* prior, orc_inn_p* lc_prior, orc_inn_p* lc_pre, orc_inn_p* lc_post, double H[36],
                                const float prior_tran[12], const float lc_prior_tran[12], const float lc_prior_tran_2[12],
                                const float lc_tran[12], int* inliers_svd, int* inliers_pnp, orc_inn_p* inn_fixed,
                                orc_inn_p* inn_moving, float* cos_angle) {
    if (!o->moving || !o->fixed || o->moving->n <= 0 || o->fixed->n <= 0) return 2;
    const Cloud& m = *o->moving; const Cloud& f = *o->fixed;
    std::vector<float> lc = transformed_xyz(m, lc_tran), pr = transformed_xyz(m, prior_tran),
                       lp = transformed_xyz(m, lc_prior_tran), lp2 = transformed_xyz(m, lc_prior_tran_2);
    *prior = fip_impl(o, pr, m, f);                                               // cvo.cpp:539
    *lc_prior = fip_impl(o, lp, m, f);                                            // cvo.cpp:541
    *lc_pre = fip_impl(o, m.xyz, m, f);                                           // cvo.cpp:543
    *lc_post = fip_impl(o, lc, m, f);                                             // cvo.cpp:545
    *inn_fixed = fip_impl(o, f.xyz, f, f);                                        // cvo.cpp:550
    *inn_moving = fip_impl(o, m.xyz, m, m);                                       // cvo.cpp:551
    *cos_angle = lc_post->value / (std::sqrt(inn_fixed->value) * std::sqrt(inn_moving->value));   // cvo.cpp:552
    *inliers_svd = 0; hessian_impl(o, lc, m, f, H, inliers_svd, nullptr);         // cvo.cpp:554-555
    double Hdummy[36];
    *inliers_pnp = 0; hessian_impl(o, lp2, m, f, Hdummy, inliers_pnp, nullptr);   // cvo.cpp:557-558
    return 0;
}

void orc_get_state(const orc_cvo* o, float R[9], float T[3], float* ell, float transform[12], int* iter, int* A_nonzero,
                   int* num_fixed, int* num_moving) {
    if (R) std::memcpy(R, o->R, sizeof(o->R));
    if (T) std::memcpy(T, o->T, sizeof(o->T));
    if (ell) *ell = o->ell;
    if (transform) std::memcpy(transform, o->transform.m, sizeof(o->transform.m));
    if (iter) *iter = o->iter;
    if (A_nonzero) *A_nonzero = o->A_nonzero;
    if (num_fixed) *num_fixed = o->num_fixed;
    if (num_moving) *num_moving = o->num_moving;
}
void orc_set_state(orc_cvo* o, const float R[9], const float T[3], float ell) {
    if (R) std::memcpy(o->R, R, sizeof(o->R));
    if (T) std::memcpy(o->T, T, sizeof(o->T));
    o->ell = ell;
}
void orc_get_accum(const orc_cvo* o, float prev_transform[12], float accum_transform[12]) {
    if (prev_transform) std::memcpy(prev_transform, o->prev_transform.m, sizeof(float) * 12);
    if (accum_transform) std::memcpy(accum_transform, o->accum_transform.m, sizeof(float) * 12);
}
int orc_get_init(const orc_cvo* o) { return o->init ? 1 : 0; }
void orc_get_timing(const orc_cvo* o, double seconds[4]) { for (int i = 0; i < 4; ++i) seconds[i] = o->t_sec[i]; }

int orc_flow_once(orc_cvo* o, float omega[3], float v[3], int* nnz, double BCDE[4], float* step,
                  int* csr_rowptr, int* csr_col, float* csr_val, int csr_cap) {
    if (!o->fixed || !o->moving || o->fixed->n <= 0 || o->moving->n <= 0) return 2;
    update_tf(o); transform_pcd(o); compute_flow(o); compute_step_size(o);
    for (int k = 0; k < 3; ++k) { omega[k] = o->omega[k]; v[k] = o->v[k]; }
    *nnz = o->A_nonzero; std::memcpy(BCDE, o->last_BCDE, sizeof(o->last_BCDE)); *step = o->step;
    if (csr_rowptr) std::memcpy(csr_rowptr, o->A_rowptr.data(), sizeof(int) * o->A_rowptr.size());
    if (csr_col && csr_val) {
        int n = std::min<int>(csr_cap, (int)o->A_col.size());
        std::memcpy(csr_col, o->A_col.data(), sizeof(int) * n); std::memcpy(csr_val, o->A_val.data(), sizeof(float) * n);
    }
    return 0;
}

int orc_radius_search(const float* cloud_xyz, int n, const float* query, float radius_sq, int* out_idx, float* out_d2, int cap, int use_kdtree) {
    KdTree tree; const KdTree* tp = nullptr;
    if (use_kdtree) { tree.build(cloud_xyz, n); tp = &tree; }
    std::vector<Match> ms;
    radius_matches(cloud_xyz, n, tp, query, radius_sq, false, ms);
    int cnt = std::min<int>(cap, (int)ms.size());
    for (int k = 0; k < cnt; ++k) { out_idx[k] = ms[k].j; out_d2[k] = ms[k].d2; }
    return (int)ms.size();
}

}  // extern "C"
