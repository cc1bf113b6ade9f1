// cvo_kernels.hip -- gfx950 kernels of the CVO alignment hot path.
//
// cvo_align_kernel: the whole of cvo::align() (thirdparty/cvo/src/cvo.cpp:763-821)
// for a batch of independent frame pairs in ONE persistent launch.  G workgroups
// cooperate on a pair (each owns a contiguous block of fixed-cloud rows) and stay
// resident for all of its iterations; R, T, ell never leave the device.
//
// One iteration (cvo.cpp:768-813):
//   T   transform_pcd (cvo.cpp:336-341): y_j = M p_j into ybuf (HBM/L2 resident)
//   S   [only when the candidate lists are stale]  dense O(N*M) cull: the moving
//       cloud streams through LDS tiles (SoA, ds_read_b128 broadcasts, 4 columns per
//       read), rows live in registers (RPT per lane), 3 sub + 1 mul + 2 fma per
//       pair and one v_min3-folded compare per 4*RPT pairs; hits are appended to the
//       row's staging list in ascending column order (= the CSR order of
//       Eigen::setFromTriplets, cvo.cpp:182), then compacted into flat arrays.
//       The lists are built with radius (1+skin)*r and stay valid until the rigid
//       motion since the build can have moved any point by skin*r (checked every
//       iteration from the two transforms and max|p|), or ell changes.
//   C1  one lane per candidate: the reference's own pair arithmetic (cvo.cpp:166-175:
//       un-fused f32 d2, colour gate, double exp, a > sp_thres) -> a, a*(x cross y), a*(y-x)
//   C2  one lane per row: f32 sums in column order (cvo.cpp:213-223), f64 across
//       rows (cvo.cpp:226-230) by wave shuffles + LDS, exchanged between the pair's
//       workgroups as tagged 8-byte granules
//   L   one lane per survivor: beta..epsil and the B..E terms (cvo.cpp:282-306), f64
//   E   one lane: cubic, stop tests, Exp_SEK3, pose update, ell schedule
//       (cvo.cpp:317-333, 782-812)
// The cull uses fused arithmetic and a widened radius (a superset of the reference's
// neighbourhood); membership in A is decided in C1 by the reference's own
// expression, so the sparse set, every kernel value and every per-row sum follow
// the oracle's float sequence regardless of when the lists were built.
//
// MFMA is deliberately not used: S is a distance test + compare, C1/L are
// exp-heavy survivor work; neither is a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvo_device.h"
#include "cvo_math.hpp"

namespace cvohip {

#ifndef CVO_WAVES_PER_SIMD
#define CVO_WAVES_PER_SIMD 2      // 2: 256 VGPRs, one 512-thread workgroup per CU (measured faster); 4: 128 VGPRs, two per CU
#endif
constexpr int MAX_WAVES = 8;        // workgroups are at most 512 threads: 256 VGPRs per lane, no spills in the survivor phases
constexpr float FAR_ROW = 3.0e18f;    // coordinates of padding rows / columns: d2 overflows, never < threshold
constexpr float FAR_COL = -3.0e18f;

struct __attribute__((aligned(16))) Shared {
    double vals[8];
    double red[MAX_WAVES * 8];
    float R[9];
    float T[3];
    float ell;
    float step;
    float M[12];
    float Rb;              // radius they were built with
    float ell_build;
    float fred[MAX_WAVES];
    int wsum[MAX_WAVES];
    int wcnt[MAX_WAVES];   // survivors each wave compacted in C1
    int list_valid;
    int dense_mode;        // candidates did not fit the lists: per-row dense fallback until the next rebuild
    int total;             // candidates in this workgroup's flat list
    int stop;
    int status;
    int iter_at_break;
    int nnz;
    int cand;
    int rebuilds;
    int dense_fallbacks;
};

// Pointers read out of a PairDesc are generic to the compiler, which then emits FLAT loads/stores
// (no counted waits, everything drains at each use).  They all point into hipMalloc'ed memory,
// so the kernel re-types them as global (address space 1) once per pair.
#define CVO_GLOBAL __attribute__((address_space(1)))
typedef CVO_GLOBAL float gfloat;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef CVO_GLOBAL v4f gv4f;
// float4 array in global memory (HIP's float4 is a class whose copy operations only take generic pointers)
struct GF4 {
    gv4f* p;
    __device__ __forceinline__ float4 operator[](size_t i) const { const v4f t = p[i]; return make_float4(t.x, t.y, t.z, t.w); }
    __device__ __forceinline__ void set(size_t i, const float4 v) const { v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; p[i] = t; }
};
typedef CVO_GLOBAL uint16_t gu16;
typedef CVO_GLOBAL uint32_t gu32;
typedef CVO_GLOBAL int gint;
typedef CVO_GLOBAL unsigned long long gu64;
__device__ __forceinline__ float4 ld4(const gfloat* p) { const v4f t = *reinterpret_cast<const gv4f*>(p); return make_float4(t.x, t.y, t.z, t.w); }

// ---------------------------------------------------------------- reductions
// butterfly inside the wave (every lane ends with the wave total), one LDS slot
// per wave, then lanes 0..K-1 of wave 0 add the waves in order: deterministic.
template <int K>
__device__ __forceinline__ void block_reduce(double (&v)[K], Shared* sh, int tid, int nwaves) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh->red[wave * 8 + k] = v[k];
    }
    __syncthreads();
    if (tid < K) {
        double s = 0;
        for (int w = 0; w < nwaves; ++w) s += sh->red[w * 8 + tid];
        sh->vals[tid] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ float block_max(float v, Shared* sh, int tid, int nwaves) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((tid & 63) == 0) sh->fred[tid >> 6] = v;
    __syncthreads();
    float m = sh->fred[0];
    for (int w = 1; w < nwaves; ++w) m = fmaxf(m, sh->fred[w]);
    __syncthreads();
    return m;
}

// exclusive prefix sum of one int per thread over the workgroup; total returned to every thread
__device__ __forceinline__ int block_exclusive_scan(int v, Shared* sh, int tid, int nwaves, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    if (lane == 63) sh->wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) { const int s = sh->wsum[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// G workgroups of one pair swap K doubles: each publishes its partials as 2K
// 8-byte {tag = epoch, 32 payload bits} granules (one relaxed agent-scope store
// each: the data is its own flag, no fence), then wave 0 polls every
// workgroup's granules and adds them in workgroup order, so all G workgroups end
// with bit-identical totals and take identical branch decisions.  Two buffers
// alternate by epoch parity: a workgroup can run at most one phase ahead of the
// slowest member, so a buffer is never rewritten while someone still reads it.
// Called by wave 0 (all 64 lanes).  Returns false on timeout.
template <int K>
__device__ __forceinline__ bool group_exchange(Shared* sh, gu64* xch, int G, int g, unsigned epoch, int lane) {
    gu64* buf = xch + (size_t)(epoch & 1u) * G * XCH_WORDS;
    if (lane < 2 * K) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sh->vals[lane >> 1]);
        const unsigned pay = (lane & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(&buf[g * XCH_WORDS + lane], ((unsigned long long)epoch << 32) | pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double tot = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    for (int gg = 0; gg < G; ++gg) {
        unsigned long long x = 0;
        for (;;) {
            bool ok = true;
            if (lane < 2 * K) {
                x = __hip_atomic_load(&buf[gg * XCH_WORDS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(x >> 32) == epoch;
            }
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t_start > 300000000ull) return false;   // 3 s at 100 MHz
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned pay = (unsigned)x;
        const unsigned lo = __shfl(pay, (2 * lane) & 63, 64), hi = __shfl(pay, (2 * lane + 1) & 63, 64);
        if (lane < K) tot += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    if (lane < K) sh->vals[lane] = tot;
    return true;
}

// ---------------------------------------------------------------- exact pair arithmetic
struct Gates {
    float d2_thres, d2c_thres, sp;
    double den_l, den_c;      // 2.0*l*l, 2.0*c_ell*c_ell
    float s2, csig2;
    double inv_den_l, inv_den_c;
    float q_lim, q_il, q_ic;  // conservative f32 pre-test of a > sp before the double exps
};

// exp(x) in double for the only arguments the survivor path produces: the pre-test
// bounds both exponents by q_lim (~0.22 with the reference's constants), so no range
// reduction is needed: degree-13 Taylor/Horner with fma, truncation < 1e-19, rounding
// ~1 ulp like libm's exp; anything outside [-0.25, 0] takes the library routine.  The
// value is rounded to f32 right after (cvo.cpp:172-173), where a 1-ulp double
// difference is invisible except on ~1e-8 of inputs.
__device__ __forceinline__ double exp_small(double x) {
    if (!(x >= -0.25 && x <= 0.0)) return exp(x);
    double p = 1.0 / 6227020800.0;                                  // 1/13!
    p = __builtin_fma(p, x, 1.0 / 479001600.0);
    p = __builtin_fma(p, x, 1.0 / 39916800.0);
    p = __builtin_fma(p, x, 1.0 / 3628800.0);
    p = __builtin_fma(p, x, 1.0 / 362880.0);
    p = __builtin_fma(p, x, 1.0 / 40320.0);
    p = __builtin_fma(p, x, 1.0 / 5040.0);
    p = __builtin_fma(p, x, 1.0 / 720.0);
    p = __builtin_fma(p, x, 1.0 / 120.0);
    p = __builtin_fma(p, x, 1.0 / 24.0);
    p = __builtin_fma(p, x, 1.0 / 6.0);
    p = __builtin_fma(p, x, 0.5);
    p = __builtin_fma(p, x, 1.0);
    p = __builtin_fma(p, x, 1.0);
    return p;
}

__device__ __forceinline__ float feat_d2(const float* fa, const float* fb) {   // fixed-size 5 reduction (t0+t1)+(t2+(t3+t4))
    float t[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) { const float e = fa[c] - fb[c]; t[c] = e * e; }
    return (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
}

// cvo.cpp:166-175.  Returns a (> sp_thres) for a member of A, 0 otherwise.
__device__ __forceinline__ float se_kernel_value(const float* xi, const float* fi, const float4 yj, const float4 gj, const Gates& G) {
    // nanoflann L2 tail loop (nanoflann.hpp:403-406): result += diff*diff, three times
    const float e0 = xi[0] - yj.x, e1 = xi[1] - yj.y, e2 = xi[2] - yj.z;
    float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;
    if (!(d2 < G.d2_thres)) return 0.f;
    const float fb[5] = {yj.w, gj.x, gj.y, gj.z, gj.w};
    const float d2c = feat_d2(fi, fb);
    if (!(d2c < G.d2c_thres)) return 0.f;
    if (d2 * G.q_il + d2c * G.q_ic > G.q_lim) return 0.f;          // far below sp_thres: skip the exps
    // k = s2*exp(-d2/(2.0*l*l)), ck = c_sigma^2*exp(-d2c/(2.0*c_ell*c_ell)) evaluated in double, stored f32
    // (cvo.cpp:172-173); the division is a multiplication by the double reciprocal (<= 1 ulp of the argument)
    const float k = (float)((double)G.s2 * exp_small((double)(-d2) * G.inv_den_l));
    const float ck = (float)((double)G.csig2 * exp_small((double)(-d2c) * G.inv_den_c));
    const float a = ck * k;
    return a > G.sp ? a : 0.f;
}

__device__ __forceinline__ Gates make_gates(float l, const DevParams& P) {
    Gates G;
    G.s2 = P.sigma * P.sigma;                                       // se_kernel(ell, sigma*sigma), cvo.cpp:189
    G.csig2 = P.c_sigma * P.c_sigma;
    G.sp = P.sp_thres;
    G.d2_thres = gate_d2_align(l, P.sp_thres, G.s2);
    G.d2c_thres = gate_d2c(P.c_ell, P.sp_thres, P.c_sigma);
    G.den_l = 2.0 * l * l;
    G.den_c = 2.0 * P.c_ell * P.c_ell;
    G.inv_den_l = 1.0 / G.den_l;
    G.inv_den_c = 1.0 / G.den_c;
    G.q_il = (float)G.inv_den_l;
    G.q_ic = (float)G.inv_den_c;
    G.q_lim = logf(G.s2 * G.csig2 / P.sp_thres) * 1.001f + 1e-3f;   // a>sp  <=>  d2/den_l + d2c/den_c < ln(s2*csig2/sp)
    return G;
}

// line-search constants of one iteration (cvo.cpp:241-267), the same in every lane
struct LsConsts {
    float omega[3], v[3];
    float O2[9], O3[9], O4[9], Ov[3], O2v[3], O3v[3];
    float s_beta, s_gamma, s_delta;
};
__device__ __forceinline__ LsConsts make_ls(const float* omega, const float* v, float ell) {
    LsConsts L;
    float Oh[9];
    for (int q = 0; q < 3; ++q) { L.omega[q] = omega[q]; L.v[q] = v[q]; }
    skew3(omega, Oh);
    mat3_mul(Oh, Oh, L.O2); mat3_mul(L.O2, Oh, L.O3); mat3_mul(L.O3, Oh, L.O4);
    mat3_vec(Oh, v, L.Ov); mat3_vec(L.O2, v, L.O2v); mat3_vec(L.O3, v, L.O3v);
    const float temp_coef = (float)(1 / (2.0 * ell * ell));                              // cvo.cpp:267
    L.s_beta = (float)(-2.0 * temp_coef); L.s_gamma = -temp_coef; L.s_delta = (float)(2.0 * temp_coef);
    return L;
}
// one nonzero of A: adds its B, C, D, E terms (cvo.cpp:282-306)
__device__ __forceinline__ void ls_terms(const float* xi, const float4 yj, float A_ij, const LsConsts& L, double& Bi, double& Ci, double& Di, double& Ei) {
    const float y[3] = {yj.x, yj.y, yj.z};
    float z1[3], z2[3], z3[3], z4[3], t[3];
    cross3(L.omega, y, t); for (int q = 0; q < 3; ++q) z1[q] = t[q] + L.v[q];       // cvo.cpp:254
    mat3_vec(L.O2, y, t);  for (int q = 0; q < 3; ++q) z2[q] = t[q] + L.Ov[q];      // cvo.cpp:255-256
    mat3_vec(L.O3, y, t);  for (int q = 0; q < 3; ++q) z3[q] = t[q] + L.O2v[q];     // cvo.cpp:257-258
    mat3_vec(L.O4, y, t);  for (int q = 0; q < 3; ++q) z4[q] = t[q] + L.O3v[q];     // cvo.cpp:259-260
    const float nrm = dot3_seq(z1, z1);                                            // cvo.cpp:261
    const float mdot = -dot3_seq(z1, z2);                                          // cvo.cpp:262
    const float econst = dot3_seq(z2, z2) + 2 * dot3_seq(z1, z3);                  // cvo.cpp:263
    const float df[3] = {xi[0] - y[0], xi[1] - y[1], xi[2] - y[2]};                // cvo.cpp:286
    const float beta_ij = sum3f((L.s_beta * z1[0]) * df[0], (L.s_beta * z1[1]) * df[1], (L.s_beta * z1[2]) * df[2]);              // cvo.cpp:288
    const float gamma_ij = L.s_gamma * (nrm + sum3f((2.f * z2[0]) * df[0], (2.f * z2[1]) * df[1], (2.f * z2[2]) * df[2]));       // cvo.cpp:290-291
    const float delta_ij = L.s_delta * (mdot + sum3f((-z3[0]) * df[0], (-z3[1]) * df[1], (-z3[2]) * df[2]));                     // cvo.cpp:293-294
    const float epsil_ij = L.s_gamma * (econst + sum3f((2.f * z4[0]) * df[0], (2.f * z4[1]) * df[1], (2.f * z4[2]) * df[2]));    // cvo.cpp:296-297
    Bi += double(A_ij * beta_ij);                                                                                              // cvo.cpp:301
    Ci += double(A_ij * (gamma_ij + beta_ij * beta_ij / 2.0));                                                                 // cvo.cpp:302
    Di += double(A_ij * (delta_ij + beta_ij * gamma_ij + beta_ij * beta_ij * beta_ij / 6.0));                                  // cvo.cpp:303
    Ei += double(A_ij * (epsil_ij + beta_ij * delta_ij + 1 / 2.0 * beta_ij * beta_ij * gamma_ij                                // cvo.cpp:304-305
                         + 1 / 2.0 * gamma_ij * gamma_ij + 1 / 24.0 * beta_ij * beta_ij * beta_ij * beta_ij));
}

// ---------------------------------------------------------------- S: dense cull
// Branch-free, 7 VALU per pair test: 3 sub, 3 fma (the last one folds "- thr" in, so the sign
// bit of t = d2 - thr is the hit) and one v_alignbit that shifts the sign into the row's
// 32-column word, w = (w << 1) | sign(t): the first column of a group ends up in bit 31.
// (Packed v_pk_*_f32 forms were measured: they issue at half the rate, no gain.)
// A wave owns 64*R rows (R per lane) and one contiguous part of the tile's columns; columns
// come from LDS 4 at a time (ds_read_b128, the same address in every lane: broadcast), the
// next two quads are fetched while the current two are tested.  Every STG_GROUPS groups the
// wave transposes its words through a private LDS patch and writes them to the ROW-major
// bitmap bits[row][word] as 16-byte row segments, so the extraction pass reads a row's words
// with consecutive lanes and writes its candidate list coalesced.
constexpr int STG_GROUPS = 4;
constexpr int STG_STRIDE = 5;                                       // words per row in the patch (4 + 1 pad: conflict-free writes)
constexpr int STG_WORDS_PER_WAVE = 4 * 64 * STG_STRIDE;             // up to 4 rows per lane

template <int R, int RPT>
__device__ __forceinline__ void sweep_part(const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
                                           int ngroups /* multiple of STG_GROUPS */, int word_base, const float (&x)[RPT][3],
                                           const int (&row0)[RPT] /* local row of lane 0, per r */, int nrows, int (&cnt)[RPT],
                                           gu32* __restrict__ bits, int nwords_pad, volatile uint32_t* stg, float thr, int lane) {
    const float4* qx = reinterpret_cast<const float4*>(lx);
    const float4* qy = reinterpret_cast<const float4*>(ly);
    const float4* qz = reinterpret_cast<const float4*>(lz);
    const float nthr = -thr;
    const int nq2 = ngroups * 4;                                    // pairs of quads
    float4 X0 = qx[0], Y0 = qy[0], Z0 = qz[0], X1 = qx[1], Y1 = qy[1], Z1 = qz[1];
    uint32_t w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = 0u;
#pragma unroll 1
    for (int p = 0; p < nq2; ++p) {
        const int pn = (p + 1 < nq2) ? p + 1 : p;
        const float4 nX0 = qx[2 * pn], nY0 = qy[2 * pn], nZ0 = qz[2 * pn], nX1 = qx[2 * pn + 1], nY1 = qy[2 * pn + 1], nZ1 = qz[2 * pn + 1];
        const float cx[8] = {X0.x, X0.y, X0.z, X0.w, X1.x, X1.y, X1.z, X1.w};
        const float cy[8] = {Y0.x, Y0.y, Y0.z, Y0.w, Y1.x, Y1.y, Y1.z, Y1.w};
        const float cz[8] = {Z0.x, Z0.y, Z0.z, Z0.w, Z1.x, Z1.y, Z1.z, Z1.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float dx = x[r][0] - cx[u], dy = x[r][1] - cy[u], dz = x[r][2] - cz[u];
                const float t = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, nthr)));
                w[r] = __builtin_amdgcn_alignbit(w[r], __float_as_uint(t), 31);
            }
        }
        if ((p & 3) == 3) {                                         // 32 columns done: park the group's words
            const int grp = p >> 2;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                stg[(r * 64 + lane) * STG_STRIDE + (grp & (STG_GROUPS - 1))] = w[r];
                cnt[r] += __popc(w[r]);
                w[r] = 0u;
            }
            if ((grp & (STG_GROUPS - 1)) == STG_GROUPS - 1) {       // patch full: lanes (4 per row) write 16-byte row segments
                const int word0 = word_base + grp - (STG_GROUPS - 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const int rl = (lane >> 2) + 16 * k4, wd = lane & 3;
                        const uint32_t v = stg[(r * 64 + rl) * STG_STRIDE + wd];
                        const int li = row0[r] + rl;
                        if (li < nrows) bits[(size_t)li * nwords_pad + word0 + wd] = v;
                    }
                }
            }
        }
        X0 = nX0; Y0 = nY0; Z0 = nZ0; X1 = nX1; Y1 = nY1; Z1 = nZ1;
    }
}

// ---------------------------------------------------------------- the kernel
template <int RPT>
__global__ __launch_bounds__(512, CVO_WAVES_PER_SIMD) void cvo_align_kernel(const PairDesc* __restrict__ descs, int n_pairs, int G, int tile, DevParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Shared* sh = reinterpret_cast<Shared*>(smem);
    int* rowoff = reinterpret_cast<int*>(smem + ((sizeof(Shared) + 15) & ~size_t(15)));
    float* lx = reinterpret_cast<float*>(rowoff + (MAX_ROWS_PER_WG + 64));
    float* ly = lx + tile;
    float* lz = ly + tile;
    volatile uint32_t* stg_all = reinterpret_cast<volatile uint32_t*>(lz + tile);

    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int slots = gridDim.x / G, slot = blockIdx.x / G, g = blockIdx.x % G;
    if (slot >= slots) return;                                      // gridDim.x is a multiple of G; defensive

    for (int p = slot; p < n_pairs; p += slots) {
        const PairDesc D = descs[p];
        const int nf = D.nf, nm = D.nm;
        // rows are dealt round-robin to the pair's workgroups (local row li <-> fixed point g + G*li):
        // near surfaces have many more neighbours per row than far ones, bands of rows would be unbalanced
        const int rows_per = (nf + G - 1) / G;
        const int nrows = (g < nf) ? (nf - g + G - 1) / G : 0;
        const int ngroups_all = (nm + 31) >> 5;                     // 32-column groups of the hit bitmap
        const gfloat* fixed = (const gfloat*)D.fixed;
        const gfloat* moving = (const gfloat*)D.moving;
        const GF4 ybuf{(gv4f*)D.ybuf + (size_t)g * D.nm_pad};
        const GF4 ybuild{(gv4f*)D.ybuild + (size_t)g * D.nm_pad};      // positions the candidate lists were built at
        const int nwords_pad = D.nwords_pad;                        // ngroups_all rounded up to the flush granule
        gu32* bits = (gu32*)D.bits + (size_t)g * D.rows_pad * nwords_pad;   // [row][word], this workgroup's rows
        gint* cnt0 = (gint*)D.cnt; gint* cnt1 = cnt0 + D.nf_pad;    // hits per row found by column part 0 / 1
        gu32* flat_ij = (gu32*)D.flat_ij;
        const GF4 rec0{(gv4f*)D.rec0}, rec1{(gv4f*)D.rec1}, surv0{(gv4f*)D.surv0}, surv1{(gv4f*)D.surv1};
        gu64* xch = (gu64*)D.xch;
        const size_t fbase = (size_t)g * rows_per * D.capf;         // this workgroup's segment of the flat arrays
        const int flat_cap = rows_per * D.capf;

        if (tid == 0) {
            const PairState* st = D.state;
            for (int i = 0; i < 9; ++i) sh->R[i] = st->R[i];
            for (int i = 0; i < 3; ++i) sh->T[i] = st->T[i];
            for (int i = 0; i < 12; ++i) sh->M[i] = st->transform[i];
            sh->ell = st->ell;
            sh->stop = 0; sh->status = 0; sh->iter_at_break = st->iter; sh->nnz = 0; sh->cand = 0;
            sh->list_valid = 0; sh->dense_mode = 0; sh->total = 0; sh->rebuilds = 0; sh->dense_fallbacks = 0;
        }
        __syncthreads();

        int k = 0;
        long long cand_total = 0;
        unsigned long long ticks[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long t_prev = __builtin_amdgcn_s_memrealtime();
        const unsigned long long clk_t0 = t_prev, clk_c0 = __builtin_amdgcn_s_memtime();
#define CVO_PHASE(idx) do { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); ticks[idx] += t_now - t_prev; t_prev = t_now; } while (0)
        const bool ok_pair = (nf > 0 && nm > 0 && rows_per <= MAX_ROWS_PER_WG);
        if (!ok_pair && tid == 0) sh->status = (nf > 0 && nm > 0) ? 4 : 2;   // CVO_ERR_INVALID / CVO_ERR_EMPTY_CLOUD (reference: assert / UB, Q8)

        for (; ok_pair && k < P.max_iter; ++k) {
            // ---- update_tf (cvo.cpp:770): every lane forms the same 3x4 transform
            float M[12];
            {
                float R[9], T[3];
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = sh->R[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) T[i] = sh->T[i];
                make_transform(R, T, M);
            }
            const float ell = sh->ell;
            const Gates gates = make_gates(ell, P);
            const float r_c = sqrtf(gates.d2_thres);

            // ---- T: transform_pcd (cvo.cpp:336-341) into ybuf; on the way, how far has any point moved
            // since the candidate lists were built?  (exact displacement of the very positions the tests use)
            const bool have_list = sh->list_valid != 0;
            float dmax2 = 0.f;
            for (int j = tid; j < nm; j += nthreads) {
                const float4 lo = ld4(moving + (size_t)j * REC);
                float y0, y1, y2;
                apply_transform(M, lo.x, lo.y, lo.z, y0, y1, y2);
                ybuf.set(j, make_float4(y0, y1, y2, lo.w));
                if (have_list) {
                    const float4 yb = ybuild[j];
                    const float e0 = y0 - yb.x, e1 = y1 - yb.y, e2 = y2 - yb.z;
                    dmax2 = fmaxf(dmax2, __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0)));
                }
            }
            dmax2 = block_max(dmax2, sh, tid, nwaves);              // also makes ybuf visible to the workgroup
            // the lists hold every pair within Rb of the build positions; a pair within r_c now was within
            // r_c + (its point's displacement) then.  Every workgroup of the pair computes the same bits here.
            const bool rebuild = !have_list || (sh->ell_build != ell) ||
                                 (r_c + sqrtf(dmax2) * 1.0001f + 1.0e-5f) * 1.00001f > sh->Rb;

            if (rebuild) {
                // ---- S: for each block of RPT*nthreads rows, stream the moving cloud through LDS tiles
                const float Rb = r_c * (1.0f + P.skin);
                const float thr_cull = Rb * Rb * 1.00001f;
                // waves form a RGN x CPN grid: RGN groups of 64*RPT rows, CPN column parts of every tile, so
                // that all waves carry the same number of pair tests (two per SIMD, none left alone)
                const int wave = tid >> 6;
                const int CPN = (nwaves >= 8) ? 2 : 1, RGN = nwaves / CPN;
                const int rg = wave % RGN, cp = wave / RGN;
                volatile uint32_t* stg = stg_all + wave * STG_WORDS_PER_WAVE;
                gint* cnt_part = cp ? cnt1 : cnt0;
                for (int rb = 0; rb < nrows; rb += RPT * RGN * 64) {
                    float x[RPT][3]; int row0[RPT]; int cnt[RPT];
                    int nv = 0;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        row0[r] = rb + r * (RGN * 64) + rg * 64;
                        const int li = row0[r] + lane;
                        cnt[r] = 0;
                        if (li < nrows) {
                            const float4 lo = ld4(fixed + (size_t)(g + G * li) * REC);
                            x[r][0] = lo.x; x[r][1] = lo.y; x[r][2] = lo.z;
                        } else {
                            x[r][0] = x[r][1] = x[r][2] = FAR_ROW;
                        }
                        if (row0[r] < nrows) nv = r + 1;            // rows are a prefix per wave: skip the all-padding ones
                    }
                    for (int t0 = 0; t0 < nm; t0 += tile) {
                        const int tn = min(tile, nm - t0);
                        const int gran = 32 * STG_GROUPS * CPN;     // every column part is a whole number of flush granules
                        const int tnp = (tn + gran - 1) / gran * gran;
                        __syncthreads();                            // previous tile fully consumed
                        for (int jj = tid; jj < tnp; jj += nthreads) {
                            float4 y = make_float4(FAR_COL, FAR_COL, FAR_COL, 0.f);
                            if (jj < tn) { y = ybuf[t0 + jj]; if (rb == 0) ybuild.set(t0 + jj, y); }
                            lx[jj] = y.x; ly[jj] = y.y; lz[jj] = y.z;
                        }
                        __syncthreads();
                        const int ngp = (tnp >> 5) / CPN;           // groups in this wave's column part
                        const int c0 = cp * ngp * 32;               // its first column inside the tile
                        const int wb = (t0 >> 5) + cp * ngp;        // its first bitmap word
                        if (RPT >= 4 && nv == 4) sweep_part<(RPT >= 4 ? 4 : 1), RPT>(lx + c0, ly + c0, lz + c0, ngp, wb, x, row0, nrows, cnt, bits, nwords_pad, stg, thr_cull, lane);
                        else if (RPT >= 3 && nv == 3) sweep_part<(RPT >= 3 ? 3 : 1), RPT>(lx + c0, ly + c0, lz + c0, ngp, wb, x, row0, nrows, cnt, bits, nwords_pad, stg, thr_cull, lane);
                        else if (RPT >= 2 && nv == 2) sweep_part<(RPT >= 2 ? 2 : 1), RPT>(lx + c0, ly + c0, lz + c0, ngp, wb, x, row0, nrows, cnt, bits, nwords_pad, stg, thr_cull, lane);
                        else if (nv >= 1) sweep_part<1, RPT>(lx + c0, ly + c0, lz + c0, ngp, wb, x, row0, nrows, cnt, bits, nwords_pad, stg, thr_cull, lane);
                    }
#pragma unroll
                    for (int r = 0; r < RPT; ++r) { const int li = row0[r] + lane; if (li < nrows) cnt_part[g + G * li] = cnt[r]; }
                }
                __syncthreads();                                    // counts, bitmap visible to the whole workgroup
                unsigned long long t_sub = __builtin_amdgcn_s_memrealtime();
                ticks[6] += t_sub - t_prev;
                // ---- row offsets (exclusive scan of the counts), then the set bits become the flat lists
                const int rps = (nrows + nthreads - 1) / nthreads;
                const int l0 = min(nrows, tid * rps), l1 = min(nrows, l0 + rps);
                int mine = 0;
                for (int li = l0; li < l1; ++li) mine += cnt0[g + G * li] + (CPN > 1 ? cnt1[g + G * li] : 0);
                int total = 0;
                int run = block_exclusive_scan(mine, sh, tid, nwaves, total);
                for (int li = l0; li < l1; ++li) { rowoff[li] = run; run += cnt0[g + G * li] + (CPN > 1 ? cnt1[g + G * li] : 0); }
                if (tid == 0) rowoff[nrows] = total;
                const int dense = (total > flat_cap) ? 1 : 0;       // lists too small: dense per-row fallback until the next rebuild
                __syncthreads();
                { const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); ticks[7] += t_now - t_sub; t_sub = t_now; }
                if (!dense) {
                    // a wave takes four rows per trip (their bitmap words are loaded together); lanes = consecutive
                    // words of a row, an in-wave prefix sum of the popcounts places every lane's hits, so the
                    // packed (row << 16 | column) entries of a row leave the wave as one contiguous run
                    const int nwords = ngroups_all;
                    for (int li0 = wave * 4; li0 < nrows; li0 += nwaves * 4) {
                        size_t o[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) o[u] = fbase + rowoff[min(li0 + u, nrows)];
                        for (int wb0 = 0; wb0 < nwords; wb0 += 64) {
                            uint32_t wv[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                wv[u] = (li0 + u < nrows && wb0 + lane < nwords) ? bits[(size_t)(li0 + u) * nwords_pad + wb0 + lane] : 0u;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                uint32_t w = wv[u];
                                const int c = __popc(w);
                                int inc = c;
#pragma unroll
                                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
                                const int tot = __shfl(inc, 63, 64);
                                size_t pos = o[u] + (size_t)(inc - c);
                                const uint32_t tag = ((uint32_t)(li0 + u) << 16) | (uint32_t)((wb0 + lane) * 32);
                                while (w) {                         // bit 31 = first column of the group: ascending columns
                                    const int kbit = __clz(w);
                                    flat_ij[pos++] = tag + (uint32_t)kbit;
                                    w &= ~(0x80000000u >> kbit);
                                }
                                o[u] += tot;
                            }
                        }
                    }
                }
                if (tid == 0) {
                    sh->Rb = Rb; sh->ell_build = ell; sh->list_valid = 1;
                    sh->dense_mode = dense; sh->total = total;
                    sh->rebuilds += 1; sh->dense_fallbacks += dense ? 1 : 0;
                }
                __syncthreads();
                ticks[8] += __builtin_amdgcn_s_memrealtime() - t_sub;
            }
            const int total = sh->total;
            const bool dense_mode = sh->dense_mode != 0;
            CVO_PHASE(0);

            // ---- C: exact kernel values + compute_flow row sums (cvo.cpp:202-231)
            const float inv_c = 1 / P.c, inv_d = 1 / P.d;
            double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // omega[3], v[3], nnz, candidates
            unsigned long long t_c2 = 0;
            constexpr int CU = 2;
            const int CW = (((total + nwaves - 1) / nwaves) + 64 * CU - 1) / (64 * CU) * (64 * CU);   // candidates per wave chunk
            if (!dense_mode) {
                // C1: one lane per candidate, a contiguous chunk per wave; survivors are compacted (ballot +
                // prefix popcount keeps candidate order) into the wave's segment for the line-search phase
                const int wave = tid >> 6;
                const int c_begin = min(total, wave * CW), c_end = min(total, c_begin + CW);
                int wcount = 0;
                for (int cb = c_begin; cb < c_end; cb += 64 * CU) {  // CU candidates per lane per trip: their gathers are in flight together
                    int cc[CU]; bool val[CU]; int li4[CU], j4[CU];
#pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        cc[u] = cb + u * 64 + lane; val[u] = cc[u] < c_end;
                        const uint32_t ij = val[u] ? flat_ij[fbase + cc[u]] : 0u;
                        li4[u] = (int)(ij >> 16); j4[u] = (int)(ij & 0xFFFFu);
                    }
                    float4 lo4[CU], hi4[CU], yj4[CU], gj4[CU];
#pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        const gfloat* xr = fixed + (size_t)(g + G * li4[u]) * REC;
                        lo4[u] = ld4(xr); hi4[u] = ld4(xr + 4);
                        yj4[u] = ybuf[j4[u]]; gj4[u] = ld4(moving + (size_t)j4[u] * REC + 4);
                    }
#pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        float a = 0.f;
                        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, s0 = q0, s1 = q0;
                        if (val[u]) {
                            const float xi[3] = {lo4[u].x, lo4[u].y, lo4[u].z};
                            const float fi[5] = {lo4[u].w, hi4[u].x, hi4[u].y, hi4[u].z, hi4[u].w};
                            a = se_kernel_value(xi, fi, yj4[u], gj4[u], gates);
                            if (a > 0.f) {
                                const float yv[3] = {yj4[u].x, yj4[u].y, yj4[u].z};
                                float cr[3]; cross3(xi, yv, cr);    // cvo.cpp:216
                                q0 = make_float4(a * cr[0], a * cr[1], a * cr[2], a * (yv[0] - xi[0]));   // cvo.cpp:217
                                q1 = make_float4(a * (yv[1] - xi[1]), a * (yv[2] - xi[2]), a, 0.f);
                                s0 = make_float4(xi[0], xi[1], xi[2], a);
                                s1 = make_float4(yv[0], yv[1], yv[2], 0.f);
                            }
                            rec0.set(fbase + cc[u], q0); rec1.set(fbase + cc[u], q1);
                        }
                        const unsigned long long mask = __ballot(a > 0.f);
                        if (a > 0.f) {
                            const int pos = wcount + __popcll(mask & ((1ull << lane) - 1ull));
                            surv0.set(fbase + c_begin + pos, s0); surv1.set(fbase + c_begin + pos, s1);
                        }
                        wcount += __popcll(mask);
                    }
                }
                if (lane == 0) { sh->wcnt[wave] = wcount; acc8[6] = (double)wcount; }
                if (tid == 0) acc8[7] = (double)total;
                __syncthreads();
                t_c2 = __builtin_amdgcn_s_memrealtime();
                for (int li = tid; li < nrows; li += nthreads) {    // C2: one lane per row, f32 sums in column order
                    float sw[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
                    const int c1 = rowoff[li + 1];
                    int c = rowoff[li];
                    for (; c + 4 <= c1; c += 4) {                   // non-members hold +0: adding them changes nothing
                        float4 a0[4], a1[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { a0[u] = rec0[fbase + c + u]; a1[u] = rec1[fbase + c + u]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            sw[0] += a0[u].x; sw[1] += a0[u].y; sw[2] += a0[u].z;
                            sv[0] += a0[u].w; sv[1] += a1[u].x; sv[2] += a1[u].y;
                        }
                    }
                    for (; c < c1; ++c) {
                        const float4 a0 = rec0[fbase + c], a1 = rec1[fbase + c];
                        sw[0] += a0.x; sw[1] += a0.y; sw[2] += a0.z;
                        sv[0] += a0.w; sv[1] += a1.x; sv[2] += a1.y;
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) { acc8[q] += (double)(inv_c * sw[q]); acc8[3 + q] += (double)(inv_d * sv[q]); }   // cvo.cpp:222-223
                }
            } else {
                for (int li = tid; li < nrows; li += nthreads) {    // dense fallback: every column of the row
                    const int i = g + G * li;
                    const float4 lo = ld4(fixed + (size_t)i * REC), hi = ld4(fixed + (size_t)i * REC + 4);
                    const float xi[3] = {lo.x, lo.y, lo.z};
                    const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
                    float sw[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
                    int nz = 0;
                    for (int j = 0; j < nm; ++j) {
                        const float4 yj = ybuf[j];
                        const float a = se_kernel_value(xi, fi, yj, ld4(moving + (size_t)j * REC + 4), gates);
                        if (a > 0.f) {
                            const float yv[3] = {yj.x, yj.y, yj.z};
                            float cr[3]; cross3(xi, yv, cr);
                            sw[0] += a * cr[0]; sw[1] += a * cr[1]; sw[2] += a * cr[2];
                            sv[0] += a * (yv[0] - xi[0]); sv[1] += a * (yv[1] - xi[1]); sv[2] += a * (yv[2] - xi[2]);
                            ++nz;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) { acc8[q] += (double)(inv_c * sw[q]); acc8[3 + q] += (double)(inv_d * sv[q]); }
                    acc8[6] += (double)nz;
                    acc8[7] += (double)(cnt0[i] + (nwaves >= 8 ? cnt1[i] : 0));
                }
            }
            __syncthreads();
            if (!dense_mode) ticks[9] += __builtin_amdgcn_s_memrealtime() - t_c2;
            CVO_PHASE(1);
            block_reduce<8>(acc8, sh, tid, nwaves);
            if (G > 1) {
                if (tid < 64) { if (!group_exchange<8>(sh, xch, G, g, 2u * (unsigned)k + 1u, lane)) sh->status = 6; }
                __syncthreads();
            }
            if (sh->status != 0) break;
            float omega[3], v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { omega[q] = (float)sh->vals[q]; v[q] = (float)sh->vals[3 + q]; }   // cvo.cpp:234-235
            const int nnz = (int)sh->vals[6];
            const int ncand = (int)sh->vals[7];
            cand_total += ncand;
            CVO_PHASE(2);

            // ---- L: compute_step_size sums (cvo.cpp:239-315); f64 terms, one lane per nonzero
            const LsConsts ls = make_ls(omega, v, ell);
            double acc4[4] = {0, 0, 0, 0};
            if (!dense_mode) {
                for (int w = 0; w < nwaves; ++w) {                  // every wave's survivor segment, all lanes striding it
                    const int cnt_w = sh->wcnt[w];
                    const size_t sb = fbase + (size_t)min(total, w * CW);
                    for (int q = tid; q < cnt_w; q += nthreads) {
                        const float4 s0 = surv0[sb + q], s1 = surv1[sb + q];
                        const float xi[3] = {s0.x, s0.y, s0.z};
                        ls_terms(xi, s1, s0.w, ls, acc4[0], acc4[1], acc4[2], acc4[3]);
                    }
                }
            } else {
                for (int li = tid; li < nrows; li += nthreads) {
                    const int i = g + G * li;
                    const float4 lo = ld4(fixed + (size_t)i * REC), hi = ld4(fixed + (size_t)i * REC + 4);
                    const float xi[3] = {lo.x, lo.y, lo.z};
                    const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
                    double Bi = 0, Ci = 0, Di = 0, Ei = 0;
                    for (int j = 0; j < nm; ++j) {
                        const float4 yj = ybuf[j];
                        const float A_ij = se_kernel_value(xi, fi, yj, ld4(moving + (size_t)j * REC + 4), gates);
                        if (A_ij > 0.f) ls_terms(xi, yj, A_ij, ls, Bi, Ci, Di, Ei);
                    }
                    acc4[0] += Bi; acc4[1] += Ci; acc4[2] += Di; acc4[3] += Ei;
                }
            }
            __syncthreads();
            CVO_PHASE(3);
            block_reduce<4>(acc4, sh, tid, nwaves);
            if (G > 1) {
                if (tid < 64) { if (!group_exchange<4>(sh, xch, G, g, 2u * (unsigned)k + 2u, lane)) sh->status = 6; }
                __syncthreads();
            }
            if (sh->status != 0) break;
            CVO_PHASE(4);

            // ---- E: one lane finishes the iteration (every workgroup of the pair computes the same bits)
            if (tid == 0) {
                const double B = sh->vals[0], C = sh->vals[1], Dd = sh->vals[2], E = sh->vals[3];
                const float c3 = (float)(4.0 * float(E)), c2 = (float)(3.0 * float(Dd)), c1 = (float)(2.0 * float(C)), c0 = float(B);   // cvo.cpp:318
                const float step = cubic_step(c3, c2, c1, c0, P.min_step);
                float dist = -1.f;
                int stop = 0;
                if (norm3f(omega) < P.eps && norm3f(v) < P.eps) {                               // cvo.cpp:782
                    stop = 1;
                } else {
                    float dR[9], dT[3], R[9], T[3], RdT[3], Rn[9];
                    for (int i = 0; i < 9; ++i) R[i] = sh->R[i];
                    for (int i = 0; i < 3; ++i) T[i] = sh->T[i];
                    exp_sek3(omega, v, step, dR, dT);                                           // cvo.cpp:793
                    mat3_vec(R, dT, RdT);
                    for (int q = 0; q < 3; ++q) sh->T[q] = RdT[q] + T[q];                        // cvo.cpp:800
                    mat3_mul(R, dR, Rn);
                    for (int i = 0; i < 9; ++i) sh->R[i] = Rn[i];                                // cvo.cpp:801
                    dist = dist_se3(dR, dT);
                    if (dist < P.eps_2) stop = 1;                                               // cvo.cpp:804
                }
                if (stop) sh->iter_at_break = k;
                else {
                    float l = ell;                                                              // cvo.cpp:810-812
                    l = (k > 2) ? (float)0.10 : l;
                    l = (k > 9) ? (float)0.06 : l;
                    l = (k > 19) ? (float)0.03 : l;
                    sh->ell = l;
                }
                sh->stop = stop; sh->nnz = nnz; sh->cand = ncand; sh->step = step;
                for (int i = 0; i < 12; ++i) sh->M[i] = M[i];
                if (g == 0 && D.trace && k < D.trace_cap) {
                    TraceRow& tr = D.trace[k];
                    for (int q = 0; q < 3; ++q) { tr.omega[q] = omega[q]; tr.v[q] = v[q]; }
                    tr.nnz = nnz; tr.candidates = ncand; tr.B = B; tr.C = C; tr.D = Dd; tr.E = E;
                    tr.step = step; tr.ell = ell; tr.dist = dist; tr.pad_ = 0;
                    *D.trace_len = k + 1;
                }
            }
            __syncthreads();
            CVO_PHASE(5);
            if (sh->stop) { ++k; break; }
        }

        // ---- after the loop (cvo.cpp:815-817): write the pair's state back
        __syncthreads();
        if (tid == 0 && g == 0) {
            PairState* st = D.state;
            float R[9], T[3], M[12];
            for (int i = 0; i < 9; ++i) { R[i] = sh->R[i]; st->R[i] = R[i]; }
            for (int i = 0; i < 3; ++i) { T[i] = sh->T[i]; st->T[i] = T[i]; }
            make_transform(R, T, M);                                                            // final update_tf, cvo.cpp:817
            for (int i = 0; i < 12; ++i) { st->prev_transform[i] = sh->M[i]; st->transform[i] = M[i]; }
            st->ell = sh->ell;
            st->iter = sh->iter_at_break;                                                       // unchanged (stale) if no break: Q4
            st->A_nonzero = sh->nnz;
            st->iterations_run = k;
            st->status = sh->status;
            st->rebuilds = sh->rebuilds;
            st->dense_fallbacks = sh->dense_fallbacks;
            st->candidates_total = cand_total;
            for (int i = 0; i < 10; ++i) st->phase_ticks[i] = ticks[i];
            st->clk_cycles = __builtin_amdgcn_s_memtime() - clk_c0; st->clk_ticks = __builtin_amdgcn_s_memrealtime() - clk_t0;
        }
        __syncthreads();
    }
}

template __global__ void cvo_align_kernel<1>(const PairDesc*, int, int, int, DevParams);
template __global__ void cvo_align_kernel<2>(const PairDesc*, int, int, int, DevParams);
template __global__ void cvo_align_kernel<3>(const PairDesc*, int, int, int, DevParams);
template __global__ void cvo_align_kernel<4>(const PairDesc*, int, int, int, DevParams);

// 64-byte result record per pair for the cross-GPU gather: 12 floats of transform,
// then iter, A_nonzero, iterations_run, status as floats (exact below 2^24).
__global__ void cvo_pack_results_kernel(const PairState* __restrict__ st, float* __restrict__ out, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    for (int i = 0; i < 12; ++i) out[p * 16 + i] = st[p].transform[i];
    out[p * 16 + 12] = (float)st[p].iter; out[p * 16 + 13] = (float)st[p].A_nonzero;
    out[p * 16 + 14] = (float)st[p].iterations_run; out[p * 16 + 15] = (float)st[p].status;
}
hipError_t launch_pack_results(const PairState* st, float* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(cvo_pack_results_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, st, out, n);
    return hipGetLastError();
}

int align_blocks_per_cu() { return CVO_WAVES_PER_SIMD / 2; }

size_t align_shared_bytes(int tile) {
    return ((sizeof(Shared) + 15) & ~size_t(15)) + (size_t)(MAX_ROWS_PER_WG + 64) * sizeof(int) + (size_t)3 * tile * sizeof(float) +
           (size_t)MAX_WAVES * STG_WORDS_PER_WAVE * sizeof(uint32_t);
}
int align_tile_granule() { return 32 * STG_GROUPS * 2; }

hipError_t launch_align(int rpt, int grid, int block, int tile, hipStream_t stream, const PairDesc* descs, int n_pairs, int G, const DevParams& P) {
    const size_t shmem = align_shared_bytes(tile);
    void (*fn)(const PairDesc*, int, int, int, DevParams) = nullptr;
    switch (rpt) {
        case 1: fn = cvo_align_kernel<1>; break;
        case 2: fn = cvo_align_kernel<2>; break;
        case 3: fn = cvo_align_kernel<3>; break;
        default: fn = cvo_align_kernel<4>; break;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(block), shmem, stream, descs, n_pairs, G, tile, P);
    return hipGetLastError();
}

}  // namespace cvohip
