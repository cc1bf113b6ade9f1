// cvo_kernels.hip -- gfx950 kernels of the CVO alignment hot path.
//
// cvo_align_kernel: the whole of cvo::align() (thirdparty/cvo/src/cvo.cpp:763-821)
// for a batch of independent frame pairs in ONE persistent launch.  G workgroups
// cooperate on a pair (each owns a contiguous block of fixed-cloud rows) and stay
// resident for all of its iterations; R, T, ell never leave the device.
//
// One iteration (cvo.cpp:768-813) is
//   T  transform_pcd (cvo.cpp:336-341): moving cloud -> LDS tile (SoA) + ybuf
//   S  dense O(N*M) cull: every (row, column) pair tested against the radius gate
//      with 3 sub + 1 mul + 2 fma + 1 cmp; rows live in registers (RPT per lane),
//      columns are broadcast LDS reads (ds_read_b128, 4 columns per read);
//      hits are appended to the row's candidate list (ascending column order =
//      CSR order of Eigen::setFromTriplets, cvo.cpp:182)
//   C  candidates -> exact se_kernel arithmetic (cvo.cpp:166-175: un-fused f32 d2,
//      colour gate, double exp) and the row's f32 omega/v partial sums
//      (cvo.cpp:213-223); f64 across rows (cvo.cpp:226-230): wave shuffle, LDS
//   L  survivors -> beta..epsil, f64 B..E (cvo.cpp:282-306), same reduction
//   E  one lane: cubic, stop tests, Exp_SEK3, pose update, ell schedule
//      (cvo.cpp:317-333, 782-812)
// The cull S uses fused arithmetic and a threshold widened by 1e-6 (a superset of
// the reference's set); membership is then decided in C by the reference's own
// expression, so the set, every kernel value and every per-row sum follow the
// oracle's float sequence.
//
// MFMA is deliberately not used: S is a distance test + compare, C/L are
// exp-heavy survivor work; neither is a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvo_device.h"
#include "cvo_math.hpp"

namespace cvohip {

constexpr int MAX_WAVES = 16;
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
    int stop;
    int status;
    int iter_at_break;
    int broke;
    int nnz;
    int cand;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

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

// G workgroups of one pair swap K doubles: each publishes its partials as 2K
// 8-byte {tag = epoch, 32 payload bits} granules (one relaxed agent-scope store
// each: the data is its own flag, no fence), then wave 0 polls every
// workgroup's granules and adds them in workgroup order, so all G workgroups end
// with bit-identical totals and take identical branch decisions.  Two buffers
// alternate by epoch parity: a workgroup can run at most one phase ahead of the
// slowest member, so a buffer is never rewritten while someone still reads it.
// Called by wave 0 (all 64 lanes).  Returns false on timeout.
template <int K>
__device__ __forceinline__ bool group_exchange(Shared* sh, unsigned long long* xch, int G, int g, unsigned epoch, int lane) {
    unsigned long long* buf = xch + (size_t)(epoch & 1u) * G * XCH_WORDS;
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
    float q_lim, q_il, q_ic;  // conservative f32 pre-test of a > sp before the double exps
};

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
    const float k = (float)((double)G.s2 * exp((double)(-d2) / G.den_l));
    const float ck = (float)((double)G.csig2 * exp((double)(-d2c) / G.den_c));
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
    G.q_il = (float)(1.0 / G.den_l);
    G.q_ic = (float)(1.0 / G.den_c);
    G.q_lim = logf(G.s2 * G.csig2 / P.sp_thres) * 1.001f + 1e-3f;   // a>sp  <=>  d2/den_l + d2c/den_c < ln(s2*csig2/sp)
    return G;
}

// ---------------------------------------------------------------- S: dense cull
template <int RPT>
__device__ __forceinline__ void sweep_tile(const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
                                           int nquads, int col_base, const float (&x)[RPT][3], const int (&row)[RPT],
                                           int (&cnt)[RPT], uint16_t* __restrict__ jlist, int npad, int cap, float thr) {
    const float4* qx = reinterpret_cast<const float4*>(lx);
    const float4* qy = reinterpret_cast<const float4*>(ly);
    const float4* qz = reinterpret_cast<const float4*>(lz);
    for (int q = 0; q < nquads; ++q) {
        const float4 X = qx[q], Y = qy[q], Z = qz[q];              // same address in every lane: LDS broadcast
        const float cx[4] = {X.x, X.y, X.z, X.w}, cy[4] = {Y.x, Y.y, Y.z, Y.w}, cz[4] = {Z.x, Z.y, Z.z, Z.w};
        bool hit[RPT][4];
        bool any = false;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dx = x[r][0] - cx[u], dy = x[r][1] - cy[u], dz = x[r][2] - cz[u];
                const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                hit[r][u] = d2 < thr;
                any |= hit[r][u];
            }
        }
        if (any) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (hit[r][u]) {
                        if (cnt[r] < cap) jlist[(size_t)cnt[r] * npad + row[r]] = (uint16_t)(col_base + 4 * q + u);
                        cnt[r]++;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- the kernel
template <int RPT>
__global__ __launch_bounds__(1024) void cvo_align_kernel(const PairDesc* __restrict__ descs, int n_pairs, int G, int tile, DevParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Shared* sh = reinterpret_cast<Shared*>(smem);
    float* lx = reinterpret_cast<float*>(smem + ((sizeof(Shared) + 15) & ~size_t(15)));
    float* ly = lx + tile;
    float* lz = ly + tile;

    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int slots = gridDim.x / G, slot = blockIdx.x / G, g = blockIdx.x % G;
    if (slot >= slots) return;                                      // gridDim.x is a multiple of G; defensive

    for (int p = slot; p < n_pairs; p += slots) {
        const PairDesc D = descs[p];
        const int nf = D.nf, nm = D.nm, npad = D.nf_pad, cap = D.cap;
        const int rows_per = (nf + G - 1) / G;
        const int r0 = min(nf, g * rows_per), r1 = min(nf, r0 + rows_per);
        float4* ybuf = D.ybuf + (size_t)g * D.nm_pad;

        if (tid == 0) {
            const PairState* st = D.state;
            for (int i = 0; i < 9; ++i) sh->R[i] = st->R[i];
            for (int i = 0; i < 3; ++i) sh->T[i] = st->T[i];
            sh->ell = st->ell;
            for (int i = 0; i < 12; ++i) sh->M[i] = st->transform[i];
            sh->stop = 0; sh->status = 0; sh->broke = 0; sh->iter_at_break = st->iter; sh->nnz = 0; sh->cand = 0;
        }
        __syncthreads();

        int k = 0;
        long long cand_total = 0;
        bool ok_pair = (nf > 0 && nm > 0);
        if (!ok_pair && tid == 0) sh->status = 2;                   // CVO_ERR_EMPTY_CLOUD (reference: assert / UB, Q8)

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
            const float thr_cull = gates.d2_thres * (1.0f + 1e-6f);

            // ---- T + S: for each block of RPT*nthreads rows, stream the moving cloud through LDS tiles
            for (int rb = r0; rb < r1; rb += RPT * nthreads) {
                float x[RPT][3]; int row[RPT]; int cnt[RPT];
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const int i = rb + tid + r * nthreads;
                    cnt[r] = 0;
                    if (i < r1) {
                        const float4 lo = ld4(D.fixed + (size_t)i * REC);
                        x[r][0] = lo.x; x[r][1] = lo.y; x[r][2] = lo.z; row[r] = i;
                    } else {
                        x[r][0] = x[r][1] = x[r][2] = FAR_ROW; row[r] = -1;
                    }
                }
                for (int t0 = 0; t0 < nm; t0 += tile) {
                    const int tn = min(tile, nm - t0);
                    const int tn4 = (tn + 3) & ~3;
                    __syncthreads();                                // previous tile fully consumed
                    for (int jj = tid; jj < tn4; jj += nthreads) {
                        float y0 = FAR_COL, y1 = FAR_COL, y2 = FAR_COL;
                        if (jj < tn) {
                            const float4 lo = ld4(D.moving + (size_t)(t0 + jj) * REC);
                            apply_transform(M, lo.x, lo.y, lo.z, y0, y1, y2);   // transform_pcd, cvo.cpp:338
                            if (rb == r0) ybuf[t0 + jj] = make_float4(y0, y1, y2, lo.w);
                        }
                        lx[jj] = y0; ly[jj] = y1; lz[jj] = y2;
                    }
                    __syncthreads();
                    sweep_tile<RPT>(lx, ly, lz, tn4 >> 2, t0, x, row, cnt, D.jlist, npad, cap, thr_cull);
                }
#pragma unroll
                for (int r = 0; r < RPT; ++r) if (row[r] >= 0) D.cnt[row[r]] = cnt[r];
            }
            __syncthreads();                                        // ybuf, cnt, jlist visible to the whole workgroup

            // ---- C: exact kernel values on the candidates + compute_flow row sums (cvo.cpp:202-231)
            const float inv_c = 1 / P.c, inv_d = 1 / P.d;
            double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // omega[3], v[3], nnz, candidates
            for (int i = r0 + tid; i < r1; i += nthreads) {
                const float4 lo = ld4(D.fixed + (size_t)i * REC), hi = ld4(D.fixed + (size_t)i * REC + 4);
                const float xi[3] = {lo.x, lo.y, lo.z};
                const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
                const int c = D.cnt[i];
                float sw[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
                int nz = 0;
                const bool listed = (c <= cap);
                const int trips = listed ? c : nm;                  // overflowed row: dense fallback over every column
                for (int n = 0; n < trips; ++n) {
                    const int j = listed ? (int)D.jlist[(size_t)n * npad + i] : n;
                    const float4 yj = ybuf[j];
                    const float4 gj = ld4(D.moving + (size_t)j * REC + 4);
                    const float a = se_kernel_value(xi, fi, yj, gj, gates);
                    if (listed) D.alist[(size_t)n * npad + i] = a;
                    if (a > 0.f) {
                        const float yv[3] = {yj.x, yj.y, yj.z};
                        float cr[3]; cross3(xi, yv, cr);            // cvo.cpp:216
                        sw[0] += a * cr[0]; sw[1] += a * cr[1]; sw[2] += a * cr[2];
                        sv[0] += a * (yv[0] - xi[0]); sv[1] += a * (yv[1] - xi[1]); sv[2] += a * (yv[2] - xi[2]);   // cvo.cpp:217
                        ++nz;
                    }
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) { acc8[q] += (double)(inv_c * sw[q]); acc8[3 + q] += (double)(inv_d * sv[q]); }   // cvo.cpp:222-223
                acc8[6] += (double)nz;
                acc8[7] += (double)c;
            }
            block_reduce<8>(acc8, sh, tid, nwaves);
            if (G > 1) {
                if (tid < 64) { if (!group_exchange<8>(sh, D.xch, G, g, 2u * (unsigned)k + 1u, lane)) sh->status = 6; }
                __syncthreads();
            }
            if (sh->status != 0) break;
            float omega[3], v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { omega[q] = (float)sh->vals[q]; v[q] = (float)sh->vals[3 + q]; }   // cvo.cpp:234-235
            const int nnz = (int)sh->vals[6];
            const int ncand = (int)sh->vals[7];
            cand_total += ncand;

            // ---- L: compute_step_size sums (cvo.cpp:239-315)
            float Oh[9], O2[9], O3[9], O4[9], Ov[3], O2v[3], O3v[3];
            skew3(omega, Oh);
            mat3_mul(Oh, Oh, O2); mat3_mul(O2, Oh, O3); mat3_mul(O3, Oh, O4);
            mat3_vec(Oh, v, Ov); mat3_vec(O2, v, O2v); mat3_vec(O3, v, O3v);
            const float temp_coef = (float)(1 / (2.0 * ell * ell));                              // cvo.cpp:267
            const float s_beta = (float)(-2.0 * temp_coef), s_gamma = -temp_coef, s_delta = (float)(2.0 * temp_coef);
            double acc4[4] = {0, 0, 0, 0};
            for (int i = r0 + tid; i < r1; i += nthreads) {
                const float4 lo = ld4(D.fixed + (size_t)i * REC);
                const float xi[3] = {lo.x, lo.y, lo.z};
                float fi[5] = {lo.w, 0, 0, 0, 0};
                const int c = D.cnt[i];
                const bool listed = (c <= cap);
                if (!listed) { const float4 hi = ld4(D.fixed + (size_t)i * REC + 4); fi[1] = hi.x; fi[2] = hi.y; fi[3] = hi.z; fi[4] = hi.w; }
                const int trips = listed ? c : nm;
                double Bi = 0, Ci = 0, Di = 0, Ei = 0;
                for (int n = 0; n < trips; ++n) {
                    int j; float A_ij; float4 yj;
                    if (listed) {
                        A_ij = D.alist[(size_t)n * npad + i];
                        if (!(A_ij > 0.f)) continue;
                        j = (int)D.jlist[(size_t)n * npad + i];
                        yj = ybuf[j];
                    } else {
                        j = n; yj = ybuf[j];
                        A_ij = se_kernel_value(xi, fi, yj, ld4(D.moving + (size_t)j * REC + 4), gates);
                        if (!(A_ij > 0.f)) continue;
                    }
                    const float y[3] = {yj.x, yj.y, yj.z};
                    float z1[3], z2[3], z3[3], z4[3], t[3];
                    cross3(omega, y, t); for (int q = 0; q < 3; ++q) z1[q] = t[q] + v[q];       // cvo.cpp:254
                    mat3_vec(O2, y, t);  for (int q = 0; q < 3; ++q) z2[q] = t[q] + Ov[q];      // cvo.cpp:255-256
                    mat3_vec(O3, y, t);  for (int q = 0; q < 3; ++q) z3[q] = t[q] + O2v[q];     // cvo.cpp:257-258
                    mat3_vec(O4, y, t);  for (int q = 0; q < 3; ++q) z4[q] = t[q] + O3v[q];     // cvo.cpp:259-260
                    const float nrm = dot3_seq(z1, z1);                                        // cvo.cpp:261
                    const float mdot = -dot3_seq(z1, z2);                                      // cvo.cpp:262
                    const float econst = dot3_seq(z2, z2) + 2 * dot3_seq(z1, z3);              // cvo.cpp:263
                    const float df[3] = {xi[0] - y[0], xi[1] - y[1], xi[2] - y[2]};            // cvo.cpp:286
                    const float beta_ij = sum3f((s_beta * z1[0]) * df[0], (s_beta * z1[1]) * df[1], (s_beta * z1[2]) * df[2]);                  // cvo.cpp:288
                    const float gamma_ij = s_gamma * (nrm + sum3f((2.f * z2[0]) * df[0], (2.f * z2[1]) * df[1], (2.f * z2[2]) * df[2]));         // cvo.cpp:290-291
                    const float delta_ij = s_delta * (mdot + sum3f((-z3[0]) * df[0], (-z3[1]) * df[1], (-z3[2]) * df[2]));                       // cvo.cpp:293-294
                    const float epsil_ij = s_gamma * (econst + sum3f((2.f * z4[0]) * df[0], (2.f * z4[1]) * df[1], (2.f * z4[2]) * df[2]));      // cvo.cpp:296-297
                    Bi += double(A_ij * beta_ij);                                                                                            // cvo.cpp:301
                    Ci += double(A_ij * (gamma_ij + beta_ij * beta_ij / 2.0));                                                               // cvo.cpp:302
                    Di += double(A_ij * (delta_ij + beta_ij * gamma_ij + beta_ij * beta_ij * beta_ij / 6.0));                                // cvo.cpp:303
                    Ei += double(A_ij * (epsil_ij + beta_ij * delta_ij + 1 / 2.0 * beta_ij * beta_ij * gamma_ij                              // cvo.cpp:304-305
                                         + 1 / 2.0 * gamma_ij * gamma_ij + 1 / 24.0 * beta_ij * beta_ij * beta_ij * beta_ij));
                }
                acc4[0] += Bi; acc4[1] += Ci; acc4[2] += Di; acc4[3] += Ei;
            }
            block_reduce<4>(acc4, sh, tid, nwaves);
            if (G > 1) {
                if (tid < 64) { if (!group_exchange<4>(sh, D.xch, G, g, 2u * (unsigned)k + 2u, lane)) sh->status = 6; }
                __syncthreads();
            }
            if (sh->status != 0) break;

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
                if (stop) { sh->broke = 1; sh->iter_at_break = k; }
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
            st->candidates_total = cand_total;
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

size_t align_shared_bytes(int tile) { return ((sizeof(Shared) + 15) & ~size_t(15)) + (size_t)3 * tile * sizeof(float); }

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
