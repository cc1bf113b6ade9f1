// cvo_score_kernels.hip -- function_inner_product (cvo.cpp:388-459) and
// se3_Hessian (cvo.cpp:620-759) as one tiled all-pairs kernel; a whole score block
// (compute_innerproduct: 4 inner products + 1 Hessian, cvo.cpp:475-503; the loop-closure
// variant: 6 + 2, cvo.cpp:505-561) is ONE launch.
//
// Grid = (64-row blocks of cloud a) x (column chunks of cloud b) x (requests).  Thread = one
// point of cloud a (optionally transformed first, cvo.cpp:485-487); the chunk of cloud b
// sits in LDS (SoA positions, broadcast reads).  The radius gate IS binding here (no
// a>sp_thres test, Q6), so the fused cull is followed by the reference's own un-fused d2
// expression before a pair counts.  Per-thread sums are f32 for the Hessian (the reference
// keeps an f32 Hessian, cvo.cpp:622,707), f64 across threads; every workgroup writes one
// partial record and a second tiny kernel adds the records of a request in a fixed order
// straight into pinned host memory, so results are reproducible run to run and no copy
// engine is involved.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvo_device.h"
#include "cvo_math.hpp"

namespace cvohip {

constexpr int SCORE_BLOCK = 64;
constexpr int SCORE_TILE = 512;
constexpr int SCORE_NOUT = 24;     // sum_A, count, 21 Hessian terms, pad

__device__ __forceinline__ float4 ld4s(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ __launch_bounds__(SCORE_BLOCK) void cvo_score_kernel(ScoreBatch B, const ScoreDesc* __restrict__ more, DevParams P, double* __restrict__ partials) {
    const ScoreDesc& D = more ? more[blockIdx.z] : B.d[blockIdx.z];     // a tracker's score block travels in the kernel arguments, a batch's in HBM
    __shared__ __attribute__((aligned(16))) float lx[SCORE_TILE];
    __shared__ __attribute__((aligned(16))) float ly[SCORE_TILE];
    __shared__ __attribute__((aligned(16))) float lz[SCORE_TILE];

    const int tid = threadIdx.x, i = blockIdx.x * SCORE_BLOCK + tid;
    const float ell = D.from ? D.from->ell : D.ell, sigma = P.sigma;
    const float d2_thres = gate_d2_score(ell, P.sp_thres, sigma);                // cvo.cpp:395 / 626
    const float d2c_thres = gate_d2c(P.c_ell, P.sp_thres, P.c_sigma);            // cvo.cpp:396 / 627
    const float thr_cull = d2_thres * (1.0f + 1e-6f);
    const double den_l = 2.0 * ell * ell, den_c = 2.0 * P.c_ell * P.c_ell;
    const float sig2 = sigma * sigma, csig2 = P.c_sigma * P.c_sigma;
    const float il2 = 1 / (ell * ell);

    float pa[3] = {3.0e18f, 3.0e18f, 3.0e18f};
    float fa[5] = {0, 0, 0, 0, 0};
    const bool valid = i < D.na;
    if (valid) {
        const float4 lo = ld4s(D.a + lo_off(i)), hi = ld4s(D.a + hi_off(D.na, i));
        if (D.use_tran) apply_transform(D.use_tran == 2 ? D.from->transform : D.tran, lo.x, lo.y, lo.z, pa[0], pa[1], pa[2]);
        else { pa[0] = lo.x; pa[1] = lo.y; pa[2] = lo.z; }
        fa[0] = lo.w; fa[1] = hi.x; fa[2] = hi.y; fa[3] = hi.z; fa[4] = hi.w;
    }

    double sumA = 0; int count = 0;
    float H[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) H[q] = 0.f;

    // this workgroup's chunk of cloud b
    const int csize = (((D.nb + (int)gridDim.y - 1) / (int)gridDim.y) + 3) & ~3;
    const int c_begin = min(D.nb, (int)blockIdx.y * csize), c_end = min(D.nb, c_begin + csize);
    for (int t0 = c_begin; t0 < c_end; t0 += SCORE_TILE) {
        const int tn = min(SCORE_TILE, c_end - t0), tn4 = (tn + 3) & ~3;
        __syncthreads();
        for (int jj = tid; jj < tn4; jj += SCORE_BLOCK) {
            float b0 = -3.0e18f, b1 = -3.0e18f, b2 = -3.0e18f;
            if (jj < tn) { const float4 lo = ld4s(D.b + lo_off(t0 + jj)); b0 = lo.x; b1 = lo.y; b2 = lo.z; }
            lx[jj] = b0; ly[jj] = b1; lz[jj] = b2;
        }
        __syncthreads();
        const float4* qx = reinterpret_cast<const float4*>(lx);
        const float4* qy = reinterpret_cast<const float4*>(ly);
        const float4* qz = reinterpret_cast<const float4*>(lz);
        for (int q = 0; q < (tn4 >> 2); ++q) {
            const float4 X = qx[q], Y = qy[q], Z = qz[q];
            const float cx[4] = {X.x, X.y, X.z, X.w}, cy[4] = {Y.x, Y.y, Y.z, Y.w}, cz[4] = {Z.x, Z.y, Z.z, Z.w};
            bool hit[4]; bool any = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dx = pa[0] - cx[u], dy = pa[1] - cy[u], dz = pa[2] - cz[u];
                hit[u] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) < thr_cull;
                any |= hit[u];
            }
            if (!any) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!hit[u]) continue;
                const int j = t0 + 4 * q + u;
                const float pb[3] = {cx[u], cy[u], cz[u]};
                const float e0 = pa[0] - pb[0], e1 = pa[1] - pb[1], e2 = pa[2] - pb[2];
                float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;            // nanoflann.hpp:403-406
                if (!(d2 < d2_thres)) continue;                                      // cvo.cpp:423 / 654
                const float4 blo = ld4s(D.b + lo_off(j)), bhi = ld4s(D.b + hi_off(D.nb, j));
                const float fb[5] = {blo.w, bhi.x, bhi.y, bhi.z, bhi.w};
                float t[5];
#pragma unroll
                for (int c = 0; c < 5; ++c) { const float e = fa[c] - fb[c]; t[c] = e * e; }
                const float d2c = (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
                if (!(d2c < d2c_thres)) continue;                                    // cvo.cpp:428 / 659
                const float k = (float)((double)sig2 * exp((double)(-d2) / den_l)); // cvo.cpp:429 / 661
                if (!D.want_hessian) {
                    const float ck = (float)((double)csig2 * exp((double)(-d2c) / den_c));   // cvo.cpp:430
                    const float a = ck * k;
                    sumA += a; count += 1;                                           // cvo.cpp:432-435
                } else {
#pragma unroll
                    for (int c = 0; c < 5; ++c) t[c] = fa[c] * fb[c];
                    const float cdot = (t[0] + t[1]) + (t[2] + (t[3] + t[4]));       // cvo.cpp:662
                    float cr[3]; cross3(pa, pb, cr);
                    const float dot1 = pa[1] * pb[1] + pa[2] * pb[2], dot2 = pa[0] * pb[0] + pa[2] * pb[2], dot3 = pa[0] * pb[0] + pa[1] * pb[1];
                    const float db[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
                    float B[21];
                    // block A (symmetric): 00 01 02 11 12 22                          cvo.cpp:670-675
                    B[0] = il2 * cr[0] * cr[0] - dot1;
                    B[1] = (float)(il2 * cr[0] * cr[1] + 0.5 * (pa[0] * pb[1] + pa[1] * pb[0]));
                    B[2] = (float)(il2 * cr[0] * cr[2] + 0.5 * (pa[0] * pb[2] + pa[2] * pb[0]));
                    B[3] = il2 * cr[1] * cr[1] - dot2;
                    B[4] = (float)(il2 * cr[1] * cr[2] + 0.5 * (pa[1] * pb[2] + pa[2] * pb[1]));
                    B[5] = il2 * cr[2] * cr[2] - dot3;
                    // block C (full 3x3, row-major C(r,c))                            cvo.cpp:680-688
                    B[6] = il2 * cr[0] * db[0];          B[7] = -pa[2] + il2 * db[0] * cr[1];  B[8] = pa[1] + il2 * db[0] * cr[2];
                    B[9] = pa[2] + il2 * db[1] * cr[0];  B[10] = il2 * cr[1] * db[1];          B[11] = -pa[0] + il2 * db[1] * cr[2];
                    B[12] = -pa[1] + il2 * db[2] * cr[0]; B[13] = pa[0] + il2 * db[2] * cr[1]; B[14] = il2 * cr[2] * db[2];
                    // block D (symmetric): 00 01 02 11 12 22                          cvo.cpp:692-697
                    B[15] = il2 * db[0] * db[0] - 1; B[16] = il2 * db[0] * db[1]; B[17] = il2 * db[0] * db[2];
                    B[18] = il2 * db[1] * db[1] - 1; B[19] = il2 * db[1] * db[2]; B[20] = il2 * db[2] * db[2] - 1;
                    const float w = il2 * cdot * k;                                  // cvo.cpp:707
#pragma unroll
                    for (int q2 = 0; q2 < 21; ++q2) H[q2] += w * B[q2];
                    count += 1;
                }
            }
        }
    }

    double v[SCORE_NOUT];
    v[0] = sumA; v[1] = (double)count;
#pragma unroll
    for (int q = 0; q < 21; ++q) v[2 + q] = (double)H[q];
    v[23] = 0;
#pragma unroll
    for (int q = 0; q < SCORE_NOUT; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
    }
    const size_t rec = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tid < SCORE_NOUT) {
        double mine = 0;
#pragma unroll
        for (int q = 0; q < SCORE_NOUT; ++q) mine = (tid == q) ? v[q] : mine;
        partials[rec * SCORE_NOUT + tid] = mine;                    // row blocks past the end of a request's cloud write zeros
    }
}

// one workgroup per request: its partial records, in record order, into out[request][24] (pinned host memory)
__global__ __launch_bounds__(64) void cvo_score_reduce_kernel(const double* __restrict__ partials, int records_per_request, double* __restrict__ out) {
    const int tid = threadIdx.x;
    if (tid >= SCORE_NOUT) return;
    const double* p = partials + (size_t)blockIdx.x * records_per_request * SCORE_NOUT;
    double s = 0;
    for (int r = 0; r < records_per_request; ++r) s += p[(size_t)r * SCORE_NOUT + tid];
    out[blockIdx.x * SCORE_NOUT + tid] = s;
}

int score_nout() { return SCORE_NOUT; }
int score_row_blocks(int na) { return (na + SCORE_BLOCK - 1) / SCORE_BLOCK; }

// one launch for the whole batch of requests; out_pinned[request][24] is complete when the stream has drained
hipError_t launch_score(const ScoreBatch& B, const ScoreDesc* more, int nreq, int row_blocks, int chunks, const DevParams& P, double* partials,
                        double* out_pinned, hipStream_t stream) {
    hipLaunchKernelGGL(cvo_score_kernel, dim3(row_blocks, chunks, nreq), dim3(SCORE_BLOCK), 0, stream, B, more, P, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cvo_score_reduce_kernel, dim3(nreq), dim3(64), 0, stream, partials, row_blocks * chunks, out_pinned);
    return hipGetLastError();
}

}  // namespace cvohip
