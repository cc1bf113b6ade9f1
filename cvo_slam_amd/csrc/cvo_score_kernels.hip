// cvo_score_kernels.hip -- function_inner_product (cvo.cpp:388-459) and
// se3_Hessian (cvo.cpp:620-759) as one all-pairs kernel; a whole score block
// (compute_innerproduct: 4 inner products + 1 Hessian, cvo.cpp:475-503; the loop-closure
// variant: 6 + 2, cvo.cpp:505-561; a batch of either) is ONE launch.
//
// Grid = (64-row blocks of cloud a) x (column chunks of cloud b) x (requests).  Workgroup =
// one wave, thread = one point of cloud a (optionally transformed first, cvo.cpp:485-487).
// The radius search the reference runs per point (KD-tree, nanoflann) is restated as a
// box cull: clouds come in image scan order, so 32 consecutive points of cloud b span the
// image width but only a few image rows -- every cloud carries the bounding boxes of its
// 32-point groups (x, y, z and the ray slope y/z; computed once per cloud by
// cvo_cloud_boxes_kernel), the wave tests its own rows' box against 64 group boxes at a
// time, and only the groups that can hold a neighbour (a few percent at the radii in use)
// are staged in LDS and swept.  The radius gate IS binding here (no a>sp_thres test, Q6),
// so a hit of the fused sweep is re-tested with the reference's own un-fused d2 expression
// before it counts.  Column order within a row is ascending as in the reference's sorted
// radius search; per-thread sums are f32 for the Hessian (the reference keeps an f32
// Hessian, cvo.cpp:622,707), f64 across threads; every workgroup writes one partial record
// and a second tiny kernel adds the records of a request in a fixed order straight into
// pinned host memory, so results are reproducible run to run and no copy engine is
// involved.  Arbitrarily ordered clouds stay correct: their boxes are just loose.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvo_device.h"
#include "cvo_math.hpp"

namespace cvohip {

constexpr int SCORE_BLOCK = 64;
constexpr int SCORE_NOUT = 24;     // sum_A, count, 21 Hessian terms, pad
constexpr int SCORE_STAGE = 4;     // near groups fetched per round

__device__ __forceinline__ float4 ld4s(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float wmin(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// boxes of a cloud's 32-point groups: planes lo x, y, z, slope then hi x, y, z, slope, ngroups floats each; behind them the cloud's
// table of cached self inner products (ScoreDesc::self_cache), emptied here: the boxes are remade whenever the points were written
__global__ __launch_bounds__(64) void cvo_cloud_boxes_kernel(const float* __restrict__ rec, int n, float* __restrict__ gbox, int ngroups, SelfCacheEntry* __restrict__ self_cache) {
    const int lane = threadIdx.x, gi = blockIdx.x * 2 + (lane >> 5), j = gi * 32 + (lane & 31);
    if (blockIdx.x == 0 && lane < SELF_CACHE_N) { SelfCacheEntry e; e.ell = 0.f; e.valid = 0; e.sum = 0.0; e.count = 0.0; self_cache[lane] = e; }
    const float INF = __builtin_inff();
    float lo[4] = {INF, INF, INF, INF}, hi[4] = {-INF, -INF, -INF, -INF};
    if (j < n) {
        const float4 p = ld4s(rec + lo_off(j));
        lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
        if (p.z > 1.0e-3f) { lo[3] = hi[3] = p.y / p.z; } else { lo[3] = -INF; hi[3] = INF; }   // behind / at the camera: no slope bound
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = fminf(lo[q], __shfl_xor(lo[q], off, 64)); hi[q] = fmaxf(hi[q], __shfl_xor(hi[q], off, 64)); }
    }
    if ((lane & 31) == 0 && gi < ngroups) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { gbox[q * ngroups + gi] = lo[q]; gbox[(4 + q) * ngroups + gi] = hi[q]; }
    }
}

__global__ __launch_bounds__(SCORE_BLOCK) void cvo_score_kernel(ScoreBatch B, const ScoreDesc* __restrict__ more, DevParams P, double* __restrict__ partials,
                                                                unsigned* __restrict__ wgs_started) {
    // adoption's "is anything queued on the device?" (cvo_capi.hip, AdoptCounters): this workgroup has started
    if (wgs_started && threadIdx.x == 0) atomicAdd(wgs_started, 1u);
    const ScoreDesc& D = more ? more[blockIdx.z] : B.d[blockIdx.z];     // a tracker's score block travels in the kernel arguments, a batch's in HBM
    __shared__ __attribute__((aligned(16))) float lx[32 * SCORE_STAGE];
    __shared__ __attribute__((aligned(16))) float ly[32 * SCORE_STAGE];
    __shared__ __attribute__((aligned(16))) float lz[32 * SCORE_STAGE];

    const int tid = threadIdx.x, i = blockIdx.x * SCORE_BLOCK + tid;
    const float ell = D.from ? D.from->ell : D.ell, sigma = P.sigma;
    if (D.self_cache) {
        // fip(cloud, cloud) at this ell is already known (cvo.cpp:496-497 depend on the cloud and ell alone): the request's first record
        // carries the cached sums, the others zeros, and the reduction below adds them up to the very same doubles.  Every workgroup of
        // the launch sees the same table: it is written by the reduce kernel only, and launches that touch a cloud are ordered (ensure_boxes).
        int hit = -1;
#pragma unroll
        for (int e = 0; e < SELF_CACHE_N; ++e) if (D.self_cache[e].valid && D.self_cache[e].ell == ell) hit = e;
        if (hit >= 0) {
            const size_t rec0 = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            const bool first = blockIdx.x == 0 && blockIdx.y == 0;
            if (tid < SCORE_NOUT) partials[rec0 * SCORE_NOUT + tid] = (first && tid == 0) ? D.self_cache[hit].sum : ((first && tid == 1) ? D.self_cache[hit].count : 0.0);
            return;
        }
    }
    const float d2_thres = gate_d2_score(ell, P.sp_thres, sigma);                // cvo.cpp:395 / 626
    const float d2c_thres = gate_d2c(P.c_ell, P.sp_thres, P.c_sigma);            // cvo.cpp:396 / 627
    const float thr_cull = d2_thres * (1.0f + 1e-6f);
    const float thr_box = thr_cull * 1.001f;                                     // box gaps are compared with a margin: a skipped group holds no hit
    const float Rb = sqrtf(fmaxf(thr_cull, 0.f));
    const double den_l = 2.0 * ell * ell, den_c = 2.0 * P.c_ell * P.c_ell;
    const float sig2 = sigma * sigma, csig2 = P.c_sigma * P.c_sigma;
    const float il2 = 1 / (ell * ell);
    const float INF = __builtin_inff();

    float pa[3] = {3.0e18f, 3.0e18f, 3.0e18f};
    float fa[5] = {0, 0, 0, 0, 0};
    float blo[4] = {INF, INF, INF, INF}, bhi[4] = {-INF, -INF, -INF, -INF};
    const bool valid = i < D.na;
    if (valid) {
        const float4 lo = ld4s(D.a + lo_off(i)), hi = ld4s(D.a + hi_off(D.na, i));
        if (D.use_tran) apply_transform(D.use_tran == 2 ? D.from->transform : D.tran, lo.x, lo.y, lo.z, pa[0], pa[1], pa[2]);
        else { pa[0] = lo.x; pa[1] = lo.y; pa[2] = lo.z; }
        fa[0] = lo.w; fa[1] = hi.x; fa[2] = hi.y; fa[3] = hi.z; fa[4] = hi.w;
#pragma unroll
        for (int q = 0; q < 3; ++q) blo[q] = bhi[q] = pa[q];
        if (pa[2] > 1.0e-3f) { blo[3] = bhi[3] = pa[1] / pa[2]; } else { blo[3] = -INF; bhi[3] = INF; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { blo[q] = wmin(blo[q]); bhi[q] = wmax(bhi[q]); }
    // points p (a row), q (a column) within Rb of each other: |y_p/z_p - y_q/z_q| <= Rb (1 + |y_q/z_q|) / z_p
    const float slope_reach = (blo[2] > 1.0e-3f) ? Rb * 1.01f / blo[2] : INF;
    const float nthr = -thr_cull;

    double sumA = 0; int count = 0;
    float H[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) H[q] = 0.f;

    // this workgroup's chunk of cloud b, in 32-point groups
    const int ngroups = D.nbox;
    const int gper = (ngroups + (int)gridDim.y - 1) / (int)gridDim.y;
    const int g_begin = min(ngroups, (int)blockIdx.y * gper), g_end = min(ngroups, g_begin + gper);
    const bool any_row = blockIdx.x * SCORE_BLOCK < D.na;
    for (int gb = g_begin; any_row && gb < g_end; gb += 64) {
        bool near = false;
        if (gb + tid < g_end) {
            float gap2 = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float glo = D.bbox[q * ngroups + gb + tid], ghi = D.bbox[(4 + q) * ngroups + gb + tid];
                const float d = fmaxf(0.f, fmaxf(glo - bhi[q], blo[q] - ghi));
                gap2 = __builtin_fmaf(d, d, gap2);
            }
            const float tlo = D.bbox[3 * ngroups + gb + tid], thi = D.bbox[7 * ngroups + gb + tid];
            const float tgap = fmaxf(0.f, fmaxf(tlo - bhi[3], blo[3] - thi));
            const float tabs = fmaxf(fabsf(tlo), fabsf(thi));
            near = (gap2 <= thr_box) && (tgap <= slope_reach * (1.0f + tabs) + 1.0e-6f);   // false for NaN (inf - inf)
        }
        unsigned long long mask = __ballot(near);
        while (mask) {
            // up to SCORE_STAGE near groups are fetched together (their loads overlap), then swept one after the other
            int gis[SCORE_STAGE]; int ns = 0;
#pragma unroll
            for (int k = 0; k < SCORE_STAGE; ++k) {
                gis[k] = -1;
                if (mask) { gis[k] = gb + __builtin_ctzll(mask); mask &= mask - 1ull; ns = k + 1; }
            }
            __syncthreads();                                        // the previous groups have been swept
#pragma unroll
            for (int pass = 0; pass < SCORE_STAGE / 2; ++pass) {
                const int gsel = (tid >> 5) ? gis[2 * pass + 1] : gis[2 * pass];
                if (gsel >= 0) {
                    const int j = gsel * 32 + (tid & 31);
                    float b0 = -3.0e18f, b1 = -3.0e18f, b2 = -3.0e18f;
                    if (j < D.nb) { const float4 lo = ld4s(D.b + lo_off(j)); b0 = lo.x; b1 = lo.y; b2 = lo.z; }
                    lx[pass * 64 + tid] = b0; ly[pass * 64 + tid] = b1; lz[pass * 64 + tid] = b2;
                }
            }
            __syncthreads();
            for (int k = 0; k < ns; ++k) {
            int gi = gis[0];
#pragma unroll
            for (int k2 = 1; k2 < SCORE_STAGE; ++k2) gi = (k == k2) ? gis[k2] : gi;
            uint32_t w = 0u;
            const float4* qx = reinterpret_cast<const float4*>(lx + k * 32);
            const float4* qy = reinterpret_cast<const float4*>(ly + k * 32);
            const float4* qz = reinterpret_cast<const float4*>(lz + k * 32);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 X = qx[q], Y = qy[q], Z = qz[q];
                const float cx[4] = {X.x, X.y, X.z, X.w}, cy[4] = {Y.x, Y.y, Y.z, Y.w}, cz[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float dx = pa[0] - cx[u], dy = pa[1] - cy[u], dz = pa[2] - cz[u];
                    const float t = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, nthr)));
                    w = __builtin_amdgcn_alignbit(w, __float_as_uint(t), 31);     // sign bit: inside the (slightly widened) radius
                }
            }
            while (w) {                                             // bit 31 = first column of the group: ascending columns
                const int kbit = __clz(w);
                w &= ~(0x80000000u >> kbit);
                const int j = gi * 32 + kbit;
                const float pb[3] = {lx[k * 32 + kbit], ly[k * 32 + kbit], lz[k * 32 + kbit]};
                const float e0 = pa[0] - pb[0], e1 = pa[1] - pb[1], e2 = pa[2] - pb[2];
                float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;            // nanoflann.hpp:403-406
                if (!(d2 < d2_thres)) continue;                                      // cvo.cpp:423 / 654
                const float4 blo4 = ld4s(D.b + lo_off(j)), bhi4 = ld4s(D.b + hi_off(D.nb, j));
                const float fb[5] = {blo4.w, bhi4.x, bhi4.y, bhi4.z, bhi4.w};
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
                    float Bq[21];
                    // block A (symmetric): 00 01 02 11 12 22                          cvo.cpp:670-675
                    Bq[0] = il2 * cr[0] * cr[0] - dot1;
                    Bq[1] = (float)(il2 * cr[0] * cr[1] + 0.5 * (pa[0] * pb[1] + pa[1] * pb[0]));
                    Bq[2] = (float)(il2 * cr[0] * cr[2] + 0.5 * (pa[0] * pb[2] + pa[2] * pb[0]));
                    Bq[3] = il2 * cr[1] * cr[1] - dot2;
                    Bq[4] = (float)(il2 * cr[1] * cr[2] + 0.5 * (pa[1] * pb[2] + pa[2] * pb[1]));
                    Bq[5] = il2 * cr[2] * cr[2] - dot3;
                    // block C (full 3x3, row-major C(r,c))                            cvo.cpp:680-688
                    Bq[6] = il2 * cr[0] * db[0];          Bq[7] = -pa[2] + il2 * db[0] * cr[1];  Bq[8] = pa[1] + il2 * db[0] * cr[2];
                    Bq[9] = pa[2] + il2 * db[1] * cr[0];  Bq[10] = il2 * cr[1] * db[1];          Bq[11] = -pa[0] + il2 * db[1] * cr[2];
                    Bq[12] = -pa[1] + il2 * db[2] * cr[0]; Bq[13] = pa[0] + il2 * db[2] * cr[1]; Bq[14] = il2 * cr[2] * db[2];
                    // block D (symmetric): 00 01 02 11 12 22                          cvo.cpp:692-697
                    Bq[15] = il2 * db[0] * db[0] - 1; Bq[16] = il2 * db[0] * db[1]; Bq[17] = il2 * db[0] * db[2];
                    Bq[18] = il2 * db[1] * db[1] - 1; Bq[19] = il2 * db[1] * db[2]; Bq[20] = il2 * db[2] * db[2] - 1;
                    const float wgt = il2 * cdot * k;                                // cvo.cpp:707
#pragma unroll
                    for (int q2 = 0; q2 < 21; ++q2) H[q2] += wgt * Bq[q2];
                    count += 1;
                }
            }
            }
        }
    }

    const size_t rec = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    double cnt = (double)count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (!D.want_hessian) {                                          // inner product: two sums
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sumA += __shfl_xor(sumA, off, 64);
        if (tid < SCORE_NOUT) partials[rec * SCORE_NOUT + tid] = tid == 0 ? sumA : (tid == 1 ? cnt : 0.0);   // row blocks past the end of a request's cloud write zeros
        return;
    }
    double mine = tid == 1 ? cnt : 0.0;
#pragma unroll
    for (int q = 0; q < 21; ++q) {
        double t = (double)H[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        mine = (tid == 2 + q) ? t : mine;
    }
    if (tid < SCORE_NOUT) partials[rec * SCORE_NOUT + tid] = mine;
}

// one workgroup per request: its partial records into out[request][24] (pinned host memory).  Eight lanes per output walk
// the records r = part, part + 8, ... in order, then the eight partial sums are added in lane order: a fixed order, so the
// result does not depend on timing.
__global__ __launch_bounds__(256) void cvo_score_reduce_kernel(ScoreBatch B, const ScoreDesc* __restrict__ more, const double* __restrict__ partials, int records_per_request, double* __restrict__ out) {
    __shared__ double part_sum[8][32];
    __shared__ double totals[SCORE_NOUT];
    const int tid = threadIdx.x, q = tid & 31, part = tid >> 5;
    const double* p = partials + (size_t)blockIdx.x * records_per_request * SCORE_NOUT;
    double s = 0;
    if (q < SCORE_NOUT) for (int r = part; r < records_per_request; r += 8) s += p[(size_t)r * SCORE_NOUT + q];
    part_sum[part][q] = s;
    __syncthreads();
    if (tid < SCORE_NOUT) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part_sum[k][tid];
        out[blockIdx.x * SCORE_NOUT + tid] = t;
        totals[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {                                                 // a self inner product that was computed: keep it with the cloud
        const ScoreDesc& D = more ? more[blockIdx.x] : B.d[blockIdx.x];
        if (D.self_cache) {
            const float ell = D.from ? D.from->ell : D.ell;
            int at = -1, free_at = -1;
            for (int e = 0; e < SELF_CACHE_N; ++e) {
                if (D.self_cache[e].valid && D.self_cache[e].ell == ell) at = e;
                if (!D.self_cache[e].valid && free_at < 0) free_at = e;
            }
            if (at < 0) {
                SelfCacheEntry ne; ne.ell = ell; ne.valid = 1; ne.sum = totals[0]; ne.count = totals[1];
                D.self_cache[free_at >= 0 ? free_at : SELF_CACHE_N - 1] = ne;
            }
        }
    }
}

int score_nout() { return SCORE_NOUT; }
int score_groups(int n) { return (n + 31) / 32; }
static size_t score_box_floats(int n) { return (8 * (size_t)score_groups(n) + 3) & ~(size_t)3; }   // the table behind the boxes starts 16-byte aligned
size_t score_box_bytes(int n) { return sizeof(float) * score_box_floats(n) + sizeof(SelfCacheEntry) * SELF_CACHE_N; }
SelfCacheEntry* score_self_cache(float* gbox, int n) { return reinterpret_cast<SelfCacheEntry*>(gbox + score_box_floats(n)); }
hipError_t launch_cloud_boxes(const float* rec, int n, float* gbox, hipStream_t stream) {
    const int ng = score_groups(n);
    hipLaunchKernelGGL(cvo_cloud_boxes_kernel, dim3((ng + 1) / 2), dim3(64), 0, stream, rec, n, gbox, ng, score_self_cache(gbox, n));
    return hipGetLastError();
}
int score_row_blocks(int na) { return (na + SCORE_BLOCK - 1) / SCORE_BLOCK; }

// one launch for the whole batch of requests; out_pinned[request][24] is complete when the stream has drained
hipError_t launch_score(const ScoreBatch& B, const ScoreDesc* more, int nreq, int row_blocks, int chunks, const DevParams& P, double* partials,
                        double* out_pinned, hipStream_t stream, unsigned* wgs_started, bool* sweep_submitted) {
    if (sweep_submitted) *sweep_submitted = false;
    hipLaunchKernelGGL(cvo_score_kernel, dim3(row_blocks, chunks, nreq), dim3(SCORE_BLOCK), 0, stream, B, more, P, partials, wgs_started);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (sweep_submitted) *sweep_submitted = true;                   // its workgroups will count themselves as started
    hipLaunchKernelGGL(cvo_score_reduce_kernel, dim3(nreq), dim3(256), 0, stream, B, more, partials, row_blocks * chunks, out_pinned);
    return hipGetLastError();
}

}  // namespace cvohip
