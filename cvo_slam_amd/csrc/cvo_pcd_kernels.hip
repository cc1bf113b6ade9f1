// cvo_pcd_kernels.hip -- the point-cloud generator in front of the alignment
// (SURVEY.md 8f next-1): what cvo::set_pcd (cvo.cpp:345-386) runs on an RGB-D frame.
//   gray image            cv::cvtColor(COLOR_RGB2GRAY) on the BGR bytes, pcd_generator.cpp:624
//   3-level pyramid       make_pyramid, pcd_generator.cpp:50-143
//   block thresholds      PixelSelector::makeHists, PixelSelector2.cpp:71-134
//   hierarchical select   PixelSelector::select, PixelSelector2.cpp:286-433
//   sub-sampling          PixelSelector::makeMaps, PixelSelector2.cpp:252-268
//   cloud                 get_points_from_pixels + get_features(type 1), pcd_generator.cpp:456-499, 590-612
// Image-sized, memory-bound work: every kernel is one pass over w*h (or fewer) elements with
// coalesced accesses; the selection has no cross-block dependence (the reference's random
// direction table is read but unused, setting_selectDirectionDistribution = false), so 16
// lanes share one 4pot x 4pot block.  The cloud is written straight into the two float4
// planes the alignment kernels read: no host round trip for the points.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvo_device.h"

namespace cvohip {

// ---- load_image: 8-bit gray, OpenCV's fixed-point weights on the first/second/third byte (pcd_generator.cpp:624)
__global__ void pcd_gray_kernel(const uint8_t* __restrict__ bgr, float* __restrict__ I0, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c0 = bgr[3 * (size_t)i], c1 = bgr[3 * (size_t)i + 1], c2 = bgr[3 * (size_t)i + 2];
    I0[i] = (float)((c0 * 4899 + c1 * 9617 + c2 * 1868 + (1 << 13)) >> 14);
}

// ---- make_pyramid: 2x2 box down-sampling (pcd_generator.cpp:103-118)
__global__ void pcd_down_kernel(const float* __restrict__ P, int pw, float* __restrict__ I, int wl, int hl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wl * hl) return;
    const int x = i % wl, y = i / wl;
    const float* p = P + (size_t)2 * x + (size_t)2 * y * pw;
    I[i] = 0.25f * (p[0] + p[1] + p[pw] + p[pw + 1]);
}

// ---- make_pyramid: central differences over the FLAT index range [wl, wl*(hl-1)) (pcd_generator.cpp:122-136):
// the first and last column use the neighbouring row's pixel, exactly like the reference
__global__ void pcd_grad_kernel(const float* __restrict__ I, int wl, int hl, float* __restrict__ dx_out, float* __restrict__ dy_out,
                                float* __restrict__ abs2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= wl * hl) return;
    float dx = 0.f, dy = 0.f, a = 0.f;
    if (idx >= wl && idx < wl * (hl - 1)) {
        dx = 0.5f * (I[idx + 1] - I[idx - 1]);
        dy = 0.5f * (I[idx + wl] - I[idx - wl]);
        if (!__builtin_isfinite(dx)) dx = 0.f;
        if (!__builtin_isfinite(dy)) dy = 0.f;
        a = dx * dx + dy * dy;
    }
    if (dx_out) { dx_out[idx] = dx; dy_out[idx] = dy; }
    abs2[idx] = a;
}

// ---- makeHists: one workgroup per 32x32 block: histogram of int(sqrt(|grad|^2)) capped at 48, median + 7 (PixelSelector2.cpp:83-103)
__global__ __launch_bounds__(256) void pcd_hist_kernel(const float* __restrict__ abs0, int w, int h, int w32, float* __restrict__ ths) {
    __shared__ int hist[100];
    const int tid = threadIdx.x, bx = blockIdx.x % w32, by = blockIdx.x / w32;
    if (tid < 100) hist[tid] = 0;
    __syncthreads();
    for (int k = tid; k < 1024; k += 256) {
        const int it = (k & 31) + 32 * bx, jt = (k >> 5) + 32 * by;
        if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
        int g = (int)sqrtf(abs0[(size_t)it + (size_t)jt * w]);
        if (g > 48) g = 48;
        atomicAdd(&hist[g + 1], 1); atomicAdd(&hist[0], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int th = (int)(hist[0] * 0.5f + 0.5f), q = 90;               // computeHistQuantil(hist, setting_minGradHistCut), :59-68
        for (int i = 0; i < 90; ++i) { th -= hist[i + 1]; if (th < 0) { q = i; break; } }
        ths[blockIdx.x] = (float)(q + 7);                             // + setting_minGradHistAdd
    }
}

// ---- makeHists: squared 3x3 box mean of the block thresholds (PixelSelector2.cpp:105-132; sums of small integers: exact in any order)
__global__ void pcd_smooth_kernel(const float* __restrict__ ths, int w32, int h32, float* __restrict__ ths_smoothed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w32 * h32) return;
    const int x = i % w32, y = i / w32;
    float sum = 0.f, num = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx, yy = y + dy;
            if (xx < 0 || xx >= w32 || yy < 0 || yy >= h32) continue;
            num += 1.f; sum += ths[xx + yy * w32];
        }
    ths_smoothed[i] = (sum / num) * (sum / num);
}

// ---- select (PixelSelector2.cpp:286-433).  16 lanes share one 4pot x 4pot block, one lane per pot x pot cell, cells numbered
// in the reference's traversal order (2pot block by 2pot block).  With setting_selectDirectionDistribution = false the walk
// reduces to three order-free rules, each a "first maximum in traversal order":
//   a cell picks its largest |grad|^2 above the level-0 threshold (map = 1);
//   a 2pot block none of whose pixels passes level 0 picks its largest level-1 value above the level-1 threshold (map = 2);
//   a 4pot block none of whose pixels passes level 0 or level 1 picks its largest level-2 value above its threshold (map = 4).
// (In the reference a level-0 hit sets bestIdx3 = bestIdx4 = -2 for good and a level-1 hit sets bestIdx4 = -2 for good: whatever
// was found at the coarser levels before is dropped, and nothing is looked for after.)
// map: 0 / 1 / 2 / 4 per pixel (pre-zeroed); counts[0..2] += n2, n3, n4.
__global__ __launch_bounds__(256) void pcd_select_kernel(const float* __restrict__ abs0, const float* __restrict__ abs1, const float* __restrict__ abs2,
                                                         const float* __restrict__ ths_smoothed, int w, int h, int pot, uint8_t* __restrict__ map,
                                                         int* __restrict__ counts) {
    const int nbx = (w + 4 * pot - 1) / (4 * pot), nby = (h + 4 * pot - 1) / (4 * pot);
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gt >> 4, sub = gt & 15, lane = threadIdx.x & 63;
    const int b3 = sub >> 2, c2 = sub & 3;                            // 2pot block inside the 4pot block, cell inside the 2pot block
    int n2 = 0, n3 = 0, n4 = 0;
    bool q0 = false, q1 = false, q2 = false;
    int best0 = -1, best1 = -1, best2 = -1; float val0 = 0.f, val1 = 0.f, val2 = 0.f;
    if (b < nbx * nby) {
        const int x4 = (b % nbx) * 4 * pot, y4 = (b / nbx) * 4 * pot;
        const int x0 = x4 + (b3 & 1) * 2 * pot + (c2 & 1) * pot, y0 = y4 + (b3 >> 1) * 2 * pot + (c2 >> 1) * pot;
        const int w1 = w / 2, w2 = w / 4, w32 = w / 32;
        const float dw1 = 0.75f, dw2 = dw1 * dw1;                     // setting_gradDownweightPerLevel
        const int my1 = min(pot, h - y0), mx1 = min(pot, w - x0);
        for (int y1 = 0; y1 < my1; ++y1) for (int x1 = 0; x1 < mx1; ++x1) {
            const int xf = x1 + x0, yf = y1 + y0, idx = xf + w * yf;
            if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
            const float th0 = ths_smoothed[(xf >> 5) + (yf >> 5) * w32];   // rows past h/32 read the zeroed slack, like the reference
            const float th1 = th0 * dw1, th2 = th1 * dw2;
            const float ag0 = abs0[idx];
            const float ag1 = abs1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1];
            const float ag2 = abs2[(int)(xf * 0.25f + 0.125f) + (int)(yf * 0.25f + 0.125f) * w2];
            if (ag0 > th0) { q0 = true; if (ag0 > val0) { val0 = ag0; best0 = idx; } }
            if (ag1 > th1) { q1 = true; if (ag1 > val1) { val1 = ag1; best1 = idx; } }
            if (ag2 > th2) { q2 = true; if (ag2 > val2) { val2 = ag2; best2 = idx; } }
        }
        if (best0 > 0) { map[best0] = 1; ++n2; }
    }
    // level 1: the first lane of every group of 4 walks its group's cells in order
    const int g4 = lane & ~3, g16 = lane & ~15;
    bool any0_4 = false; int pick1 = -1; float pv1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool k0 = __shfl((int)q0, g4 + k, 64) != 0, k1 = __shfl((int)q1, g4 + k, 64) != 0;
        const float v = __shfl(val1, g4 + k, 64); const int ix = __shfl(best1, g4 + k, 64);
        any0_4 |= k0;
        if (k1 && v > pv1) { pv1 = v; pick1 = ix; }
    }
    if (c2 == 0 && !any0_4 && pick1 > 0) { map[pick1] = 2; ++n3; }
    // level 2: the first lane of every group of 16
    bool any01_16 = false; int pick2 = -1; float pv2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const bool k01 = (__shfl((int)q0, g16 + k, 64) | __shfl((int)q1, g16 + k, 64)) != 0, k2 = __shfl((int)q2, g16 + k, 64) != 0;
        const float v = __shfl(val2, g16 + k, 64); const int ix = __shfl(best2, g16 + k, 64);
        any01_16 |= k01;
        if (k2 && v > pv2) { pv2 = v; pick2 = ix; }
    }
    if (sub == 0 && !any01_16 && pick2 > 0) { map[pick2] = 4; ++n4; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { n2 += __shfl_xor(n2, off, 64); n3 += __shfl_xor(n3, off, 64); n4 += __shfl_xor(n4, off, 64); }
    if (lane == 0) { atomicAdd(&counts[0], n2); atomicAdd(&counts[1], n3); atomicAdd(&counts[2], n4); }
}

// ---- makeMaps sub-sampling (PixelSelector2.cpp:252-268) + get_points_from_pixels' filter (pcd_generator.cpp:471), order
// preserving and coalesced: a workgroup owns a tile of PCD_TILE consecutive pixels, its waves take 64 consecutive pixels at a
// time (ballot + popcount give every marked pixel its rank), tiles are chained by per-tile counts (a tile adds up the counts
// of the tiles before it: there are only w*h/4096 of them).
//   pass 1  marked pixels per tile
//   pass 2  rank among ALL marked pixels -> position in the random byte pattern -> dropped pixels leave the map;
//           per tile: pixels kept, and kept with a valid depth
//   pass 3  (the cloud is allocated by then) rank among the kept, valid pixels = index of the point: planes + pixel list
constexpr int PCD_TILE = 4096;
constexpr int PCD_TILE_THREADS = 256;

struct PcdCam { float scaling_factor, fx, fy, cx, cy; };

// exclusive prefix of `flag` over the workgroup's current 256 pixels, in pixel order; `run` carries on across chunks
__device__ __forceinline__ int chunk_rank(bool flag, int* wsum, int& run) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int before = run, all = 0;
#pragma unroll
    for (int k = 0; k < PCD_TILE_THREADS / 64; ++k) { const int c = wsum[k]; if (k < wave) before += c; all += c; }
    __syncthreads();
    run += all;
    return before + __popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ int tiles_before(const int* counts, int tile, int* lds) {
    int v = 0;
    for (int k = threadIdx.x; k < tile; k += PCD_TILE_THREADS) v += counts[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int k = 0; k < PCD_TILE_THREADS / 64; ++k) s += lds[k];
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(PCD_TILE_THREADS) void pcd_count_marked_kernel(const uint8_t* __restrict__ map, int n, int* __restrict__ tile_marked) {
    __shared__ int wsum[PCD_TILE_THREADS / 64];
    const int base = blockIdx.x * PCD_TILE;
    int cnt = 0;
    for (int k = threadIdx.x; k < PCD_TILE; k += PCD_TILE_THREADS) { const int i = base + k; cnt += (i < n && map[i] != 0) ? 1 : 0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int k = 0; k < PCD_TILE_THREADS / 64; ++k) s += wsum[k]; tile_marked[blockIdx.x] = s; }
}

__global__ __launch_bounds__(PCD_TILE_THREADS) void pcd_subsample_kernel(uint8_t* __restrict__ map, const uint8_t* __restrict__ pattern, int subsample, int char_th,
                                                                         const uint16_t* __restrict__ depth, int n, const int* __restrict__ tile_marked,
                                                                         int* __restrict__ tile_valid, int* __restrict__ tile_kept) {
    __shared__ int wsum[PCD_TILE_THREADS / 64];
    const int base = blockIdx.x * PCD_TILE;
    int run = tiles_before(tile_marked, blockIdx.x, wsum);            // marked pixels in front of this tile = index into the byte pattern
    int valid = 0, kept = 0;
    for (int k0 = 0; k0 < PCD_TILE; k0 += PCD_TILE_THREADS) {
        const int i = base + k0 + threadIdx.x;
        const bool marked = i < n && map[i] != 0;
        const int rn = chunk_rank(marked, wsum, run);
        if (marked) {
            const bool keep = !(subsample && pattern[rn] > char_th);  // PixelSelector2.cpp:261-265
            if (!keep) map[i] = 0;
            else { ++kept; valid += depth[i] != 0; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { valid += __shfl_xor(valid, off, 64); kept += __shfl_xor(kept, off, 64); }
    __shared__ int red[2][PCD_TILE_THREADS / 64];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = valid; red[1][threadIdx.x >> 6] = kept; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < PCD_TILE_THREADS / 64; ++k) { a += red[0][k]; b += red[1][k]; }
        tile_valid[blockIdx.x] = a; tile_kept[blockIdx.x] = b;
    }
}

__global__ __launch_bounds__(PCD_TILE_THREADS) void pcd_cloud_kernel(const uint8_t* __restrict__ map, const uint16_t* __restrict__ depth, const uint8_t* __restrict__ bgr,
                                                                     const float* __restrict__ dx0, const float* __restrict__ dy0, int w, int n, PcdCam cam,
                                                                     const int* __restrict__ tile_valid, int n_points, float* __restrict__ cloud,
                                                                     uint16_t* __restrict__ px) {
    __shared__ int wsum[PCD_TILE_THREADS / 64];
    const int base = blockIdx.x * PCD_TILE;
    int run = tiles_before(tile_valid, blockIdx.x, wsum);
    for (int k0 = 0; k0 < PCD_TILE; k0 += PCD_TILE_THREADS) {
        const int i = base + k0 + threadIdx.x;
        const int dep = i < n ? (int)depth[i] : 0;
        const bool ok = i < n && map[i] != 0 && dep != 0;             // pcd_generator.cpp:471
        const int at = chunk_rank(ok, wsum, run);
        if (ok && at < n_points) {
            const int x = i % w, y = i / w;
            const float p2 = (float)dep / cam.scaling_factor;         // pcd_generator.cpp:473-476
            const float p0 = ((float)x - cam.cx) * p2 / cam.fx;
            const float p1 = ((float)y - cam.cy) * p2 / cam.fy;
            float* lo = cloud + lo_off(at); float* hi = cloud + hi_off(n_points, at);
            lo[0] = p0; lo[1] = p1; lo[2] = p2; lo[3] = (float)bgr[3 * (size_t)i];               // B  (:601)
            hi[0] = (float)bgr[3 * (size_t)i + 1]; hi[1] = (float)bgr[3 * (size_t)i + 2];        // G, R
            hi[2] = dx0[i]; hi[3] = dy0[i];                                                      // :608-609
            px[2 * at] = (uint16_t)x; px[2 * at + 1] = (uint16_t)y;
        }
    }
}

// cloud planes back to the reference layout (tests, get_*_selected_points callers)
__global__ void pcd_unpack_kernel(const float* __restrict__ cloud, int n, float* __restrict__ xyz, float* __restrict__ feat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* lo = cloud + lo_off(i); const float* hi = cloud + hi_off(n, i);
    xyz[3 * (size_t)i] = lo[0]; xyz[3 * (size_t)i + 1] = lo[1]; xyz[3 * (size_t)i + 2] = lo[2];
    feat[i] = lo[3]; feat[(size_t)n + i] = hi[0]; feat[2 * (size_t)n + i] = hi[1]; feat[3 * (size_t)n + i] = hi[2]; feat[4 * (size_t)n + i] = hi[3];
}

// host clouds in the reference layout (n x 3 positions AoS, data_type.h:30; 5 channel-major feature arrays of n, data_type.h:75), copied
// to the device as they are, into the cloud's two float4 planes {x, y, z, f0}, {f1..f4}: one launch for all clouds of a hand-over
// (grid.y = cloud).  The positions of 256 consecutive points are 768 consecutive floats: fetched coalesced through LDS.
__global__ __launch_bounds__(256) void cvo_pack_clouds_kernel(const float* __restrict__ raw, const PackDesc* __restrict__ descs) {
    __shared__ float pos[768];
    const PackDesc D = descs[blockIdx.y];
    const int i0 = blockIdx.x * 256, tid = threadIdx.x;
    if (i0 >= D.n) return;
    const float* xyz = raw + D.raw_off; const float* feat = D.feat_off ? raw + D.feat_off : xyz + 3 * (size_t)D.n;
    const int cnt = min(256, D.n - i0);
    for (int k = tid; k < 3 * cnt; k += 256) pos[k] = xyz[3 * (size_t)i0 + k];
    __syncthreads();
    if (tid < cnt) {
        const int i = i0 + tid;
        float4 lo, hi;
        lo.x = pos[3 * tid]; lo.y = pos[3 * tid + 1]; lo.z = pos[3 * tid + 2]; lo.w = feat[i];
        hi.x = feat[(size_t)D.n + i]; hi.y = feat[2 * (size_t)D.n + i]; hi.z = feat[3 * (size_t)D.n + i]; hi.w = feat[4 * (size_t)D.n + i];
        *reinterpret_cast<float4*>(D.dst + lo_off(i)) = lo;
        *reinterpret_cast<float4*>(D.dst + hi_off(D.n, i)) = hi;
    }
}
hipError_t launch_pack_clouds(const float* raw, const PackDesc* descs, int n_clouds, int n_max, hipStream_t s) {
    if (n_clouds > 0 && n_max > 0) hipLaunchKernelGGL(cvo_pack_clouds_kernel, dim3((n_max + 255) / 256, n_clouds), dim3(256), 0, s, raw, descs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ host-side launchers
#define PCD_LAUNCH_1D(kernel, n, stream, ...) hipLaunchKernelGGL(kernel, dim3(((n) + 255) / 256), dim3(256), 0, stream, __VA_ARGS__)

hipError_t pcd_launch_pyramid(const uint8_t* bgr, int w, int h, float* I0, float* I1, float* I2, float* dx0, float* dy0, float* abs0, float* abs1, float* abs2,
                              hipStream_t s) {
    const int w1 = w / 2, h1 = h / 2, w2 = w1 / 2, h2 = h1 / 2;
    PCD_LAUNCH_1D(pcd_gray_kernel, w * h, s, bgr, I0, w * h);
    PCD_LAUNCH_1D(pcd_grad_kernel, w * h, s, I0, w, h, dx0, dy0, abs0);
    PCD_LAUNCH_1D(pcd_down_kernel, w1 * h1, s, I0, w, I1, w1, h1);
    PCD_LAUNCH_1D(pcd_grad_kernel, w1 * h1, s, I1, w1, h1, (float*)nullptr, (float*)nullptr, abs1);
    PCD_LAUNCH_1D(pcd_down_kernel, w2 * h2, s, I1, w1, I2, w2, h2);
    PCD_LAUNCH_1D(pcd_grad_kernel, w2 * h2, s, I2, w2, h2, (float*)nullptr, (float*)nullptr, abs2);
    return hipGetLastError();
}
hipError_t pcd_launch_thresholds(const float* abs0, int w, int h, float* ths, float* ths_smoothed, hipStream_t s) {
    const int w32 = w / 32, h32 = h / 32;
    if (w32 * h32 > 0) {
        hipLaunchKernelGGL(pcd_hist_kernel, dim3(w32 * h32), dim3(256), 0, s, abs0, w, h, w32, ths);
        PCD_LAUNCH_1D(pcd_smooth_kernel, w32 * h32, s, ths, w32, h32, ths_smoothed);
    }
    return hipGetLastError();
}
hipError_t pcd_launch_select(const float* abs0, const float* abs1, const float* abs2, const float* ths_smoothed, int w, int h, int pot, uint8_t* map, int* counts,
                             hipStream_t s) {
    const int nb = ((w + 4 * pot - 1) / (4 * pot)) * ((h + 4 * pot - 1) / (4 * pot));
    hipLaunchKernelGGL(pcd_select_kernel, dim3((nb * 16 + 255) / 256), dim3(256), 0, s, abs0, abs1, abs2, ths_smoothed, w, h, pot, map, counts);
    return hipGetLastError();
}
int pcd_tiles(int w, int h) { return (w * h + PCD_TILE - 1) / PCD_TILE; }
// tile_counts: 3 arrays of pcd_tiles() ints {marked, valid, kept}
hipError_t pcd_launch_subsample(uint8_t* map, const uint8_t* pattern, int subsample, int char_th, const uint16_t* depth, int w, int h, int* tile_counts, hipStream_t s) {
    const int nt = pcd_tiles(w, h);
    hipLaunchKernelGGL(pcd_count_marked_kernel, dim3(nt), dim3(PCD_TILE_THREADS), 0, s, map, w * h, tile_counts);
    hipLaunchKernelGGL(pcd_subsample_kernel, dim3(nt), dim3(PCD_TILE_THREADS), 0, s, map, pattern, subsample, char_th, depth, w * h, tile_counts, tile_counts + nt,
                       tile_counts + 2 * nt);
    return hipGetLastError();
}
hipError_t pcd_launch_cloud(const uint8_t* map, const uint16_t* depth, const uint8_t* bgr, const float* dx0, const float* dy0, int w, int h, const float cam[5],
                            const int* tile_counts, int n_points, float* cloud, uint16_t* px, hipStream_t s) {
    const int nt = pcd_tiles(w, h);
    PcdCam c{cam[0], cam[1], cam[2], cam[3], cam[4]};
    hipLaunchKernelGGL(pcd_cloud_kernel, dim3(nt), dim3(PCD_TILE_THREADS), 0, s, map, depth, bgr, dx0, dy0, w, w * h, c, tile_counts + nt, n_points, cloud, px);
    return hipGetLastError();
}
hipError_t pcd_launch_unpack(const float* cloud, int n, float* xyz, float* feat, hipStream_t s) {
    if (n > 0) PCD_LAUNCH_1D(pcd_unpack_kernel, n, s, cloud, n, xyz, feat);
    return hipGetLastError();
}

}  // namespace cvohip
