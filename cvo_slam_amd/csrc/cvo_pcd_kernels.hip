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
// direction table is read but unused, setting_selectDirectionDistribution = false), so a
// thread owns one 4pot x 4pot block.  The cloud is written straight into the two float4
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

// ---- select (PixelSelector2.cpp:286-433): one thread walks one 4pot x 4pot block in the reference's order.
// map: 0 / 1 / 2 / 4 per pixel (pre-zeroed); counts[0..2] += n2, n3, n4.
__global__ __launch_bounds__(64) void pcd_select_kernel(const float* __restrict__ abs0, const float* __restrict__ abs1, const float* __restrict__ abs2,
                                                        const float* __restrict__ ths_smoothed, int w, int h, int pot, uint8_t* __restrict__ map,
                                                        int* __restrict__ counts) {
    const int nbx = (w + 4 * pot - 1) / (4 * pot), nby = (h + 4 * pot - 1) / (4 * pot);
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int n2 = 0, n3 = 0, n4 = 0;
    if (b < nbx * nby) {
        const int x4 = (b % nbx) * 4 * pot, y4 = (b / nbx) * 4 * pot;
        const int w1 = w / 2, w2 = w / 4, w32 = w / 32;
        const float dw1 = 0.75f, dw2 = dw1 * dw1;                     // setting_gradDownweightPerLevel
        const int my3 = min(4 * pot, h - y4), mx3 = min(4 * pot, w - x4);
        int best4 = -1; float val4 = 0.f;
        for (int y3 = 0; y3 < my3; y3 += 2 * pot) for (int x3 = 0; x3 < mx3; x3 += 2 * pot) {
            const int x34 = x3 + x4, y34 = y3 + y4;
            const int my2 = min(2 * pot, h - y34), mx2 = min(2 * pot, w - x34);
            int best3 = -1; float val3 = 0.f;
            for (int y2 = 0; y2 < my2; y2 += pot) for (int x2 = 0; x2 < mx2; x2 += pot) {
                const int x234 = x2 + x34, y234 = y2 + y34;
                const int my1 = min(pot, h - y234), mx1 = min(pot, w - x234);
                int best2 = -1; float val2 = 0.f;
                for (int y1 = 0; y1 < my1; ++y1) for (int x1 = 0; x1 < mx1; ++x1) {
                    const int xf = x1 + x234, yf = y1 + y234, idx = xf + w * yf;
                    if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
                    const float th0 = ths_smoothed[(xf >> 5) + (yf >> 5) * w32];   // rows past h/32 read the zeroed slack, like the reference
                    const float th1 = th0 * dw1, th2 = th1 * dw2;
                    const float ag0 = abs0[idx];
                    if (ag0 > th0 && ag0 > val2) { val2 = ag0; best2 = idx; best3 = -2; best4 = -2; }
                    if (best3 == -2) continue;
                    const float ag1 = abs1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1];
                    if (ag1 > th1 && ag1 > val3) { val3 = ag1; best3 = idx; best4 = -2; }
                    if (best4 == -2) continue;
                    const float ag2 = abs2[(int)(xf * 0.25f + 0.125f) + (int)(yf * 0.25f + 0.125f) * w2];
                    if (ag2 > th2 && ag2 > val4) { val4 = ag2; best4 = idx; }
                }
                if (best2 > 0) { map[best2] = 1; val3 = 1e10f; ++n2; }
            }
            if (best3 > 0) { map[best3] = 2; val4 = 1e10f; ++n3; }
        }
        if (best4 > 0) { map[best4] = 4; ++n4; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { n2 += __shfl_xor(n2, off, 64); n3 += __shfl_xor(n3, off, 64); n4 += __shfl_xor(n4, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[0], n2); atomicAdd(&counts[1], n3); atomicAdd(&counts[2], n4); }
}

// ---- makeMaps sub-sampling (PixelSelector2.cpp:252-268) + get_points_from_pixels' filter (pcd_generator.cpp:471): ONE workgroup of
// 1024 threads scans the image in order.  Pass `write` = 0 only counts (result[0] = points with valid depth, result[1] = pixels left
// in the map); pass 1 also writes the cloud planes, the selected pixels and clears dropped map entries.
constexpr int PCD_SCAN_THREADS = 1024;
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* lds, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < PCD_SCAN_THREADS / 64; ++k) { const int s = lds[k]; if (k < wave) base += s; tot += s; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

struct PcdCam { float scaling_factor, fx, fy, cx, cy; };

__global__ __launch_bounds__(PCD_SCAN_THREADS) void pcd_compact_kernel(uint8_t* __restrict__ map, const uint8_t* __restrict__ pattern, int subsample, int char_th,
                                                                       const uint16_t* __restrict__ depth, const uint8_t* __restrict__ bgr,
                                                                       const float* __restrict__ dx0, const float* __restrict__ dy0, int w, int h, PcdCam cam,
                                                                       int write, int n_points, float* __restrict__ cloud, uint16_t* __restrict__ px,
                                                                       int* __restrict__ result) {
    __shared__ int lds[PCD_SCAN_THREADS / 64];
    const int tid = threadIdx.x, n = w * h;
    const int per = (n + PCD_SCAN_THREADS - 1) / PCD_SCAN_THREADS;
    const int i0 = min(n, tid * per), i1 = min(n, i0 + per);
    int marked = 0;
    for (int i = i0; i < i1; ++i) marked += map[i] != 0;
    int total_marked = 0;
    int rn = block_exclusive_scan_1024(marked, lds, total_marked);      // position in the random pattern of this thread's first marked pixel
    int kept_valid = 0, kept = 0;
    for (int i = i0; i < i1; ++i) {
        if (map[i] == 0) continue;
        const bool keep = !(subsample && pattern[rn] > char_th);
        ++rn;
        if (keep) { ++kept; kept_valid += depth[i] != 0; }
    }
    int total_valid = 0, total_kept = 0;
    int at = block_exclusive_scan_1024(kept_valid, lds, total_valid);
    (void)block_exclusive_scan_1024(kept, lds, total_kept);
    if (tid == 0) { result[0] = total_valid; result[1] = total_kept; result[2] = total_marked; }
    if (!write) return;
    rn -= marked;                                                       // back to this thread's first marked pixel
    for (int i = i0; i < i1; ++i) {
        if (map[i] == 0) continue;
        const bool keep = !(subsample && pattern[rn] > char_th);
        ++rn;
        if (!keep) { map[i] = 0; continue; }
        const int dep = depth[i];
        if (dep == 0) continue;
        if (at < n_points) {
            const int x = i % w, y = i / w;
            const float p2 = (float)dep / cam.scaling_factor;           // pcd_generator.cpp:473-476
            const float p0 = ((float)x - cam.cx) * p2 / cam.fx;
            const float p1 = ((float)y - cam.cy) * p2 / cam.fy;
            float* lo = cloud + lo_off(at); float* hi = cloud + hi_off(n_points, at);
            lo[0] = p0; lo[1] = p1; lo[2] = p2; lo[3] = (float)bgr[3 * (size_t)i];               // B  (:601)
            hi[0] = (float)bgr[3 * (size_t)i + 1]; hi[1] = (float)bgr[3 * (size_t)i + 2];        // G, R
            hi[2] = dx0[i]; hi[3] = dy0[i];                                                      // :608-609
            px[2 * at] = (uint16_t)x; px[2 * at + 1] = (uint16_t)y;
        }
        ++at;
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
    hipLaunchKernelGGL(pcd_select_kernel, dim3((nb + 63) / 64), dim3(64), 0, s, abs0, abs1, abs2, ths_smoothed, w, h, pot, map, counts);
    return hipGetLastError();
}
hipError_t pcd_launch_compact(uint8_t* map, const uint8_t* pattern, int subsample, int char_th, const uint16_t* depth, const uint8_t* bgr, const float* dx0,
                              const float* dy0, int w, int h, const float cam[5], int write, int n_points, float* cloud, uint16_t* px, int* result, hipStream_t s) {
    PcdCam c{cam[0], cam[1], cam[2], cam[3], cam[4]};
    hipLaunchKernelGGL(pcd_compact_kernel, dim3(1), dim3(PCD_SCAN_THREADS), 0, s, map, pattern, subsample, char_th, depth, bgr, dx0, dy0, w, h, c, write, n_points,
                       cloud, px, result);
    return hipGetLastError();
}
hipError_t pcd_launch_unpack(const float* cloud, int n, float* xyz, float* feat, hipStream_t s) {
    if (n > 0) PCD_LAUNCH_1D(pcd_unpack_kernel, n, s, cloud, n, xyz, feat);
    return hipGetLastError();
}

}  // namespace cvohip
