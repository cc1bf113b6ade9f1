// cvo_selftest.hip -- the scalar closed forms of the align kernel's epilogue, evaluated ON THE DEVICE for caller-supplied
// inputs (include/cvo_hip.h: cvo_selftest_*).  They are the very functions phase_epilogue calls (cvo_math.hpp: cubic_step =
// cvo.cpp:76-92,317-333; exp_sek3 = LieGroup.cpp:159-186; dist_se3 = cvo.cpp:94-104), compiled for gfx950, one lane per
// case, so a test can pin the device code against independent known answers (tests/golden/closed_forms.json: numpy roots,
// scipy expm / logm) without going through the oracle, whose source text the device functions share.
#include <hip/hip_runtime.h>
#include "cvo_math.hpp"

namespace cvohip {

__global__ void selftest_cubic_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cubic_step(in[i * 5 + 0], in[i * 5 + 1], in[i * 5 + 2], in[i * 5 + 3], in[i * 5 + 4]);
}
__global__ void selftest_exp_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float omega[3], v[3], dR[9], dT[3];
    for (int q = 0; q < 3; ++q) { omega[q] = in[i * 7 + q]; v[q] = in[i * 7 + 3 + q]; }
    exp_sek3(omega, v, in[i * 7 + 6], dR, dT);
    for (int q = 0; q < 9; ++q) out[i * 12 + q] = dR[q];
    for (int q = 0; q < 3; ++q) out[i * 12 + 9 + q] = dT[q];
}
__global__ void selftest_dist_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float dR[9], dT[3];
    for (int q = 0; q < 9; ++q) dR[q] = in[i * 12 + q];
    for (int q = 0; q < 3; ++q) dT[q] = in[i * 12 + 9 + q];
    out[i] = dist_se3(dR, dT);
}

// the device's float routines where the epilogue and the gates call them (OCML: sinf, cosf in exp_sek3 -- LieGroup.cpp:174-175 --, logf in the gates,
// cvo.cpp:125-126), element by element: out[6 i ..] = {sinf(x), cosf(x), logf(x), sin_f32_cr(x), cos_f32_cr(x), log_f32_cr(x)} -- the last three are what exp_sek3 and the gates call
__global__ void selftest_libm_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    out[i * 6 + 0] = sinf(x); out[i * 6 + 1] = cosf(x); out[i * 6 + 2] = logf(x); out[i * 6 + 3] = sin_f32_cr(x); out[i * 6 + 4] = cos_f32_cr(x); out[i * 6 + 5] = log_f32_cr(x);
}

hipError_t launch_selftest(int kind, const float* in, float* out, int n, hipStream_t s) {
    const dim3 grid((n + 63) / 64), block(64);
    if (kind == 0) hipLaunchKernelGGL(selftest_cubic_kernel, grid, block, 0, s, in, out, n);
    else if (kind == 1) hipLaunchKernelGGL(selftest_exp_kernel, grid, block, 0, s, in, out, n);
    else if (kind == 3) hipLaunchKernelGGL(selftest_libm_kernel, grid, block, 0, s, in, out, n);
    else hipLaunchKernelGGL(selftest_dist_kernel, grid, block, 0, s, in, out, n);
    return hipGetLastError();
}

}  // namespace cvohip
