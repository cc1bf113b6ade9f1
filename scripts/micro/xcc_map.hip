// development aid: which XCD does block p of a launch land on?  Eight streams (GPU_MAX_HW_QUEUES=8), 64 blocks of 512 threads with 140 KB of LDS each (one per CU, like the
// align kernel), every block spins for a time that depends on its index (8 duration classes), launches queued back to back like the bench does.  Each block records its XCC_ID.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/micro/xcc_map.hip -o /tmp/xcc_map && GPU_MAX_HW_QUEUES=8 /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void spin(int* out, int launch, unsigned base_ticks) {
    extern __shared__ char lds[];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[launch * 64 + blockIdx.x] = (int)(xcc & 0xF); lds[0] = 1; }
    const unsigned cls = blockIdx.x % 8;                               // class 0 spins longest
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), dt = (unsigned long long)base_ticks * (8 - cls);
    while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(16);
}
int main() {
    const int S = 8, L = 12;
    hipStream_t st[S]; int* out[S];
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int s = 0; s < S; ++s) { hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking); hipMalloc(&out[s], sizeof(int) * 64 * L); hipMemset(out[s], 0xFF, sizeof(int) * 64 * L); }
    hipDeviceSynchronize();
    for (int l = 0; l < L; ++l) for (int s = 0; s < S; ++s) hipLaunchKernelGGL(spin, dim3(64), dim3(512), 140 * 1024, st[s], out[s], l, 10000u /* 100 us per class step */);
    hipDeviceSynchronize();
    for (int s = 0; s < S; ++s) {
        std::vector<int> h(64 * L); hipMemcpy(h.data(), out[s], sizeof(int) * 64 * L, hipMemcpyDeviceToHost);
        printf("stream %d:", s);
        for (int l = 0; l < L; ++l) {
            int ok = 1; for (int p = 0; p < 64; ++p) if (h[l * 64 + p] != (h[l * 64] + p) % 8) ok = 0;
            printf(" L%d b0->X%d%s", l, h[l * 64], ok ? "" : "*");
        }
        printf("\n   launch 0 blocks 0..15:"); for (int p = 0; p < 16; ++p) printf(" %d", h[p]);
        printf("\n   last launch blocks 0..15:"); for (int p = 0; p < 16; ++p) printf(" %d", h[(L - 1) * 64 + p]); printf("\n");
    }
    printf("('*' = the launch's blocks are NOT at XCD (block0 + p) mod 8)\n");
    return 0;
}
