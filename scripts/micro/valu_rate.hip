// Microbenchmark: how many wave64 VALU instructions one SIMD of gfx950 issues per clock with 1, 2, 3, 4 waves resident.
// Decides what the "peak" of a VALU-issue roofline is.  build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ void __launch_bounds__(1024) stream_kernel(float* out, int iters, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const float m = seed * 0.5f + 1.0f, c = seed + 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 4) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
                if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2" : "+v"(a[i]) : "v"(c), "v"(m));   // dependent pair
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}


template <int KIND>
__global__ void __launch_bounds__(1024) stream64_kernel(double* out, int iters, double seed) {
    double a[12]; float f[12]; unsigned long long q[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { a[i] = seed + (double)(threadIdx.x + i); f[i] = (float)a[i]; q[i] = (unsigned long long)(threadIdx.x + i) * 77u; }
    const double m = seed * 0.5 + 1.0, c = seed + 0.25;
    const unsigned um = (unsigned)seed + 3u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (KIND == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[i]) : "v"(um));
                if (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(um), "v"(f[i]) : "vcc");
                if (KIND == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
                if (KIND == 8) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(um));
                if (KIND == 9) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
                if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(um));
                if (KIND == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(um) : "vcc");
                if (KIND == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 13) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 14) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(f[i]) : "v"(um));
                if (KIND == 15) asm volatile("v_sub_f32 %0, %0, %1 \n v_fma_f32 %0, %0, %0, %1" : "+v"(f[i]) : "v"(um));
                if (KIND == 16) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "s"(um));
                if (KIND == 17) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "s"(m));
                if (KIND == 18) asm volatile("v_mul_f64 %0, %0, 0.5" : "+v"(a[i]));
                if (KIND == 19) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(q[i]));
                if (KIND == 20) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[i]) : "v"(a[i]));
                if (KIND == 21) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(f[i]) : "s20");
                if (KIND == 22) asm volatile("v_cmp_lt_f32 vcc, %0, %1 \n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(um) : "vcc");
                if (KIND == 23) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i]) : "v"(um));
            }
        }
    }
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += a[i] + (double)f[i] + (double)q[i];
    if (s == 12345.678) out[threadIdx.x] = s;
}

// dependent chain: one accumulator, every instruction waits for the one before (what a single wave's latency costs)
__global__ void __launch_bounds__(1024) chain_kernel(float* out, int iters, float seed) {
    float a = seed + (float)threadIdx.x; const float c = seed + 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
    }
    if (a == 12345.678f) out[threadIdx.x] = a;
}

template <class F> static double time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;   // Hz
    printf("device %s, %d CUs, clockRate %.0f MHz\n", p.gcnArchName, cus, clk / 1e6);
    float* out; hipMalloc(&out, 4096);
    const int iters = 20000;
    const char* names[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_and_b32", "v_lshrrev_b32", "add+mul dependent pairs"};
    for (int kind = 0; kind < 6; ++kind)
        for (int waves = 1; waves <= 4; ++waves) {
            const int block = 256 * waves;
            auto go = [&] {
                switch (kind) {
                    case 0: stream_kernel<0><<<cus, block>>>(out, iters, 1.f); break;
                    case 1: stream_kernel<1><<<cus, block>>>(out, iters, 1.f); break;
                    case 2: stream_kernel<2><<<cus, block>>>(out, iters, 1.f); break;
                    case 3: stream_kernel<3><<<cus, block>>>(out, iters, 1.f); break;
                    case 4: stream_kernel<4><<<cus, block>>>(out, iters, 1.f); break;
                    default: stream_kernel<5><<<cus, block>>>(out, iters, 1.f); break;
                }
            };
            const double ms = time_ms(go);
            const double per_wave = (double)iters * 64 * (kind == 5 ? 2 : 1);
            const double instr = per_wave * waves * 4 * cus;      // wave-instructions of the launch
            printf("%-26s %d wave/SIMD: %8.3f ms  %.3e wave-instr/s  %.3f per SIMD per clock (at %.0f MHz)\n", names[kind], waves, ms, instr / (ms * 1e-3),
                   instr / (ms * 1e-3) / (4.0 * cus) / clk, clk / 1e6);
        }

    {
        const char* n64[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_mul_lo_u32", "v_mad_u64_u32", "v_rcp_f32", "v_ldexp_f64", "v_rndne_f64",
                             "v_cndmask_b32", "v_cmp_lt_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_alignbit_b32", "sub+fma dependent pairs", "v_mul_f32 v,s,v", "v_mul_f64 v,v,s", "v_mul_f64 v,v,0.5", "v_mov_b64", "v_lshl_add_u64", "v_readfirstlane_b32", "cmp+cndmask pairs", "v_add_u32"};
        double* o64; hipMalloc(&o64, 8192);
        const int it64 = 5000;
        for (int kind = 0; kind < 24; ++kind)
            for (int waves = 1; waves <= 4; waves *= 2) {
                const int block = 256 * waves;
                auto go = [&] {
                    switch (kind) {
#define K(n) case n: stream64_kernel<n><<<cus, block>>>(o64, it64, 1.0); break;
                        K(0) K(1) K(2) K(3) K(4) K(5) K(6) K(7) K(8) K(9) K(10) K(11) K(12) K(13) K(14) K(15) K(16) K(17) K(18) K(19) K(20) K(21) K(22) K(23)
#undef K
                    }
                };
                const double ms = time_ms(go);
                const double instr = (double)it64 * 48 * (kind == 15 || kind == 22 ? 2 : 1) * waves * 4 * cus;
                printf("%-26s %d wave/SIMD: %8.3f ms  %.3e wave-instr/s  cycles per wave-instr per SIMD at 2.4 GHz: %.2f\n", n64[kind], waves, ms, instr / (ms * 1e-3), 4.0 * cus * clk / (instr / (ms * 1e-3)));
            }
    }
    for (int waves = 1; waves <= 4; ++waves) {
        const int block = 256 * waves;
        const double ms = time_ms([&] { chain_kernel<<<cus, block>>>(out, iters, 1.f); });
        const double instr = (double)iters * 64 * waves * 4 * cus;
        printf("%-26s %d wave/SIMD: %8.3f ms  %.3e wave-instr/s  %.3f per SIMD per clock\n", "v_add_f32 dependent chain", waves, ms, instr / (ms * 1e-3), instr / (ms * 1e-3) / (4.0 * cus) / clk);
    }
    return 0;
}
