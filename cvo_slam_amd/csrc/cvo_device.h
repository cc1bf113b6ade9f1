// cvo_device.h -- device-side data structures shared by the HIP kernels and the
// host-side C-ABI implementation (cvo_capi.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvohip {

// One point = 8 floats {x, y, z, f0, f1, f2, f3, f4}: the reference's position
// (data_type.h:30) and 5-channel feature row (data_type.h:75) packed to 32 B so a
// lane fetches a whole point with two 16-B loads and consecutive lanes stay
// coalesced.
constexpr int REC = 8;

struct DevParams {           // cvo.cpp:35-51
    float sigma, sp_thres, c, d, c_ell, c_sigma, min_step, eps, eps_2;
    int max_iter;
};

// per-pair state, read at kernel start and written back at the end (Q1, Q2)
struct PairState {
    float R[9];
    float T[3];
    float ell;
    float transform[12];      // in: cvo::transform before the call; out: final [R^T | -R^T T] (cvo.cpp:817)
    float prev_transform[12]; // out: transform of the last executed iteration (cvo.cpp:815)
    int iter;            // value of k at the break (stale if max_iter is hit, Q4)
    int A_nonzero;       // nnz of the last iteration (Q5)
    int iterations_run;
    int status;
    long long candidates_total;
};

struct TraceRow {        // == cvo_trace_row (include/cvo_hip.h)
    float omega[3];
    float v[3];
    int nnz;
    int candidates;
    double B, C, D, E;
    float step;
    float ell;
    float dist;
    int pad_;
};

// exchange area for the G workgroups that cooperate on one pair: two buffers
// (alternating by phase), G slots of XCH_WORDS 8-byte {tag, payload} granules.
constexpr int XCH_WORDS = 16;

struct PairDesc {
    const float* fixed;      // [nf][REC]
    const float* moving;     // [nm][REC]
    int nf, nm;
    int nf_pad;              // row stride of jlist/alist (multiple of 64)
    int cap;                 // candidate capacity per row
    float4* ybuf;            // [G][nm_pad]  transformed moving points {y0,y1,y2,g0}
    int nm_pad;
    uint16_t* jlist;         // [cap][nf_pad] candidate column indices, ascending per row
    float* alist;            // [cap][nf_pad] kernel value of each candidate (0 = not a survivor)
    int* cnt;                // [nf_pad]      candidates found per row (may exceed cap => dense fallback)
    unsigned long long* xch; // [2][G][XCH_WORDS]
    PairState* state;
    TraceRow* trace;         // optional
    int trace_cap;
    int* trace_len;
};

// score kernels (function_inner_product / se3_Hessian)
struct ScoreDesc {
    const float* a;          // [na][REC] queried cloud (positions optionally transformed by tran)
    const float* b;          // [nb][REC] searched cloud
    int na, nb;
    float tran[12];
    int use_tran;
    float ell;
    int want_hessian;        // 0: inner product only, 1: Hessian terms too
    double* out;             // [24]: 0 = sum_A, 1 = count, 2..22 = 21 upper-triangle Hessian terms (f64 of f32 row sums)
};

}  // namespace cvohip
