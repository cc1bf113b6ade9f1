// cvo_device.h -- device-side data structures shared by the HIP kernels and the
// host-side C-ABI implementation (cvo_capi.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvohip {

// A cloud of n points in HBM is two planes of n float4: {x, y, z, f0} -- the reference's position
// (data_type.h:30) and the first channel of its 5-channel feature row (data_type.h:75), all that the
// transform, the cull and the geometric half of the kernel value touch -- then {f1, f2, f3, f4}.
// Consecutive lanes fetch consecutive 16-byte records of one plane.
constexpr int REC = 8;                   // floats per point over both planes
__host__ __device__ inline size_t lo_off(int i) { return (size_t)i * 4; }
__host__ __device__ inline size_t hi_off(int n, int i) { return ((size_t)n + (size_t)i) * 4; }
constexpr int MAX_ROWS_PER_WG = 4096;   // fixed-cloud rows one workgroup can own (LDS row-offset table)

struct DevParams {           // cvo.cpp:35-51
    float sigma, sp_thres, c, d, c_ell, c_sigma, min_step, eps, eps_2;
    int max_iter;
    float skin;              // candidate lists are built with radius (1+skin)*r and reused until the cloud has moved skin*r
    float skin_alpha;        // depth-proportional part of the list margin: a moving point may travel skin*r + skin_alpha*|y| (its distance from the camera when the lists were
                             // built) before the lists are stale -- a rotation moves far points most, and far rows have few neighbours (density falls with 1/z^2), so their
                             // wider lists cost little; row i's list radius is (r (1 + skin) + skin_alpha |x_i|) / (1 - skin_alpha).  0 = one margin for all rows
    int fuse_refine;         // the last candidate walk at an ell also makes the next ell's lists (cand_steady<REFINE>) instead of a filter pass of its own at the drop (CVO_HIP_FUSE_REFINE)
    float first_scale;       // the margins of a pair's first lists (built before any twist is known) are this many times skin / skin_alpha
    float alpha_gamma;       // skin_alpha applies at ell = 0.15 and falls with (ell / 0.15)^alpha_gamma (0 = the same at every ell)
    float predict;           // candidate lists are built around positions extrapolated along the previous iteration's twist, this fraction of every point's allowance ahead
                             // (0 = at the current positions); predict_steps caps the extrapolation in units of the last iteration's step
    float predict_steps;
    int overlap_stop_test;   // the next iteration's transform starts before the second stop test of this one (dist_se3) is done (phase_epilogue); 0 = behind it (CVO_HIP_OVERLAP_STOP)
    int nt_min;              // experiment (CVO_HIP_NT_MIN): lists of fewer entries than this are read with plain loads instead of non-temporal ones (0 = always non-temporal)
    int resort;              // rows re-sorted after a list refinement: 0 never, 1 when the cost model says it pays (default), 2 always (tests)
    int colocate;            // the workgroups of a pair on ONE XCD (blocks b, b + 8, ... share one): they read the same moving cloud and swap partial sums through
                             // L2 twice per iteration.  Takes a launch whose pair slots are a multiple of 8; 0 = consecutive blocks (four XCDs for G = 4)
    int adopt_kmax;          // adoption: a finished workgroup only offers its help to pairs with fewer iterations than this behind them (the heavy
                             // early iterations divide well between workgroups; the light late ones are bound by the iteration's fixed latency)
    int adopt_on;            // set per launch by the host: finished workgroups of this launch may help with its pairs that still run (one workgroup and one slot per pair)
    int adopt_inject;        // test knob (CVO_HIP_ADOPT_INJECT): 1 = a helper whose offer has been accepted leaves instead of confirming -- the owner must take the
                             // acceptance back and carry on with the members it has
    int adopt_dwell;         // adoption: a finished workgroup offers its help only when nothing has been queued on the device for this long (ticks of 10 ns; 0 = at once):
                             // a caller that resubmits as launches complete leaves the queue dry for a moment each time, and a helper that joins then holds its CU
                             // for the rest of the pair while the next launch's workgroups wait for one (CVO_HIP_ADOPT_DWELL_US)
};

// per-pair state, read at kernel start and written back at the end (Q1, Q2)
struct PairState {
    float R[9];
    float T[3];
    float ell;
    float transform[12];      // in: cvo::transform before the call; out: final [R^T | -R^T T] (cvo.cpp:817)
    float prev_transform[12]; // out: transform of the last executed iteration (cvo.cpp:815)
    int iter;                 // value of k at the break (stale if max_iter is hit, Q4)
    int A_nonzero;            // nnz of the last iteration (Q5)
    int iterations_run;
    int status;
    int joined_at;            // iteration at which a finished workgroup of the launch joined this pair (adoption), 0 = none
    int adopt_retracted;      // acceptances the owner took back because the helper did not confirm in time (adoption)
    int rebuilds;             // dense culls executed
    int dense_fallbacks;      // rebuilds whose candidates did not fit the lists (dense per-row path taken)
    long long candidates_total;
    long long nonzeros_total; // nonzeros of A summed over the executed iterations (the reference's work: cvo.cpp:166-175 members, :282-306 terms)
    // wall time (100 MHz ticks) workgroup 0 of the pair spent per phase (slots as cvo_batch_last_phase_seconds documents them)
    unsigned long long clk_cycles, clk_ticks;   // shader-clock cycles and 100 MHz ticks workgroup 0 spent on the pair: cycles/ticks*100 MHz = clock
    unsigned long long tail_ticks[4];           // the score block in the kernel's tail (phase_tail_scores), 100 MHz ticks of workgroup 0: transform + list walk (inn_post, Hessian), the cull for inn_pre, its walk, all of it
    unsigned long long predict_mask;            // ... and the cull built its lists around extrapolated positions (DevParams::predict)
    unsigned long long cull_mask;               // bit min(k, 63) set: iteration k began with a dense cull (diagnostics: when do the lists go stale)
    unsigned long long clk_t0;                  // the device's 100 MHz counter when workgroup 0 took the pair up (one counter for the whole device: launches can be laid on one time axis)
    unsigned long long phase_ticks[10];
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

constexpr int TAIL_PRE = 1, TAIL_POST = 2, TAIL_FIXED = 4, TAIL_MOVING = 8, TAIL_HESSIAN = 16;   // PairDesc::score_out[23]

// exchange area for the G workgroups that cooperate on one pair: two buffers
// (alternating by phase), G slots of XCH_WORDS 8-byte {tag, payload} granules.
constexpr int XCH_WORDS = 16;

struct PairDesc {
    const float* fixed;      // two planes of nf float4 (lo_off / hi_off)
    const float* moving;     // two planes of nm float4
    int nf, nm;
    int nf_pad, nm_pad;      // multiples of 64
    int rows_pad;            // rows_per rounded up to 128: row stride of the transposed lists
    int capf;                // flat capacity per row, on average (a workgroup owning r rows may hold r*capf candidates)
    float4* ybuf;            // [G][nm_pad]     transformed moving points {y0,y1,y2,g0} (used when the cloud does not fit in LDS)
    // survivor planes: (nf_pad + G) * capf entries; workgroup g owns [g*rows_per*capf, (g+1)*rows_per*capf)
    // The work buffers below belong to the launch's pair SLOTS, not to the pairs: the pointers are slot 0's, slot s (= blockIdx.x / G)
    // adds s * the strides.  A slot aligns one pair after the other (in-kernel pair queue), so a launch with fewer slots than pairs
    // needs slots x, not pairs x, the memory.
    long long ws_y_stride, ws_list_stride, ws_surv_stride;   // elements per slot of ybuf, of jT (in 16-bit columns) / ent, of surv
    int capn;                // longest row the transposed lists hold (longer => dense fallback)
    uint16_t* jT;            // [G][capn][rows_pad]  the cull's transposed lists: column of entry n of local row li, columns ascending
    uint2* ent;              // same shape, by slot (rows sorted by list length): {colour factor ck as bits (NaN = failed the gate), column}
    uint2* surv;             // nonzeros of A compacted per wave, in the order they were found: {a as bits, slot << 16 | column}
    unsigned long long* xch; // [2][G][XCH_WORDS]
    const PairState* state_in;   // start state: the device copy (carried from the last launch) or a pinned host buffer the caller filled
    PairState* state;            // device copy of the final state
    PairState* state_host;       // pinned host mirror the kernel writes the final state to (no copy engine between launches)
    TraceRow* trace;         // optional
    int trace_cap;
    int* trace_len;
    // Tracker score block in the kernel's tail (cvo::compute_innerproduct, cvo.cpp:475-503, with tran = this alignment's own result and the
    // ell it left behind): non-null = 5 x 24 doubles in pinned host memory, request r at [24 r]: {sum_A, count, 21 Hessian terms}
    // for r = 0 fip(moving, fixed), 1 fip(T moving, fixed), 2 fip(fixed, fixed), 3 fip(moving, moving), 4 se3_Hessian(T moving, fixed);
    // [23] of request 0 = bit mask of the requests the kernel has answered (TAIL_*), the host's score kernel answers the others
    double* score_out;
    const struct SelfCacheEntry* self_fixed;    // the clouds' tables of cached self inner products (ScoreDesc::self_cache)
    const struct SelfCacheEntry* self_moving;
    int run_pair;            // NOT a property of this pair: the pair the launch's position `index of this descriptor` works on (slot s of a launch with a slot per
                             // pair, the s-th pull from the pair queue otherwise).  The host ranks the pairs by the density of their clouds and deals them so
                             // that workgroups sharing an XCD's L2 run pairs of similar density (Engine::launch_impl)
    float* record;           // this pair's 64-byte result record {transform[12], iter, A_nonzero, iterations_run, status as floats}: written by the kernel's
                             // final block, so the cross-GPU gather (or a caller that wants the records on the device) needs no pack kernel behind the launch
    int member_regions;      // adoption launches: member g of a pair keeps its lists and records in a region of its own (sized for the rows it owns
                             // when it joins, at g + 1 members) instead of sharing one region cut into G parts -- see make_ctx
};

// adaptive-ell variant (acvo::align, thirdparty/cvo/src/adaptive_cvo.cpp:490-555)
struct AdaptiveRow { float omega[3], v[3], dl, ell, step; int nnz_xy, nnz_xx, nnz_yy; };   // == cvo_adaptive_row
struct AdaptiveState {
    float R[9], T[3], ell, ell_max, transform[12];
    int iter, iterations_run, status;
    // between the kernels of one iteration
    int stop;                // a stop test fired (or max_iter reached): the kernels still queued return at once
    float M[12];             // this iteration's transform (update_tf)
    float omega[3], v[3], step;
    double dl;
    int nnz_xy, nnz_xx, nnz_yy;
};
struct AdaptiveArgs {
    const float* fixed; const float* moving;   // two planes of float4 each (lo_off / hi_off)
    int nf, nm;
    float4* ybuf;            // nm transformed moving points
    AdaptiveState* state;    // in/out
    double* partials;        // [row blocks][16]: per-workgroup partial sums of a sweep
    AdaptiveRow* trace; int trace_cap; int* trace_len;
    float ell_min, dl_step;
    DevParams P;
};

// score kernels (function_inner_product / se3_Hessian)
struct SelfCacheEntry { float ell; int valid; double sum, count; };
constexpr int SELF_CACHE_N = 4;
struct ScoreDesc {
    const float* a;          // two planes of na float4: queried cloud (positions optionally transformed by tran)
    const float* b;          // two planes of nb float4: searched cloud
    const float* bbox;       // its 32-point group boxes: planes lo x, y, z, y/z then hi x, y, z, y/z of nbox floats each
    int nbox;
    int na, nb;
    float tran[12];
    int use_tran;            // 0: a as stored, 1: a transformed by tran, 2: a transformed by from->transform (the pair's own align() result)
    float ell;
    const PairState* from;   // non-null: ell (and the transform when use_tran == 2) come from this device-resident state, so a score block can be
                             // queued behind the align launch that produces it without a host round trip
    int want_hessian;        // 0: inner product only, 1: Hessian terms too
    double* out;
    // fip(cloud, cloud) of an untransformed cloud against itself depends on the cloud and ell only (cvo.cpp:496-497), and the moving cloud
    // of one frame is the fixed cloud of the next (update_fixed_pcd, cvo.cpp:578-582): non-null = the cloud's table of SELF_CACHE_N
    // entries {ell, valid, sum, count} in HBM (cleared when the cloud's points are written); a hit skips the sweep, a miss fills an entry
    SelfCacheEntry* self_cache;
};

// host clouds in the reference layout -> the two float4 planes (cvo_pack_clouds_kernel): one descriptor per cloud
struct PackDesc {
    unsigned long long raw_off;   // floats from the start of the raw staging buffer: n x 3 positions (AoS, data_type.h:30), then 5 channel-major arrays of n (data_type.h:75)
    float* dst;                   // two planes of n float4
    int n, pad_;
    unsigned long long feat_off;  // floats from the same start to the 5 channel-major feature arrays; 0 = right behind the positions (raw_off + 3 n).  Not 0 for clouds
                                  // handed over in caller-registered memory (cvo_host_register), whose two arrays lie where the caller has them
};

// a score block: up to 8 inner-product / Hessian requests evaluated by one launch
constexpr int SCORE_MAXREQ = 8;
struct ScoreBatch {
    ScoreDesc d[SCORE_MAXREQ];
    int n;
};

}  // namespace cvohip
