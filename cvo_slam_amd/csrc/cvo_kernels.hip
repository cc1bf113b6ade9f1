// cvo_kernels.hip -- gfx950 kernels of the CVO alignment hot path.
//
// cvo_align_kernel: the whole of cvo::align() (thirdparty/cvo/src/cvo.cpp:763-821)
// for a batch of independent frame pairs in ONE persistent launch.  G workgroups
// cooperate on a pair (rows of the fixed cloud are dealt round-robin to them) and stay
// resident for all of its iterations; R, T, ell never leave the device.
//
// One iteration (cvo.cpp:768-813), each phase a non-inlined device function:
//   T   transform_pcd (cvo.cpp:336-341): y_j = M p_j into LDS (HBM when the cloud is too large);
//       exact displacement of every point since the candidate lists were built (the build
//       transform is kept, not a snapshot); decision: keep the lists / cull / refine
//   S   [lists stale by motion]  dense cull straight into per-row candidate lists: the moving
//       cloud sits in an LDS tile (SoA, ds_read_b128 broadcasts), a lane carries two rows,
//       3 sub + 3 fma + 1 v_alignbit per pair test; 32-column groups whose box (x, y, z and the
//       ray slope y/z) is out of reach of the wave's rows are skipped; hits are appended in
//       ascending column order (= the CSR order of Eigen::setFromTriplets, cvo.cpp:182) to
//       jT[n][row].  Lists are built with radius (1+skin)*r and stay valid until some point has
//       moved skin*r
//   X   [after S]  rows sorted by list length (deterministic counting sort) into slots; 64
//       slots = one block of near-equal lists, blocks dealt to the waves in serpentine order
//   R   [ell dropped]  the old lists are a superset of the new radius: filtered in place
//   C   one lane per slot walks its list, 4 entries per step: the reference's own pair
//       arithmetic (cvo.cpp:166-175: un-fused f32 d2, colour gate, double exp, a > sp_thres),
//       f32 row sums in column order (cvo.cpp:213-223), f64 across rows (cvo.cpp:226-230) by
//       wave shuffles + LDS, exchanged between the pair's workgroups as tagged 8-byte granules;
//       nonzeros {a, slot, column} are compacted per wave.  The first pass over new lists also
//       evaluates the colour gate / factor once per entry
//   L   one lane per nonzero: beta..epsil and the B..E terms (cvo.cpp:282-306), f64
//   E   one lane: cubic, stop tests, Exp_SEK3, pose update, ell schedule
//       (cvo.cpp:317-333, 782-812)
// The cull uses fused arithmetic and a widened radius (a superset of the reference's
// neighbourhood); membership in A is decided in C by the reference's own expression, so the
// sparse set, every kernel value and every per-row sum follow the oracle's float sequence
// regardless of when the lists were built.
//
// MFMA is deliberately not used: S is a distance test + compare, C/L are exp-heavy
// per-nonzero work; neither is a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvo_device.h"
#include "cvo_math.hpp"
#include <algorithm>
#include <type_traits>

// The file is compiled twice into the library: as namespace cvohip with workgroups of up to 512 threads (two waves per SIMD, 256 VGPRs: the 3 k-point
// shape, bound by issue) and, with -DCVO_KNS=cvohip_w3 -DCVO_BLOCK_MAX=768, as namespace cvohip_w3 with up to 768 threads (three waves per SIMD, 168
// VGPRs: clouds in the plane layout, whose walks wait for memory more than they issue -- +5.8 % at 9 k points, -3 % at 3 k; profiles/r03_block_size_ab.txt).
#ifndef CVO_KNS
#define CVO_KNS cvohip
#else
namespace CVO_KNS { using namespace cvohip; }
#endif
namespace CVO_KNS {

// phase timers (PairState::phase_ticks): a read of the 100 MHz clock is a scalar memory instruction every wave waits for
#ifdef CVO_NO_TIMERS
#define CVO_NOW() 0ull
#else
#define CVO_NOW() __builtin_amdgcn_s_memrealtime()
#endif

#ifndef CVO_WAVES_PER_SIMD
#define CVO_WAVES_PER_SIMD 2      // 2: 256 VGPRs, one 512-thread workgroup per CU (measured faster); 4: 128 VGPRs, two per CU
#endif
#ifndef CVO_BLOCK_MAX
#define CVO_BLOCK_MAX 512         // threads of the largest workgroup: 512 = 2 waves per SIMD with 256 VGPRs each.  (1024 = 4 waves per SIMD with 128 each
#endif                            //  is an experiment knob: measured, no phase gets faster -- DESIGN.md "Measured in round 2")
constexpr int BLOCK_MAX = CVO_BLOCK_MAX;
// Phases are functions of their own (own register allocation).  CVO_INLINE_PHASES: a bit mask of phases compiled INTO run_pair instead (experiment builds):
// 1 line search, 2 epilogue, 4 transform (all three measured: run_pair then spills around its remaining calls), 8 below -- a call costs the callee-saved
// saves and, at its return, a wait for their reloads from scratch
#ifndef CVO_INLINE_PHASES
#if CVO_BLOCK_MAX > 512
#define CVO_INLINE_PHASES 0       // the three-wave build (168 registers): the merged function spills inside its loops
#else
#define CVO_INLINE_PHASES 8       // measured: +1.2 % in the long run, -2.2 % on one pair's latency (profiles/r05_one_call_per_iteration_ab.txt)
#endif
#endif
#define CVO_PHASE_FN(bit) static __device__ __attribute__((CVO_PHASE_ATTR_##bit))
#if CVO_INLINE_PHASES & 8     // 8: candidate walk + line search + epilogue compiled into ONE function, phase_iteration (one call per iteration instead of three)
#define CVO_PHASE_ATTR_8 always_inline
#else
#define CVO_PHASE_ATTR_8 noinline
#endif
#if CVO_INLINE_PHASES & (1 | 8)
#define CVO_PHASE_ATTR_1 always_inline
#else
#define CVO_PHASE_ATTR_1 noinline
#endif
#if CVO_INLINE_PHASES & (2 | 8)
#define CVO_PHASE_ATTR_2 always_inline
#else
#define CVO_PHASE_ATTR_2 noinline
#endif
#if CVO_INLINE_PHASES & 4
#define CVO_PHASE_ATTR_4 always_inline
#else
#define CVO_PHASE_ATTR_4 noinline
#endif
constexpr int MAX_WAVES = BLOCK_MAX / 64;
constexpr unsigned ADOPT_FREE = 0u, ADOPT_REQUEST = 1u, ADOPT_ACCEPT = 2u, ADOPT_CLOSED = 3u, ADOPT_CONFIRMED = 4u;   // states of a pair's adoption word (cvo_align_kernel)
constexpr unsigned long long ADOPT_CONFIRM_TICKS = 5000ull;   // 50 us at 100 MHz: how long an owner waits for an accepted helper to confirm before it takes the acceptance back
#ifndef CVO_ADOPT_GMAX
#define CVO_ADOPT_GMAX 4
#endif
constexpr int ADOPT_GMAX = CVO_ADOPT_GMAX;                                    // workgroups a pair can grow to by adoption (the host sizes the exchange area and the buffers' slack for it)
constexpr float SKIN_DENSE_SCENE = 0.25f;   // list radius margin a pair falls back to when the lists of the launch's margin overflow (round 2's value)
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
    float Mn[12];          // the next iteration's transform while lane 0's second stop test may still fire (transform_body_t, the epilogue's call)
    float Mb[12];          // the transform the candidate lists were built (or last filtered) under
    float Rb;              // radius they were built with: its part that is the same for every row, r (1 + skin)
    float alpha_build;     // ... and the depth-proportional part of the margin they were built (or last filtered) with: row i holds every column within
                           // (Rb + alpha_build |x_i|) / (1 - alpha_build) of it (DevParams::skin_alpha)
    float reach;           // phase_transform -> phase_refine: how far the points are from where they were listed, beyond what the lists allow for by themselves
    float xmax;            // largest |x_i| of the pair's rows (phase_cull; of this workgroup's rows when it is the pair's only one)
    int predicted;         // culls of this pair whose lists were built around extrapolated positions (diagnostics)
    float skin0, alpha0;   // the launch's list margins (a pair that falls back to the dense-scene margin changes P.skin / P.skin_alpha for itself)
    int twist_ok;          // omega, v, step below are those of the pair's previous iteration (set by this workgroup's own candidate phase and epilogue)
    float ell_build;
    float fred[MAX_WAVES];
    int wsum[MAX_WAVES];
    int wcnt[MAX_WAVES];   // survivors each wave compacted in the candidate phase
    int wtot[MAX_WAVES];   // candidates each wave owns (sum of its rows' list lengths)
    int wbase[MAX_WAVES];  // start of the wave's survivor segment (exclusive prefix of wtot)
    int wnb[MAX_WAVES];    // 64-slot blocks each wave walks in the candidate phase (serpentine deal, see phase_sort)
    unsigned short blk_lmax[MAX_ROWS_PER_WG / 64 + 2];   // longest list of each block (kept by phase_sort / refine_lists: the walks need it before their first step)
    int lmax;              // longest candidate list of this workgroup's rows
    int list_valid;
    int dense_mode;        // candidates did not fit the lists: per-row dense fallback until the next rebuild
    int total;             // candidates in this workgroup's flat list
    int stop;
    int status;
    int iter_at_break;
    int nnz;
    int cand;
    int rebuilds;
    int refines;           // list rebuilds done by filtering the old lists (ell drops)
    int resort;            // this refinement re-sorts the rows by their new list lengths (phase_refine)
    int resort_pending;    // ... asked for by a candidate walk that filtered the lists on its way; done before the next iteration's walk (phase_resort)
    int fused;             // list refinements made by a candidate walk on its way (diagnostics)
    float rc, rc_ell;      // gate radius sqrt(gate_d2_align(ell)) and the ell it was worked out for (transform_body_t; rc_ell < 0: none yet)
    float inv_c, inv_d;    // 1 / P.c, 1 / P.d and ...
    unsigned long long gates_store[10];   // ... struct Gates of rc_ell: the candidate phase's constants (two double-precision logs, a float log, four divisions) once per ell
    float reach_now;       // this iteration's reach (phase_transform), for the candidate walk that makes the next ell's lists
    int cull_next;         // next block pair of the cull to hand out
    int ws_slot;           // pair slot of the launch whose work buffers this workgroup uses (its own, or the one of the pair it helps with)
    int adopt_req;         // a finished workgroup of the launch has asked to help with this pair (1 + its block index), seen by the epilogue
    unsigned adopt_k;      // iteration at which a helper joins the pair it adopted
    int joined_at;         // owner: iteration at which a helper joined this pair (0 = none)
    int retracted;         // owner: acceptances taken back (the helper did not confirm in time)
    unsigned long long* adopt_word;   // this pair's adoption word in the launch's queue area, or null (no adoption for this pair / any more)
    int dense_fallbacks;
    int rebuild;           // this iteration rebuilds the candidate lists
    int rows_cap;          // entries of the three row/slot tables in LDS (the workgroup's rows, padded)
    int y_cap;             // points the LDS-resident moving cloud has room for (stride of the SoA planes)
    int x_lds;             // the fixed points, by slot, sit in the (otherwise idle) cull tile: lx/ly/lz[slot]
    int ctx_rows_per, ctx_nrows;   // this workgroup's share of the current pair's rows (pair_rows)
    int tab_cols;          // columns the line-search table has room for (0 = no table)
    unsigned launch_tag;   // high 16 bits of every exchange tag: this launch's sequence number (granules of earlier launches never match)
    unsigned long long sub[4];   // thread 0's time inside the candidate phase: prologue, row loop, workgroup reduction, exchange
    unsigned long long ticks[10];   // thread 0's time per phase (PairState::phase_ticks), summed over the pair's iterations
    unsigned long long cull_mask;   // PairState::cull_mask
    unsigned long long tail_ticks[4];
    unsigned long long predict_mask;
    unsigned long long cand_total;  // list candidates evaluated so far (PairState::candidates_total)
    unsigned long long nnz_total;   // nonzeros of A so far (PairState::nonzeros_total)
#ifdef CVO_KTRACE
    unsigned long long ksub[4];  // experiment builds: line-search walk, line-search reduction, epilogue scalar part, epilogue transform (ticks, this iteration)
    unsigned long long kabs[16];  // CVO_KTRACE_EPI == 2: absolute times inside the epilogue (lane 0: part A done, part B done, staleness maximum there, decision made; thread 64: past the first barrier, its points done)
#endif
    float omega[3];        // this iteration's twist (f32, cvo.cpp:234-235)
    float v[3];
    float dist;
    DevParams P;
    // this workgroup's view of the current pair (struct Ctx), worked out once per pair (and again when the pair gains a member): every phase is a
    // function of its own and would otherwise fetch the descriptor's fields from global memory first thing, a round trip per phase call
    unsigned long long ctx_store[16];
};

// Pointers read out of a PairDesc are generic to the compiler, which then emits FLAT loads/stores
// (no counted waits, everything drains at each use).  They all point into hipMalloc'ed memory,
// so the kernel re-types them as global (address space 1) once per pair.
#define CVO_GLOBAL __attribute__((address_space(1)))
typedef CVO_GLOBAL float gfloat;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef CVO_GLOBAL v4f gv4f;
// float4 array in global memory (HIP's float4 is a class whose copy operations only take generic pointers)
struct GF4 {
    gv4f* p;
    __device__ __forceinline__ float4 operator[](size_t i) const { const v4f t = p[i]; return make_float4(t.x, t.y, t.z, t.w); }
    __device__ __forceinline__ void set(size_t i, const float4 v) const { v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; p[i] = t; }
};
typedef CVO_GLOBAL uint16_t gu16;
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef CVO_GLOBAL v2u gv2u;
typedef CVO_GLOBAL uint32_t gu32;
typedef CVO_GLOBAL int gint;
typedef CVO_GLOBAL unsigned long long gu64;
__device__ __forceinline__ float4 ld4(const gfloat* p) { const v4f t = *reinterpret_cast<const gv4f*>(p); return make_float4(t.x, t.y, t.z, t.w); }
// Cache policy of the per-iteration streams.  With every CU at work the walks wait for memory a quarter longer than one launch alone does (a bandwidth hog
// beside one launch slows them by 60-75 %, profiles/r03_contention_probe.txt), so what the L2 keeps matters:
//  * list entries read by the steady candidate walk: non-temporal loads -- a dense pair's list is 2 MB per iteration, it never survives in the XCD's 4 MiB L2
//    until the next iteration, and marked as a stream it no longer evicts what the XCD's other workgroups re-read (+2 % at 3 k points, +3.5 % at 9 k);
//  * list entries written by the first pass after a cull: non-temporal 16-byte stores (+1 % for the width, +1 ... 1.7 % for the policy);
//  * nonzero records (written by the candidate walk, read by the line search tens of microseconds later), the cull's raw lists (written, then read once by
//    the first pass) and the in-place refinement: plain -- streaming them loses 2 ... 8 % (they are re-read soon enough to hit).
// profiles/r03_cache_policy_ab.txt, r03_entry_stores_ab.txt, r03_raw_list_and_refine_policy_ab.txt, r03_eth3d_entries_nt_ab.txt, r03_masked_entry_loads_ab.txt.
// -DCVO_NT_REC / _LD / _ST, -DCVO_NT_JT, -DCVO_NT_REFINE, -DCVO_PLAIN_ENT_LD, -DCVO_PLAIN_ENT_ST build the alternatives.
// Entries of a list in slot order: entry n of slot s is ent[2 * ((n >> 1) * rows_pad + s) + (n & 1)] -- two consecutive entries of a row are neighbours, so the
// steady walk fetches its four entries per step with two 16-byte loads per lane (the vector memory pipe is as busy as the VALU in that walk: four loads and
// up to four record stores per step; see DESIGN.md, "Measured in round 3").  ent_ix(n, rows_pad) is the offset from the slot's base ent + 2 * s.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef CVO_GLOBAL v4u gv4u;
__device__ __forceinline__ size_t ent_ix(int n, size_t rows_pad) { return (size_t)(n >> 1) * rows_pad * 2 + (size_t)(n & 1); }
// the in-place filter of the lists after an ell drop (refine_lists): -DCVO_NT_REFINE streams its reads and writes
__device__ __forceinline__ v2u ld_rf(const gv2u* p) {
#ifdef CVO_NT_REFINE
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_rf(gv2u* p, const v2u v) {
#ifdef CVO_NT_REFINE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// the cull's raw lists are written once and read once (by the first candidate pass after the cull): -DCVO_NT_JT marks both as streams
__device__ __forceinline__ void st_jt(gv2u* p, const v2u v) {
#ifdef CVO_NT_JT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ v2u ld_jt(const gv2u* p) {
#ifdef CVO_NT_JT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
#ifdef CVO_PLAIN_ENT_LD
#define ENT_NT(YM) false
#else
#define ENT_NT(YM) true      // every layout: +2 % at 3 k points, +3.5 % at 9 k (profiles/r03_cache_policy_ab.txt, r03_eth3d_entries_nt_ab.txt)
#endif
template <bool NT>
__device__ __forceinline__ v4u ld_ent2(const gv4u* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
// element i of a wave-uniform array, addressed as uniform base + 32-bit BYTE offset: the form the hardware takes as SGPR base + VGPR offset -- no 64-bit address
// arithmetic per lane (the arrays indexed this way are far below 4 GB)
typedef CVO_GLOBAL char gchar;
template <typename T>
__device__ __forceinline__ T* at_off(T* base, unsigned index) {
#ifdef CVO_ADDR64
    return base + index;
#else
    return reinterpret_cast<T*>(reinterpret_cast<gchar*>(const_cast<typename std::remove_const<T>::type*>(base)) + index * (unsigned)sizeof(T));
#endif
}
template <bool NT>
__device__ __forceinline__ v2u ld_ent(const gv2u* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
__device__ __forceinline__ void st_rec(gv2u* p, const v2u v) {
#if defined(CVO_NT_REC) || defined(CVO_NT_REC_ST)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ v2u ld_rec(const gv2u* p) {
#if defined(CVO_NT_REC) || defined(CVO_NT_REC_LD)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// ---------------------------------------------------------------- reductions
// butterfly inside the wave (every lane ends with the wave total), one LDS slot
// per wave, then lanes 0..K-1 of wave 0 add the waves in order: deterministic.
// v of lane (i ^ X) for X = 8, 4, 2, 1 by DPP moves inside the row of 16 lanes (vector pipe, no LDS round trip):
// 8 = row_ror:8, 2 = quad_perm [2,3,0,1], 1 = quad_perm [1,0,3,2]; 4 = row_shl:4 into the lanes whose bit 2 is clear (banks 0, 2)
// and row_shr:4 into the others (banks 1, 3).
template <int X>
__device__ __forceinline__ int dpp_xor_i32(int x) {
    if (X == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);
    if (X == 4) { const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0x5, false); return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false); }
    if (X == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
    return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);
}
template <int X>
__device__ __forceinline__ double dpp_xor(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)dpp_xor_i32<X>((int)(unsigned)b), hi = (unsigned)dpp_xor_i32<X>((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int X>
__device__ __forceinline__ float dpp_xor(float v) { return __uint_as_float((unsigned)dpp_xor_i32<X>((int)__float_as_uint(v))); }
// all-lanes sum of a wave: the xor-butterfly over 32, 16, 8, 4, 2, 1, the last four exchanges without the LDS (same partners, same bits)
__device__ __forceinline__ double wave_sum_all(double v) {
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    v += dpp_xor<8>(v);
    v += dpp_xor<4>(v);
    v += dpp_xor<2>(v);
    v += dpp_xor<1>(v);
    return v;
}
// KB < K: values KB..K-1 are per-wave numbers carried by lane 0 alone (counts): no butterfly for them.
// The totals end up in sh->vals[0..K) AND in the return value of lanes 0..K-1 of wave 0.  SYNC_AFTER = false leaves out the closing
// barrier: only wave 0 may then read sh->vals (its own lanes wrote them; LDS operations of one wave complete in order) until the
// caller's next barrier.
template <int K, int KB = K, bool SYNC_AFTER = true>
__device__ __forceinline__ double block_reduce(double (&v)[K], Shared* sh, int tid, int nwaves) {
#pragma unroll
    for (int k = 0; k < KB; ++k) v[k] = wave_sum_all(v[k]);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh->red[wave * 8 + k] = v[k];
    }
    __syncthreads();
    double s = 0;
    if (tid < K) {
        for (int w = 0; w < nwaves; ++w) s += sh->red[w * 8 + tid];
        sh->vals[tid] = s;
    }
    if (SYNC_AFTER) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    return s;
}

__device__ __forceinline__ float block_max(float v, Shared* sh, int tid, int nwaves) {
    v = fmaxf(v, __shfl_xor(v, 32, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, dpp_xor<8>(v)); v = fmaxf(v, dpp_xor<4>(v)); v = fmaxf(v, dpp_xor<2>(v)); v = fmaxf(v, dpp_xor<1>(v));
    if ((tid & 63) == 0) sh->fred[tid >> 6] = v;
    __syncthreads();
    float m = sh->fred[0];
    for (int w = 1; w < nwaves; ++w) m = fmaxf(m, sh->fred[w]);
    __syncthreads();
    return m;
}

// maximum over the wave of a value that is >= 0 in every lane (its bits then order like the numbers), wave-uniform: four DPP exchanges inside the rows
// of 16 lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: a maximum does not care who its partners are), then the four rows' values through
// scalar registers -- no LDS round trip
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = (int)__float_as_uint(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16), c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return __uint_as_float((unsigned)max(max(a, b), max(c, d)));
}

// exclusive prefix sum of one int per thread over the workgroup; total returned to every thread
__device__ __forceinline__ int block_exclusive_scan(int v, Shared* sh, int tid, int nwaves, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    if (lane == 63) sh->wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) { const int s = sh->wsum[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    total = tot;
    return base + inc - v;
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
__device__ __forceinline__ bool group_exchange(Shared* sh, gu64* xch, int G, int g, unsigned epoch, int lane) {
    gu64* buf = xch + (size_t)(epoch & 1u) * G * XCH_WORDS;
    if (lane < 2 * K) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sh->vals[lane >> 1]);
        const unsigned pay = (lane & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(&buf[g * XCH_WORDS + lane], ((unsigned long long)epoch << 32) | pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double tot = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    // four workgroups' granules per load (a wave's 64 lanes = 4 x XCH_WORDS), and the loads of up to sixteen members in flight together: ONE L2 round trip per
    // poll for the whole group, not one per four members (round 5: the exchange of eight members took two round trips at least, twice per iteration)
    static_assert(2 * K <= XCH_WORDS && 4 * XCH_WORDS == 64, "granule layout");
    const int sub = lane >> 4, wrd = lane & 15;
    for (int gq = 0; gq < G; gq += 16) {
        unsigned long long x[4] = {0, 0, 0, 0};
        bool mine[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = (gq + 4 * r + sub < G) && (wrd < 2 * K);
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mine[r] && (unsigned)(x[r] >> 32) != epoch) x[r] = __hip_atomic_load(&buf[(gq + 4 * r + sub) * XCH_WORDS + wrd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < 4; ++r) ok = ok && (!mine[r] || (unsigned)(x[r] >> 32) == epoch);
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t_start > 300000000ull) return false;   // 3 s at 100 MHz
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (gq + 4 * r >= G) break;                             // (wave-uniform)
            const unsigned pay = (unsigned)x[r];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                           // members in workgroup order: the sum has one fixed order everywhere
                const unsigned lo = __shfl(pay, (16 * i + 2 * lane) & 63, 64), hi = __shfl(pay, (16 * i + 2 * lane + 1) & 63, 64);
                if (lane < K && gq + 4 * r + i < G) tot += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            }
        }
    }
    if (lane < K) sh->vals[lane] = tot;
    return true;
}

// ---------------------------------------------------------------- exact pair arithmetic
struct Gates {
    float d2_thres, d2c_thres, sp;
    double den_l, den_c;      // 2.0*l*l, 2.0*c_ell*c_ell
    float s2, csig2;
    double inv_den_l, inv_den_c;
    float q_lim, q_il, q_ic;  // conservative f32 pre-test of a > sp before the double exps
    bool poly_ok;             // d2_thres/(2 l^2) <= 0.25: exponents of pairs inside the radius need no range reduction
};
static_assert(sizeof(Gates) <= sizeof(((Shared*)nullptr)->gates_store), "Shared::gates_store holds a Gates");

// exp(x) in double for the only arguments the survivor path produces: the pre-test
// bounds both exponents by q_lim (~0.22 with the reference's constants), so no range
// reduction is needed: degree-13 Taylor/Horner with fma, truncation < 1e-19, rounding
// ~1 ulp like libm's exp; anything outside [-0.25, 0] takes the library routine.  The
// value is rounded to f32 right after (cvo.cpp:172-173), where a 1-ulp double
// difference is invisible except on ~1e-8 of inputs.
// p*x + c as the three-address v_fma_f64 with the coefficient in an SGPR pair.  Left to itself the compiler picks the
// two-address v_fmac_f64 for half of the Horner steps and pays a v_mov_b64 of the coefficient for each (28 moves per 4 entries).
__device__ __forceinline__ double horner_step(double p, double x, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(x), "s"(c));
    return d;
}
// the same polynomial for PFN arguments at once, coefficient by coefficient: the PFN Horner chains are independent, and issued
// side by side no v_fma_f64 waits for the one before it (evaluated one after the other, every step is a dependent
// double-precision op behind a wait state)
template <int PFN, int DEG = 13>
__device__ __forceinline__ void exp_poly13_n(const double (&x)[PFN], double (&p)[PFN]) {
    // 1/13!, 1/12!, ..., 1/3!; a DEG-term chain starts at 1/DEG!
    const double cf[11] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                           1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0};
    static_assert(DEG >= 4 && DEG <= 13, "degree");
#pragma unroll
    for (int u = 0; u < PFN; ++u) p[u] = cf[13 - DEG];
#pragma unroll
    for (int k = 14 - DEG; k < 11; ++k) {
#pragma unroll
        for (int u = 0; u < PFN; ++u) p[u] = horner_step(p[u], x[u], cf[k]);
    }
#pragma unroll
    for (int u = 0; u < PFN; ++u) p[u] = __builtin_fma(p[u], x[u], 0.5);
#pragma unroll
    for (int u = 0; u < PFN; ++u) p[u] = __builtin_fma(p[u], x[u], 1.0);
#pragma unroll
    for (int u = 0; u < PFN; ++u) p[u] = __builtin_fma(p[u], x[u], 1.0);
}
__device__ __forceinline__ double exp_poly13(double x) {
    double p = 1.0 / 6227020800.0;                                  // 1/13!
    p = horner_step(p, x, 1.0 / 479001600.0);
    p = horner_step(p, x, 1.0 / 39916800.0);
    p = horner_step(p, x, 1.0 / 3628800.0);
    p = horner_step(p, x, 1.0 / 362880.0);
    p = horner_step(p, x, 1.0 / 40320.0);
    p = horner_step(p, x, 1.0 / 5040.0);
    p = horner_step(p, x, 1.0 / 720.0);
    p = horner_step(p, x, 1.0 / 120.0);
    p = horner_step(p, x, 1.0 / 24.0);
    p = horner_step(p, x, 1.0 / 6.0);
    p = __builtin_fma(p, x, 0.5);                                   // inline constants: no register either way
    p = __builtin_fma(p, x, 1.0);
    p = __builtin_fma(p, x, 1.0);
    return p;
}
// exp(x) for x <= 0 of any size, branch-free: Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2, the same polynomial
// (truncation 4e-18 at |r| = 0.35), scaled by 2^k with v_ldexp_f64.  ~1 ulp like libm's exp; the value is rounded to f32
// right after (cvo.cpp:172-173).
__device__ __forceinline__ double exp_neg(double x) {
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    return __builtin_ldexp(exp_poly13(r), (int)k);
}
__device__ __forceinline__ double exp_small(double x) {
    if (!(x >= -0.25 && x <= 0.0)) return exp(x);
    return exp_poly13(x);
}

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
    // k = s2*exp(-d2/(2.0*l*l)), ck = c_sigma^2*exp(-d2c/(2.0*c_ell*c_ell)) evaluated in double, stored f32
    // (cvo.cpp:172-173); the division is a multiplication by the double reciprocal (<= 1 ulp of the argument)
    const float k = (float)((double)G.s2 * exp_small((double)(-d2) * G.inv_den_l));
    const float ck = (float)((double)G.csig2 * exp_small((double)(-d2c) * G.inv_den_c));
    const float a = ck * k;
    return a > G.sp ? a : 0.f;
}

// A listed candidate carries its colour factor ck (cvo.cpp:169-173 depend on the two points only; NaN = failed the colour
// gate), so only the geometric half of cvo.cpp:166-175 is left per iteration.
__device__ __forceinline__ float se_kernel_value_ck(const float* xi, const float4 yj, float ck, const Gates& G) {
    const float e0 = xi[0] - yj.x, e1 = xi[1] - yj.y, e2 = xi[2] - yj.z;
    float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;                  // nanoflann.hpp:403-406
    if (!(d2 < G.d2_thres)) return 0.f;
    const float k = (float)((double)G.s2 * exp_small((double)(-d2) * G.inv_den_l));
    const float a = ck * k;
    return a > G.sp ? a : 0.f;                                                 // false for NaN
}
// Branch-free form for the steady-state loop (several independent entries in flight per lane); requires Gates::poly_ok,
// i.e. every exponent of a pair inside the radius lies in [-0.25, 0].  A rejected pair's arithmetic runs on a clamped
// argument and is thrown away.
// se_kernel_value_flat for PFN entries of one row at once
template <int PFN>
__device__ __forceinline__ void se_kernel_values_flat(const float* xi, const float4 (&yj)[PFN], const float (&ck)[PFN], const bool (&active)[PFN], const Gates& G,
                                                      float (&a_out)[PFN], float (&e_out)[PFN][3], float* d2_out = nullptr) {
    double x[PFN], p[PFN]; bool pass[PFN];
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
        const float e0 = xi[0] - yj[u].x, e1 = xi[1] - yj[u].y, e2 = xi[2] - yj[u].z;
        float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;              // nanoflann.hpp:403-406
        if (d2_out) d2_out[u] = d2;
        pass[u] = active[u] & (d2 < G.d2_thres);
        // inside the radius the exponent is in [-0.25, 0] (Gates::poly_ok): twelve terms leave 2.4e-18 of exp there.  Outside it
        // the polynomial returns some finite or infinite number that `pass` throws away; no clamp needed
        x[u] = (double)(-d2) * G.inv_den_l;
        e_out[u][0] = e0; e_out[u][1] = e1; e_out[u][2] = e2;
    }
    exp_poly13_n<PFN, 12>(x, p);
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
        const float k = (float)((double)G.s2 * p[u]);
        const float a = ck[u] * k;
        a_out[u] = (pass[u] & (a > G.sp)) ? a : 0.f;
    }
}
// The steady walk's k = (float)(s2 * exp(x)), x in [-0.25, 0] (Gates::poly_ok), from a SHORTER polynomial -- with a guard that makes the f32 result the one the
// long polynomial gives.  The double is rounded to f32 right away (cvo.cpp:172), so all the f32 result needs of it is which side of a rounding boundary it
// lies on: a degree-7 interpolant of exp on [-1/4, 0] (Chebyshev nodes, scripts/derive/exp7_coefficients.py: 1.3e-14 relative, measured on 2e6 points) times s2
// (folded into the coefficients once per phase: 1.1e-16 each) decides that for every value farther than its own error from a boundary.  The 29 mantissa bits
// the conversion drops say how far: within 256 double ulps (>= 2.8e-14 relative, twice the error budget) of the half-way pattern the entry is re-evaluated with
// the 12-term chain -- 512 of 2^29 patterns, one entry in a million.  Seven half-rate Horner steps instead of twelve and no product with s2: 6 of an entry's 16
// double-precision instructions, for 3 integer ones.
struct Exp7 { double c[7]; double c7; double inv_den; };   // c[0..6], inv_den wave-uniform (scalar operands of the Horner steps); c7 in a vector register pair (the chain's start)
// a wave-uniform double in a scalar register pair.  (Not __builtin_amdgcn_readfirstlane: the compiler folds that away for a value it can prove uniform and
// then hands the "s" operands of the Horner steps a VECTOR register pair -- three 64-bit vector operands per v_fma_f64, measurably slower than two.)
__device__ __forceinline__ double uni_d(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    unsigned lo, hi;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(lo) : "v"((unsigned)b));
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(hi) : "v"((unsigned)(b >> 32)));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ Exp7 make_exp7(const Gates& G) {
    const double c[8] = {0x1.fffffffffffa3p-1, 0x1.fffffffff45b5p-1, 0x1.fffffff859bf4p-2, 0x1.5555536ab4a73p-3, 0x1.5554dc6d808f1p-5, 0x1.1100d93fb8b12p-7,
                         0x1.69aae1f1adc28p-10, 0x1.6f5e2611a1689p-13};
    Exp7 E;
#pragma unroll
    for (int k = 0; k < 7; ++k) E.c[k] = uni_d((double)G.s2 * c[k]);    // wave-uniform: the Horner steps take them as scalar operands
    E.c7 = (double)G.s2 * c[7];
    E.inv_den = uni_d(G.inv_den_l);
    return E;
}
template <int PFN>
__device__ __forceinline__ void se_kernel_values_flat7(const float* xi, const float4 (&yj)[PFN], const float (&ck)[PFN], const bool (&active)[PFN], const Gates& G, const Exp7& E7,
                                                       float (&a_out)[PFN], float (&e_out)[PFN][3], float* d2_out = nullptr) {
    double x[PFN], p[PFN]; bool pass[PFN]; float kk[PFN], d2s[PFN];
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
        const float e0 = xi[0] - yj[u].x, e1 = xi[1] - yj[u].y, e2 = xi[2] - yj[u].z;
        float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;              // nanoflann.hpp:403-406
        if (d2_out) d2_out[u] = d2;
        d2s[u] = d2;
        pass[u] = active[u] & (d2 < G.d2_thres);
        asm("v_mul_f64 %0, %1, %2" : "=v"(x[u]) : "v"((double)(-d2)), "s"(E7.inv_den));
        e_out[u][0] = e0; e_out[u][1] = e1; e_out[u][2] = e2;
    }
#pragma unroll
    for (int u = 0; u < PFN; ++u) p[u] = horner_step(E7.c7, x[u], E7.c[6]);
#pragma unroll
    for (int k = 5; k >= 0; --k) {
#pragma unroll
        for (int u = 0; u < PFN; ++u) p[u] = horner_step(p[u], x[u], E7.c[k]);
    }
    bool amb = false, ambu[PFN];
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
        const unsigned lo = (unsigned)(unsigned long long)__double_as_longlong(p[u]);
        ambu[u] = pass[u] & (((lo + (256u - (1u << 28))) & 0x1FFFFFFFu) < 512u);
        amb |= ambu[u];
        kk[u] = (float)p[u];
    }
    if (__builtin_expect(__ballot(amb) != 0ull, 0)) {                           // one entry in a million: the long chain, as se_kernel_values_flat evaluates it
#pragma unroll
        for (int u = 0; u < PFN; ++u) {
            if (ambu[u]) {
                const double xe[1] = {(double)(-d2s[u]) * G.inv_den_l};
                double pe[1];
                exp_poly13_n<1, 12>(xe, pe);
                kk[u] = (float)((double)G.s2 * pe[0]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
        const float a = ck[u] * kk[u];
        a_out[u] = (pass[u] & (a > G.sp)) ? a : 0.f;
    }
}
__device__ __forceinline__ float se_kernel_value_flat(const float* xi, const float4 yj, float ck, bool active, const Gates& G) {
    const float e0 = xi[0] - yj.x, e1 = xi[1] - yj.y, e2 = xi[2] - yj.z;
    float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;                  // nanoflann.hpp:403-406
    const bool pass = active & (d2 < G.d2_thres);
    const double x = fmax((double)(-d2) * G.inv_den_l, -0.25);
    const float k = (float)((double)G.s2 * exp_poly13(x));
    const float a = ck * k;
    return (pass & (a > G.sp)) ? a : 0.f;
}

// The colour factor of PFN listed pairs (cvo.cpp:171, 173): ck = (float)(c_sigma^2 exp(-d2c / (2.0 c_ell^2))), or NaN for a pair that fails the colour gate
// (a NaN factor never passes a > sp_thres).  exp_neg with its PFN range reductions and polynomials side by side; evaluated once per entry when a list is made.
template <int PFN>
__device__ __forceinline__ void colour_factors(const float (&d2c)[PFN], const Gates& gates, float (&ckv)[PFN]) {
    double kc[PFN], rc[PFN], pc[PFN];
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
        const double xc = (double)(-d2c[u]) * gates.inv_den_c;
        kc[u] = __builtin_rint(xc * 1.44269504088896338700e+00);
        rc[u] = __builtin_fma(kc[u], -1.90821492927058770002e-10, __builtin_fma(kc[u], -6.93147180369123816490e-01, xc));
    }
    exp_poly13_n<PFN>(rc, pc);
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
        const float ckx = (float)((double)gates.csig2 * __builtin_ldexp(pc[u], (int)kc[u]));
        ckv[u] = (d2c[u] < gates.d2c_thres) ? ckx : __builtin_nanf("");
    }
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
    G.inv_den_l = 1.0 / G.den_l;
    G.inv_den_c = 1.0 / G.den_c;
    G.q_il = (float)G.inv_den_l;
    G.q_ic = (float)G.inv_den_c;
    G.q_lim = logf(G.s2 * G.csig2 / P.sp_thres) * 1.001f + 1e-3f;   // a>sp  <=>  d2/den_l + d2c/den_c < ln(s2*csig2/sp)
    G.poly_ok = (double)G.d2_thres * G.inv_den_l <= 0.2499;
    return G;
}

// line-search constants of one iteration (cvo.cpp:241-267), the same in every lane
struct LsConsts {
    float omega[3], v[3];
    float O2[9], O3[9], O4[9], Ov[3], O2v[3], O3v[3];
    float s_beta, s_gamma, s_delta;
};
__device__ __forceinline__ LsConsts make_ls(const float* omega, const float* v, float ell) {
    LsConsts L;
    float Oh[9];
    for (int q = 0; q < 3; ++q) { L.omega[q] = omega[q]; L.v[q] = v[q]; }
    skew3(omega, Oh);
    mat3_mul(Oh, Oh, L.O2); mat3_mul(L.O2, Oh, L.O3); mat3_mul(L.O3, Oh, L.O4);
    mat3_vec(Oh, v, L.Ov); mat3_vec(L.O2, v, L.O2v); mat3_vec(L.O3, v, L.O3v);
    const float temp_coef = (float)(1 / (2.0 * ell * ell));                              // cvo.cpp:267
    L.s_beta = (float)(-2.0 * temp_coef); L.s_gamma = -temp_coef; L.s_delta = (float)(2.0 * temp_coef);
    return L;
}
// x / 6.0 correctly rounded without the division sequence: q = RN(x * RN(1/6)), r = x - 6 q exactly (fma), q + r * RN(1/6)
// (the quotient by a constant c is right whenever q is within an ulp and x / c is no rounding tie, which x / 6 never is)
__device__ __forceinline__ double div6(double x) {
    const double y = 1.0 / 6.0;
    const double q = x * y;
    const double r = __builtin_fma(-6.0, q, x);
    return __builtin_fma(r, y, q);
}
// The line-search terms of one nonzero (cvo.cpp:282-306) in two parts.  ls_point: what depends on the moving point alone
// (cvo.cpp:252-264 and the scalar factors Eigen applies to the rows in :288-297), as four float4:
//   t0 = {s_beta*z1, |z1|^2}   t1 = {2*z2, -z1.z2}   t2 = {-z3, |z2|^2 + 2 z1.z3}   t3 = {2*z4, -}
// ls_pair: the rest, from df = x_i - y_j.  Evaluated per nonzero in one go (ls_terms), or with the last quarter of the point part
// ({2*z4, t2.w}) read from a table made once per iteration (phase_linesearch): the float sequence is the same either way.
template <bool WITH_Z4>
__device__ __forceinline__ void ls_point(const float4 yj, const LsConsts& L, float4& t0, float4& t1, float4& t2, float4& t3) {
    const float y[3] = {yj.x, yj.y, yj.z};
    float z1[3], z2[3], z3[3], t[3];
    cross3(L.omega, y, t); for (int q = 0; q < 3; ++q) z1[q] = t[q] + L.v[q];       // cvo.cpp:254
    mat3_vec(L.O2, y, t);  for (int q = 0; q < 3; ++q) z2[q] = t[q] + L.Ov[q];      // cvo.cpp:255-256
    mat3_vec(L.O3, y, t);  for (int q = 0; q < 3; ++q) z3[q] = t[q] + L.O2v[q];     // cvo.cpp:257-258
    const float nrm = dot3_seq(z1, z1);                                            // cvo.cpp:261
    const float mdot = -dot3_seq(z1, z2);                                          // cvo.cpp:262
    t0 = make_float4(L.s_beta * z1[0], L.s_beta * z1[1], L.s_beta * z1[2], nrm);   // the row factors of cvo.cpp:288, 290, 293, 296
    t1 = make_float4(2.f * z2[0], 2.f * z2[1], 2.f * z2[2], mdot);
    t2 = make_float4(-z3[0], -z3[1], -z3[2], 0.f);
    if (WITH_Z4) {
        float z4[3];
        mat3_vec(L.O4, y, t);  for (int q = 0; q < 3; ++q) z4[q] = t[q] + L.O3v[q]; // cvo.cpp:259-260
        t2.w = dot3_seq(z2, z2) + 2 * dot3_seq(z1, z3);                            // cvo.cpp:263
        t3 = make_float4(2.f * z4[0], 2.f * z4[1], 2.f * z4[2], 0.f);
    }
}
__device__ __forceinline__ void ls_pair(const float (&df)[3] /* x_i - y_j, cvo.cpp:286 */, float A_ij, const float4 t0, const float4 t1, const float4 t2, const float4 t3,
                                        const LsConsts& L, double& Bi, double& Ci, double& Di, double& Ei) {
    const float beta_ij = sum3f(t0.x * df[0], t0.y * df[1], t0.z * df[2]);                        // cvo.cpp:288
    const float gamma_ij = L.s_gamma * (t0.w + sum3f(t1.x * df[0], t1.y * df[1], t1.z * df[2]));  // cvo.cpp:290-291
    const float delta_ij = L.s_delta * (t1.w + sum3f(t2.x * df[0], t2.y * df[1], t2.z * df[2]));  // cvo.cpp:293-294
    const float epsil_ij = L.s_gamma * (t2.w + sum3f(t3.x * df[0], t3.y * df[1], t3.z * df[2]));  // cvo.cpp:296-297
#ifdef CVO_LS_FULL      // the brackets operation by operation as cvo.cpp:301-305 writes them (experiment builds; the default is the condensed form below)
    Bi += double(A_ij * beta_ij);                                                                                              // cvo.cpp:301
    Ci += double(A_ij * (gamma_ij + beta_ij * beta_ij / 2.0));                                                                 // cvo.cpp:302
    Di += double(A_ij * (delta_ij + beta_ij * gamma_ij + div6((double)(beta_ij * beta_ij * beta_ij))));                        // cvo.cpp:303
    Ei += double(A_ij * (epsil_ij + beta_ij * delta_ij + 1 / 2.0 * beta_ij * beta_ij * gamma_ij                                // cvo.cpp:304-305
                         + 1 / 2.0 * gamma_ij * gamma_ij + 1 / 24.0 * beta_ij * beta_ij * beta_ij * beta_ij));
#else
    // The same four terms with the double-precision part of the brackets condensed (cvo.cpp:301-305).  Everything the reference rounds to FLOAT stays as
    // it is (A*beta, beta*beta, delta + beta*gamma, beta*beta*beta, epsil + beta*delta: each a float product or sum converted afterwards); what it evaluates
    // in DOUBLE -- because of the 2.0, 6.0, 24.0 literals -- is a polynomial in beta, gamma whose value these lines give to within 2-3 ulps of a double:
    //   q = beta^2 (exact in double: two 24-bit factors);  gamma + bb/2 = fma(0.5, bb, gamma) (0.5 bb is exact: the same bits as the reference's sum);
    //   bbb/6.0 as a product with RN(1/6);  1/2 q gamma + 1/2 gamma^2 + q^2/24 = 1/2 gamma (q + gamma) + q^2 RN(1/24);  term and sum fused.
    // B..E are sums of ~1e4 such terms in an order the reference leaves to its threads (cvo.cpp:309-314), rounded to float before the cubic (cvo.cpp:318): an
    // ulp of a double in a term is to the result what the order of the sum is (the oracle's shuffled-order variants: no pose bit moves).  12 double-precision
    // instructions per nonzero instead of 27; tests/test_gpu_config3.py holds the poses of all 64 pairs at 0.0 / 0.0.
    const double Ad = (double)A_ij, bd = (double)beta_ij, gd = (double)gamma_ij;
    const double q = bd * bd;
    Bi += double(A_ij * beta_ij);                                                                                              // cvo.cpp:301
    Ci = __builtin_fma(Ad, __builtin_fma(0.5, (double)(beta_ij * beta_ij), gd), Ci);                                           // cvo.cpp:302
    Di = __builtin_fma(Ad, __builtin_fma((double)(beta_ij * beta_ij * beta_ij), 1.0 / 6.0, (double)(delta_ij + beta_ij * gamma_ij)), Di);   // cvo.cpp:303
    double e = __builtin_fma(0.5, gd * (q + gd), (double)(epsil_ij + beta_ij * delta_ij));                                     // cvo.cpp:304-305
    e = __builtin_fma(q * q, 1.0 / 24.0, e);
    Ei = __builtin_fma(Ad, e, Ei);
#endif
}
// one nonzero of A: adds its B, C, D, E terms (cvo.cpp:282-306)
__device__ __forceinline__ void ls_terms(const float* xi, const float4 yj, float A_ij, const LsConsts& L, double& Bi, double& Ci, double& Di, double& Ei) {
    float4 t0, t1, t2, t3;
    ls_point<true>(yj, L, t0, t1, t2, t3);
    const float df[3] = {xi[0] - yj.x, xi[1] - yj.y, xi[2] - yj.z};                // cvo.cpp:286
    ls_pair(df, A_ij, t0, t1, t2, t3, L, Bi, Ci, Di, Ei);
}

// ---------------------------------------------------------------- S: dense cull
// 7 VALU per pair test: 3 sub, 3 fma (the last one folds "- thr" in, so the sign bit of
// t = d2 - thr is the hit) and one v_alignbit that shifts the sign into the row's 32-column
// word, w = (w << 1) | sign(t): the first column of a group ends up in bit 31.
// (Packed v_pk_*_f32 forms were measured: they issue at half the rate, no gain.)
// Columns come from the LDS tile 4 at a time (ds_read_b128, the same address in every lane:
// broadcast); a lane carries SWEEP_R rows (consecutive 64-row blocks) so one read feeds
// SWEEP_R tests.
constexpr int SWEEP_R = 2;
#ifndef CVO_LS_TAB_FACTOR
#define CVO_LS_TAB_FACTOR 4        // phase_linesearch: the per-column table is made when the workgroup has more than this many nonzeros per column
#endif
#ifndef CVO_LS_EVEN_MIN
#define CVO_LS_EVEN_MIN 512        // ... when the workgroup has at least this many records per wave
#endif
#ifndef CVO_LS_EVEN
#define CVO_LS_EVEN 1              // phase_linesearch: every wave an equal share of the workgroup's nonzero records (0 = the segment it compacted itself)
#endif
#ifndef CVO_PAIR_XMAX
#define CVO_PAIR_XMAX 1            // phase_cull: the farthest row that bounds later list margins is the pair's, not the member's (members rebuild their lists together)
#endif
#ifndef CVO_CULL_SPLIT
#define CVO_CULL_SPLIT 1           // phase_cull: single 64-row blocks as units of work when block pairs are scarce (one pair on many workgroups)
#endif

template <int NR = SWEEP_R>                                    // NR = 1: the lane's first row only (phase_cull's single-block units); w[1] is left at 0
__device__ __forceinline__ void sweep_group(const float* lx, const float* ly, const float* lz, int c0 /* wave-uniform */,
                                            const float (&x)[SWEEP_R][3], const float (&nthr)[SWEEP_R], uint32_t (&w)[SWEEP_R]) {
    const float4* qx = reinterpret_cast<const float4*>(lx + c0);
    const float4* qy = reinterpret_cast<const float4*>(ly + c0);
    const float4* qz = reinterpret_cast<const float4*>(lz + c0);
#pragma unroll
    for (int r = 0; r < SWEEP_R; ++r) w[r] = 0u;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 X = qx[q], Y = qy[q], Z = qz[q];
        const float cx[4] = {X.x, X.y, X.z, X.w}, cy[4] = {Y.x, Y.y, Y.z, Y.w}, cz[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float dx = x[r][0] - cx[u], dy = x[r][1] - cy[u], dz = x[r][2] - cz[u];
                const float t = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, nthr[r])));
                w[r] = __builtin_amdgcn_alignbit(w[r], __float_as_uint(t), 31);
            }
        }
    }
}

__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, __shfl_xor(v, 32, 64)); v = fminf(v, __shfl_xor(v, 16, 64));
    v = fminf(v, dpp_xor<8>(v)); v = fminf(v, dpp_xor<4>(v)); v = fminf(v, dpp_xor<2>(v)); v = fminf(v, dpp_xor<1>(v));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 32, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, dpp_xor<8>(v)); v = fmaxf(v, dpp_xor<4>(v)); v = fmaxf(v, dpp_xor<2>(v)); v = fmaxf(v, dpp_xor<1>(v));
    return v;
}

// ---------------------------------------------------------------- the kernel, phase by phase
// Every phase is its own non-inlined function: each gets its own register allocation (the whole
// iteration as one function spilled ~150 VGPRs into the candidate loop), and nothing but a few
// scalars lives across phases -- the iteration state is in LDS (Shared).  Arguments arrive in
// VGPRs; they are wave-uniform and are moved back to SGPRs (readfirstlane) first thing.
extern __shared__ __attribute__((aligned(16))) unsigned char cvo_smem[];

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Issue priority of this wave by what it has left to do (r = units of work behind the current one; wave-uniform).  A SIMD issues for its OLDEST ready wave first:
// of the two or three waves that share one, the oldest runs ahead and finishes early, and the youngest walks its last blocks alone -- a single wave fills
// little more than half of the SIMD's issue slots.  With the wave that has more left ranked higher, the waves of a SIMD reach their (short) last blocks
// together: the walk of one pair alone ends 5-8 % earlier (profiles/r05_wave_priority_ab.txt).  With every CU busy the gain is gone (the lone wave's
// stalls are memory stalls then): on in the three-wave build only.  s_setprio takes an immediate.
#ifndef CVO_WAVE_PRIO
#define CVO_WAVE_PRIO (CVO_BLOCK_MAX > 512)   // measured: one pair alone walks 5-8 % faster either way; under load +0.6 % with three waves per SIMD (config 5), -0.7 % with two
#endif
__device__ __forceinline__ void prio_by_remaining(int r) {
#if CVO_WAVE_PRIO
    if (r >= 3) __builtin_amdgcn_s_setprio(3); else if (r == 2) __builtin_amdgcn_s_setprio(2); else if (r == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ float uni_f(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
template <class T> __device__ __forceinline__ T* uni_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// the depth-proportional list margin at the present ell: skin_alpha at the first ell (0.15, cvo.cpp:35), falling with ell to the power alpha_gamma -- the cloud moves less
// from iteration to iteration as the alignment converges, and at ell = 0.03 a margin of 0.0125 |x| is as large as the radius itself (DevParams::alpha_gamma)
__device__ __forceinline__ float list_alpha(const DevParams& P, float ell) {
    return fminf(P.alpha_gamma > 0.f ? P.skin_alpha * __powf(fminf(ell * (1.0f / 0.15f), 1.0f), P.alpha_gamma) : P.skin_alpha, 0.5f);
}
constexpr int PF = 4;                 // list entries a lane evaluates side by side
constexpr int NCLS = 128;             // list-length classes (ceil(len / PF), the last one open-ended) the rows are sorted by
// The transformed moving cloud lives in LDS as float4 {y, g0} (mode 1), as three float planes when that does not fit but
// 12 bytes a point do (mode 2, e.g. 9 k-point clouds), or in HBM/L2 (mode 0).
struct Lds {
    Shared* sh; uint16_t* lenS; uint16_t* row_of; uint16_t* rowlen; int* hist; int* base; float* gbox; float* lx; float* ly; float* lz; float4* ylds;
    float* ysx; float* ysy; float* ysz;
    float4* tab;
};
// Shared | list length per slot | row of a slot | cull tile, SoA (between culls: the fixed points by slot) | resident transformed moving
// cloud (optional) | scratch of the list rebuild: list length per local row, sort histograms, group boxes (8 planes).  The line-search
// table (16 bytes per column) is laid over the scratch, which is dead outside the rebuild.  The table and cloud sizes are launch
// parameters kept in Shared.
// tgeo packs the launch's LDS geometry into one scalar every phase receives: tile | rows_cap/64 << 13 | y_cap/64 << 20
__device__ __forceinline__ int pack_geometry(int tile, int rows_cap, int y_cap) { return tile | ((rows_cap >> 6) << 13) | ((y_cap >> 6) << 20); }
__device__ __forceinline__ Lds lds_layout(int tgeo, int y_mode) {
    Lds L;
    L.sh = reinterpret_cast<Shared*>(cvo_smem);
    const int tile = tgeo & 0x1FFF, rows_cap = ((tgeo >> 13) & 0x7F) << 6, y_cap = ((tgeo >> 20) & 0x7FF) << 6;
    L.lenS = reinterpret_cast<uint16_t*>(cvo_smem + ((sizeof(Shared) + 15) & ~size_t(15)));
    L.row_of = L.lenS + rows_cap;
    L.lx = reinterpret_cast<float*>(L.row_of + rows_cap);
    L.ly = L.lx + tile; L.lz = L.ly + tile;
    L.ylds = reinterpret_cast<float4*>(L.lz + tile);
    L.ysx = reinterpret_cast<float*>(L.ylds); L.ysy = L.ysx + y_cap; L.ysz = L.ysy + y_cap;
    float* behind = L.ysx + (y_mode == 1 ? 4 * (size_t)y_cap : (y_mode == 2 ? 3 * (size_t)y_cap : 0));
    L.tab = reinterpret_cast<float4*>(behind);
    // mode 2 (12-byte planes: large clouds): the cull reads its columns straight from the planes, so the tile is idle during a
    // rebuild and holds the scratch itself -- the tile region then only has to be large enough for the fixed points by slot
    float* scratch = (y_mode == 2) ? L.lx : behind;
    L.gbox = scratch;                                                // 8 planes of (tile/32) floats, or of (y_cap/32) in mode 2
    L.hist = reinterpret_cast<int*>(L.gbox + 8 * (size_t)((y_mode == 2 ? y_cap : tile) >> 5));
    L.base = L.hist + MAX_WAVES * NCLS;
    L.rowlen = reinterpret_cast<uint16_t*>(L.base + MAX_WAVES * NCLS);
    return L;
}

// one pair as seen by workgroup g of its G: the fixed cloud is cut into blocks of ROW_DEAL consecutive rows (scan order:
// a thin slab of the image, so the cull's boxes stay tight) and the blocks are dealt round-robin (near surfaces have many
// more neighbours per row than far ones: whole bands of the image per workgroup would be unbalanced)
#ifndef CVO_ROW_DEAL
#define CVO_ROW_DEAL 128
#endif
constexpr int ROW_DEAL = CVO_ROW_DEAL;
struct Ctx {
    const gfloat* fixed; const gfloat* moving;
    int nf, nm, nrows, rows_per, rows_pad, capn, nm_pad, flat_cap, g, G;
    GF4 ybuf; gv2u* surv;
    gv2u* jT4; gv2u* ent; gu64* xch;
    size_t fbase;
    TraceRow* trace; int* trace_len; int trace_cap;                 // optional per-iteration trace (PairDesc)
};
__device__ __forceinline__ void pair_rows(int nf, int g, int G, int& rows_per, int& nrows) {
    const int nblocks = (nf + ROW_DEAL - 1) / ROW_DEAL;
    rows_per = ((nblocks + G - 1) / G) * ROW_DEAL;                  // the most rows any workgroup of the pair owns
    const int mine = (g < nblocks) ? (nblocks - g + G - 1) / G : 0; // blocks g, g + G, g + 2G, ...; only the cloud's last block may be short
    nrows = mine * ROW_DEAL - ((mine > 0 && (nblocks - 1) % G == g) ? nblocks * ROW_DEAL - nf : 0);
}
__device__ __forceinline__ Ctx build_ctx(const PairDesc* Dp, int g, int G) {
    const PairDesc& D = *Dp;
    Ctx c;
    c.g = g; c.G = G;
    c.nf = D.nf; c.nm = D.nm; c.nm_pad = D.nm_pad; c.capn = D.capn;
    // rows_per / nrows of this workgroup: worked out once per pair (pair_rows: three divisions by G) and kept in Shared -- every
    // phase builds its own Ctx
    const Shared* shc = reinterpret_cast<const Shared*>(cvo_smem);
    c.rows_per = shc->ctx_rows_per; c.nrows = shc->ctx_nrows;
    c.rows_pad = (c.rows_per + 127) & ~127;                         // = PairDesc::rows_pad for the launch's own G; smaller when a helper has joined (two workgroups share the slot's buffers)
    c.fixed = (const gfloat*)D.fixed; c.moving = (const gfloat*)D.moving;
    const size_t ws = (size_t)shc->ws_slot;                         // the launch's pair slot whose work buffers this workgroup uses
    c.ybuf = GF4{(gv4f*)D.ybuf + ws * (size_t)D.ws_y_stride + (size_t)g * D.nm_pad};
    c.surv = (gv2u*)D.surv + ws * (size_t)D.ws_surv_stride;
    // Where member g's part of the slot's buffers starts, in rows.  A launch with a fixed G cuts the slot into G equal parts.  A launch
    // whose pairs can gain members on the way (adoption) gives member g a region of its own, sized for the rows it owns when it joins
    // (g + 1 members): a region is then only ever written and read by one workgroup -- the L2 caches of the XCDs are not coherent with
    // each other for plain loads and stores inside a kernel, so a region handed from one workgroup to another could be overwritten by the
    // first one's late write-backs.
    size_t off_rows = (size_t)g * c.rows_pad, off_recs = (size_t)g * c.rows_per;
    if (D.member_regions) {
        const int nblocks = (D.nf + ROW_DEAL - 1) / ROW_DEAL;
        off_rows = 0;
        for (int j = 0; j < g; ++j) off_rows += (size_t)((nblocks + j) / (j + 1)) * ROW_DEAL;   // rows_per at j + 1 members (a multiple of 128)
        off_recs = off_rows;
    }
    c.jT4 = (gv2u*)D.jT + (ws * (size_t)D.ws_list_stride) / 4 + off_rows * (D.capn / 4);   // the cull's lists: four 16-bit columns per 8-byte word, word q of local row li at [q][li]
    c.ent = (gv2u*)D.ent + ws * (size_t)D.ws_list_stride + off_rows * D.capn;
    c.xch = (gu64*)D.xch;
    c.fbase = off_recs * D.capf;
    c.flat_cap = c.rows_per * D.capf;
    c.trace = D.trace; c.trace_len = D.trace_len; c.trace_cap = D.trace_cap;
    return c;
}

static_assert(sizeof(Ctx) <= sizeof(((Shared*)nullptr)->ctx_store), "Shared::ctx_store holds a Ctx");
// one thread, after Shared::ctx_rows_per / ctx_nrows / ws_slot are set for the pair (or changed); a workgroup barrier follows at the call sites
__device__ __forceinline__ void store_ctx(const PairDesc* Dp, int g, int G) {
    Shared* shc = reinterpret_cast<Shared*>(cvo_smem);
    const Ctx c = build_ctx(Dp, g, G);
    *reinterpret_cast<Ctx*>(shc->ctx_store) = c;
}
__device__ __forceinline__ Ctx make_ctx(const PairDesc*, int, int) { return *reinterpret_cast<const Ctx*>(reinterpret_cast<const Shared*>(cvo_smem)->ctx_store); }

// fixed-cloud row of a workgroup's local row
__device__ __forceinline__ int global_row(const Ctx& c, int li) { return (c.g + c.G * (li / ROW_DEAL)) * ROW_DEAL + (li % ROW_DEAL); }

// fixed point of a slot: from the LDS copy made after the sort, or gathered from the cloud
__device__ __forceinline__ void load_x(const Ctx& c, const Lds& L, bool x_lds, int slot, float (&xi)[3]) {
    if (x_lds) { xi[0] = L.lx[slot]; xi[1] = L.ly[slot]; xi[2] = L.lz[slot]; }
    else { const float4 lo = ld4(c.fixed + lo_off(global_row(c, (int)L.row_of[slot]))); xi[0] = lo.x; xi[1] = lo.y; xi[2] = lo.z; }
}

// transformed moving point j; .w = first feature channel in modes 0 and 1, not available (0) in mode 2
template <int YM>
__device__ __forceinline__ float4 load_y(const Ctx& c, const Lds& L, int j) {
    if (YM == 1) return L.ylds[j];
    if (YM == 2) return make_float4(L.ysx[j], L.ysy[j], L.ysz[j], 0.f);
    return c.ybuf[j];
}
__device__ __forceinline__ float4 load_y_rt(const Ctx& c, const Lds& L, int y_mode, int j) {
    return y_mode == 1 ? load_y<1>(c, L, j) : (y_mode == 2 ? load_y<2>(c, L, j) : load_y<0>(c, L, j));
}

// ---- T: transform_pcd (cvo.cpp:336-341) into LDS (or ybuf); how far has any point moved since the candidate lists were
// built (exact displacement of the very positions the tests use); rebuild decision.  The first PRE_T points of a thread may
// arrive pre-loaded (the epilogue of the previous iteration fetches them while one lane does the scalar work).
constexpr int PRE_T = 4096 / BLOCK_MAX;
#ifndef CVO_FUSE_PLANES
#define CVO_FUSE_PLANES 1          // the epilogue's fused transform (begun before the second stop test is done, lane 0 off its critical path) also in the plane layout of large clouds
#endif
// first_worker = 64: wave 0 takes no points (the epilogue's lane 0 is still busy with the stop test of the iteration when the others start): the points are dealt
// to threads first_worker .. nthreads - 1, `pre` as the caller loaded it with the same deal.  keep_M_on_stop: the transform was started before the stop test was
// known; if the iteration turns out to be the pair's last, cvo::transform of that iteration (Shared::M, cvo.cpp:815) stays.
// preb (may be null): for the pre-loaded points, {where the point was listed (Mb p), alpha_build x its distance from the camera there}: that half of the
// staleness test does not depend on the new pose, the epilogue works it out while lane 0 is at the step and the pose update.
// v_sqrt_f32 as it is (1 ulp; sqrtf is the IEEE sequence around it, ten instructions more): for the staleness bounds, which carry a margin of 1e-4
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
template <int YM>
__device__ __forceinline__ void transform_body_t(const Ctx& c, const Lds& L, Shared* sh, const float4 (&pre)[PRE_T], bool have_pre, int first_worker = 0, bool keep_M_on_stop = false,
                                                 bool have_preb = false, const float4* preb = nullptr) {
    const int tid = threadIdx.x, nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int wstride = nthreads - first_worker;
    // first_worker > 0 (the epilogue's call): the threads below it have no points -- wave 0, whose lane 0 comes from the second stop test and is the one everybody
    // would wait for.  It touches nothing here that it does not need: the transform is made (and kept for the next phases) by the workers.
    const bool worker = tid >= first_worker;
    const bool have_list = sh->list_valid != 0;
    float M[12], Mb[12];
    if (worker) {
        float R[9], T[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = sh->R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) T[i] = sh->T[i];
        make_transform(R, T, M);                                    // update_tf, cvo.cpp:770
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) M[i] = 0.f;
    }
    if (worker && have_list && (!have_preb || c.nm > PRE_T * wstride)) {   // (the points behind the pre-loaded ones have no pre-computed list position)
#pragma unroll
        for (int i = 0; i < 12; ++i) Mb[i] = sh->Mb[i];
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) Mb[i] = 0.f;
    }
    // how much closer a pair can be now than when the lists were built: the largest displacement of a point since then, less the part of it the lists of
    // its neighbourhood allow for by themselves (alpha_build x its distance from the camera at build time; 0 with one margin for all rows)
    float dmax = 0.f;
    const float alpha_b = sh->alpha_build;
    auto one = [&](int j, const float4 lo, bool have_b, const float4 bq) {   // (have_b a flag of its own, not a null test of a pointer to a local array: see cand_steady)
        float y0, y1, y2;
        apply_transform(M, lo.x, lo.y, lo.z, y0, y1, y2);
        if (YM == 1) L.ylds[j] = make_float4(y0, y1, y2, lo.w);     // the cloud stays in LDS for the whole iteration ...
        else if (YM == 2) { L.ysx[j] = y0; L.ysy[j] = y1; L.ysz[j] = y2; }
        else c.ybuf.set(j, make_float4(y0, y1, y2, lo.w));          // ... or, too large for that, in HBM/L2
        if (have_list) {                                            // where the point was when the lists were built: the same arithmetic, then
            float b0, b1, b2, afar;
            if (have_b) { b0 = bq.x; b1 = bq.y; b2 = bq.z; afar = bq.w; }
            else {
                apply_transform(Mb, lo.x, lo.y, lo.z, b0, b1, b2);
                afar = alpha_b * (fast_sqrt(__builtin_fmaf(b2, b2, __builtin_fmaf(b1, b1, b0 * b0))) * 0.9999f);
            }
            const float e0 = y0 - b0, e1 = y1 - b1, e2 = y2 - b2;
            const float disp = fast_sqrt(__builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0))) * 1.0001f + 1.0e-5f;
            dmax = fmaxf(dmax, disp - afar);
        }
    };
    int j = worker ? tid - first_worker : c.nm;
    if (have_pre) {
#pragma unroll
        for (int u = 0; u < PRE_T; ++u) { if (j < c.nm) one(j, pre[u], have_preb, have_preb ? preb[u] : make_float4(0.f, 0.f, 0.f, 0.f)); j += wstride; }
    }
    for (; j < c.nm; j += wstride) one(j, ld4(c.moving + lo_off(j)), false, make_float4(0.f, 0.f, 0.f, 0.f));
#if defined(CVO_KTRACE_EPI) && CVO_KTRACE_EPI == 2
    if (tid == 64) { __builtin_amdgcn_sched_barrier(0); sh->kabs[5] = CVO_NOW() + (dmax == 12345.f ? 1 : 0); }
#endif
    // What lane 0's decision needs besides the maximum is in its registers before the barrier; the new transform is kept for the next phases by the first worker
    // (in Shared::Mn when lane 0's second stop test may still fire: Shared::M stays the transform of the last executed iteration then).
    float r_c = 0.f, Rb_l = 0.f, ell_b = 0.f, skin_l = 0.f, ell = 0.f;
    int dense_l = 0;
    if (tid == 0) {
        ell = sh->ell; Rb_l = sh->Rb; ell_b = sh->ell_build; skin_l = sh->P.skin; dense_l = sh->dense_mode;
        if (sh->rc_ell == ell) r_c = sh->rc;                        // the gate radius of this ell (a double-precision log): once per ell, not per iteration
        else {
            r_c = sqrtf(gate_d2_align(ell, sh->P.sp_thres, sh->P.sigma * sh->P.sigma)); sh->rc = r_c; sh->rc_ell = ell;
            *reinterpret_cast<Gates*>(sh->gates_store) = make_gates(ell, sh->P); sh->inv_c = 1 / sh->P.c; sh->inv_d = 1 / sh->P.d;   // (read behind this function's barriers)
        }
    }
    if (tid == first_worker) {
        float* Mdst = keep_M_on_stop ? sh->Mn : sh->M;
#pragma unroll
        for (int i = 0; i < 12; ++i) Mdst[i] = M[i];
    }
    // the workgroup's maximum with ONE barrier: only lane 0 needs it, and the barrier that publishes its decision closes the phase anyway
    if (worker) { dmax = wave_max_nonneg(dmax); if ((tid & 63) == 0) sh->fred[tid >> 6] = dmax; }
    else if ((tid & 63) == 0) sh->fred[tid >> 6] = 0.f;
#if defined(CVO_KTRACE_EPI) && CVO_KTRACE_EPI == 2
    if (tid == 0) { __builtin_amdgcn_sched_barrier(0); sh->kabs[8] = CVO_NOW(); }
    if (tid == 64) { __builtin_amdgcn_sched_barrier(0); sh->kabs[9] = CVO_NOW(); }
#endif
    __syncthreads();                                                // (also makes ybuf / ylds visible to the workgroup)
#if defined(CVO_KTRACE_EPI) && CVO_KTRACE_EPI == 2
    if (tid == 0) { __builtin_amdgcn_sched_barrier(0); sh->kabs[2] = CVO_NOW() + (dmax == 12345.f ? 1 : 0); }
#endif
    if (keep_M_on_stop && tid < 12) { if (!sh->stop) sh->M[tid] = sh->Mn[tid]; }   // (wave 0: it sees lane 0's sh->stop, LDS operations of one wave complete in order)
    if (tid == 0) {
        unsigned mx = 0u;
#pragma unroll
        for (int w = 0; w < MAX_WAVES; ++w) { const unsigned f = __float_as_uint(sh->fred[w]); mx = max(mx, w < nwaves ? f : 0u); }   // (>= 0: the bits order like the numbers)
        dmax = __uint_as_float(mx);
        // Row i's list holds every column within (Rb + a |x_i|) / (1 - a) of it at the build positions (a = alpha_build).  A pair within r_c now was
        // within D < r_c + (its point's displacement) <= r_c + reach + a |y_j| <= r_c + reach + a (|x_i| + D) then: inside the list while
        // r_c + reach <= Rb.  (a = 0: the lists hold every pair within Rb, reach = the largest displacement.)
        const float reach = have_list ? dmax : 1.0e-5f;
        sh->reach_now = reach;
        int rb = (!have_list || (ell_b != ell) || (r_c + reach) * 1.00001f > Rb_l) ? 1 : 0;
        // ell has dropped (cvo.cpp:810-812) and the old, wider lists still hold every pair within the NEW list radius of the
        // current positions: filter them in place instead of a dense cull (2 = refine).  Not in dense mode (no lists to filter).
        // With new radii (Rn + a' |x_i|) / (1 - a') =: Rn_i the old lists must hold D < Rn_i + reach + a (|x_i| + D), i.e. Rn_i + reach <= Rb for the
        // farthest row: that bounds the depth-proportional margin a' the filtered lists can have.
        const float Rn = r_c * (1.0f + skin_l);
        if (rb && have_list && !dense_l && ell_b != ell && (Rn + reach) * 1.00001f <= Rb_l) {
            rb = 2;
            sh->reach = reach;                                      // phase_refine works the margins of the filtered lists out from it
        }
        sh->rebuild = rb;
#if defined(CVO_KTRACE_EPI) && CVO_KTRACE_EPI == 2
        __builtin_amdgcn_sched_barrier(0); sh->kabs[3] = CVO_NOW() + (sh->rebuild == 12345 ? 1 : 0);
#endif
    }
    __syncthreads();
}
// the layouts of large clouds (12-byte LDS planes, HBM): out of line, nothing pre-loaded -- keeps the common path's code small
static __device__ __noinline__ void transform_large(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), y_lds = uni(y_lds_in);
    const Lds L = lds_layout(tgeo, y_lds);
    const Ctx c = make_ctx(Dp, g, G);
    float4 none[PRE_T];
#pragma unroll
    for (int u = 0; u < PRE_T; ++u) none[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y_lds == 2) transform_body_t<2>(c, L, L.sh, none, false); else transform_body_t<0>(c, L, L.sh, none, false);
}
CVO_PHASE_FN(4) void phase_transform(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), y_lds = uni(y_lds_in);
    const Lds L = lds_layout(tgeo, y_lds);
    const Ctx c = make_ctx(Dp, g, G);
    if (y_lds != 1) { transform_large(Dp, g, G, tgeo, y_lds); return; }
    float4 none[PRE_T];
#pragma unroll
    for (int u = 0; u < PRE_T; ++u) none[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    transform_body_t<1>(c, L, L.sh, none, false);
}

// ---- S: dense cull straight into the transposed candidate lists.
// A wave owns pairs of consecutive 64-row blocks (a lane = one row of each) and walks ALL columns for them, so a row's
// hits are found in ascending column order (= the CSR order of Eigen::setFromTriplets, cvo.cpp:182) and can be appended to
// the row's list jT[n][row] on the spot: no bitmap, no scan, no second pass.  Only 32-column groups whose bounding box
// comes within the cull radius of the wave's rows' bounding box are tested at all (one lane per group decides, a ballot
// turns the decisions into a scalar mask the wave then iterates).
static __device__ __noinline__ void phase_cull(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), tile = tgeo & 0x1FFF, y_lds = uni(y_lds_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh; uint16_t* rowlen = L.rowlen;
    const Ctx c = make_ctx(Dp, g, G);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int nrows = c.nrows;
    const float r_c = sqrtf(gate_d2_align(sh->ell, sh->P.sp_thres, sh->P.sigma * sh->P.sigma));
    // the pair's first lists (no twist yet to extrapolate the path with, and the cloud moves fastest in its first iterations) take `first_scale` times the margins
    const float mscale = sh->twist_ok ? 1.0f : sh->P.first_scale;
    const float Rb = r_c * (1.0f + sh->P.skin * mscale);
    // row i is listed with radius (Rb + alpha |x_i|) / (1 - alpha) (DevParams::skin_alpha; phase_transform's staleness test is its counterpart)
    // (alpha well below 1: the row radius (Rb + alpha |x|) / (1 - alpha) and the staleness proof need 1 - alpha > 0; only experiment knobs could push it there)
    const float alpha = fminf(list_alpha(sh->P, sh->ell) * mscale, 0.5f), inv_1ma = 1.0f / (1.0f - alpha);
    float xmax_l = 0.f;
    // The lists are built around where the cloud is HEADING, not where it is: a list stays valid while every point is within its allowance
    // (skin r + alpha |b_j|) of the position b_j it was listed at, so with b_j a stretch ahead on the path the same radius covers up to twice the
    // travel.  The path is extrapolated with the previous iteration's twist (the pose update of cvo.cpp:793-801 applied once more with a longer
    // step), as far as `predict` of the allowance of every point: y moves by about t (omega x y + v), |.| <= t (|omega| |y| + |v|), so
    // t = predict * min(skin r / |v|, alpha / |omega|) keeps every point inside.  Whether it really does is checked below, point by point, with
    // the staleness test's own arithmetic; if not, the lists are built at the current positions after all.
    float Mp[12];
    bool predicted = false;
    if (sh->P.predict > 0.f && sh->twist_ok && y_lds != 2 && sh->ell_build == sh->ell) {
        float om[3], vv[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { om[q] = sh->omega[q]; vv[q] = sh->v[q]; }
        const float wn = norm3f(om), vn = norm3f(vv);
        const float beta = r_c * sh->P.skin;
        float t = sh->P.predict * fminf(vn > 1.0e-12f ? beta / vn : 1.0e9f, wn > 1.0e-12f ? alpha / wn : 1.0e9f);
        t = fminf(t, sh->P.predict_steps * sh->step);               // no further ahead than a few of the last iteration's steps
        if (t > 0.f && t < 1.0e8f) {
            float dR[9], dT[3], R[9], T[3], RdT[3], Rn[9], Tn[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = sh->R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) T[i] = sh->T[i];
            exp_sek3(om, vv, t, dR, dT);
            mat3_vec(R, dT, RdT);
#pragma unroll
            for (int q = 0; q < 3; ++q) Tn[q] = RdT[q] + T[q];
            mat3_mul(R, dR, Rn);
            make_transform(Rn, Tn, Mp);
            predicted = true;
        }
    }
#ifdef CVO_KTRACE_CULL
    if (tid == 0) sh->kabs[0] = CVO_NOW();
#endif
    const bool planes = y_lds == 2;                                 // columns come straight from the resident y planes: one pass over the whole cloud
    const int span = planes ? max(c.nm, 1) : tile;
    const float* colx = planes ? L.ysx : L.lx; const float* coly = planes ? L.ysy : L.ly; const float* colz = planes ? L.ysz : L.lz;
    const int gplane = (planes ? (((tgeo >> 20) & 0x7FF) << 6) : tile) >> 5;   // stride of the eight bounding-box planes: lo/hi of x, y, z and of y/z
    const int nblk2 = (nrows + 64 * SWEEP_R - 1) / (64 * SWEEP_R);  // row-block pairs of this workgroup
    const float INF = __builtin_inff();
    if (span < c.nm) predicted = false;                             // (the check below looks at the whole cloud in one tile)
    bool again;
    do {                                                            // a second trip only when the extrapolated positions turn out to be too far from the current ones
    again = false;
    float off_l = 0.f;
    for (int t0 = 0; t0 < c.nm; t0 += span) {
        const int tn = min(span, c.nm - t0);
        const int tnp = (tn + 31) & ~31;
        __syncthreads();                                            // previous tile fully consumed
        for (int jj = tid; jj < tnp; jj += nthreads) {              // a wave's half = one 32-column group
            float4 y = make_float4(FAR_COL, FAR_COL, FAR_COL, 0.f);
            // box of the group in x, y, z and in the ray slope y/z: clouds come in image scan order, 32 consecutive points
            // cover the whole width and depth range but only a few image rows
            float lo[4] = {INF, INF, INF, INF}, hi[4] = {-INF, -INF, -INF, -INF};
            if (jj < tn) {
                y = load_y_rt(c, L, y_lds, t0 + jj);
                if (predicted) {
                    const float4 p = ld4(c.moving + lo_off(t0 + jj));
                    float b0, b1, b2;
                    apply_transform(Mp, p.x, p.y, p.z, b0, b1, b2);
                    const float e0 = y.x - b0, e1 = y.y - b1, e2 = y.z - b2;
                    const float disp = sqrtf(__builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0))) * 1.0001f + 1.0e-5f;
                    const float far = sqrtf(__builtin_fmaf(b2, b2, __builtin_fmaf(b1, b1, b0 * b0))) * 0.9999f;
                    off_l = fmaxf(off_l, disp - alpha * far);       // phase_transform's test of the lists about to be built, at the current positions
                    y.x = b0; y.y = b1; y.z = b2;
                }
                lo[0] = hi[0] = y.x; lo[1] = hi[1] = y.y; lo[2] = hi[2] = y.z;
                if (y.z > 1.0e-3f) { lo[3] = hi[3] = y.y / y.z; } else { lo[3] = -INF; hi[3] = INF; }   // behind / at the camera: no slope bound
            }
            if (!planes) { L.lx[jj] = y.x; L.ly[jj] = y.y; L.lz[jj] = y.z; }
            else if (jj >= tn) { L.ysx[jj] = FAR_COL; L.ysy[jj] = FAR_COL; L.ysz[jj] = FAR_COL; }   // padding columns of the last group (the planes have room: y_cap is a multiple of 64)
#pragma unroll
            for (int q = 0; q < 4; ++q) {                           // min / max over each half wave (a 32-column group): one LDS exchange, the rest in the vector pipe
                lo[q] = fminf(lo[q], __shfl_xor(lo[q], 16, 64)); hi[q] = fmaxf(hi[q], __shfl_xor(hi[q], 16, 64));
                lo[q] = fminf(lo[q], dpp_xor<8>(lo[q])); hi[q] = fmaxf(hi[q], dpp_xor<8>(hi[q]));
                lo[q] = fminf(lo[q], dpp_xor<4>(lo[q])); hi[q] = fmaxf(hi[q], dpp_xor<4>(hi[q]));
                lo[q] = fminf(lo[q], dpp_xor<2>(lo[q])); hi[q] = fmaxf(hi[q], dpp_xor<2>(hi[q]));
                lo[q] = fminf(lo[q], dpp_xor<1>(lo[q])); hi[q] = fmaxf(hi[q], dpp_xor<1>(hi[q]));
            }
            if ((lane & 31) == 0) {
                const int gi = jj >> 5;
#pragma unroll
                for (int q = 0; q < 4; ++q) { L.gbox[q * gplane + gi] = lo[q]; L.gbox[(4 + q) * gplane + gi] = hi[q]; }
            }
        }
        if (tid == 0) sh->cull_next = 0;
        __syncthreads();
#ifdef CVO_KTRACE_CULL
        if (tid == 0) sh->kabs[1] = CVO_NOW();
#endif
        if (predicted) {
            // (all of the cloud is in this one tile when the prediction is on: tile >= nm is required below)
            off_l = block_max(off_l, sh, tid, nthreads >> 6);
            if ((r_c + off_l) * 1.00002f > Rb) { again = true; break; }   // the current positions are not inside the lists' reach: once more, at the current positions
        }
        const int ngr = tnp >> 5;
        // Block pairs are handed out as the waves come for them: a pair's cost follows the hits of its 128 rows (near surfaces have several
        // times the neighbours of far ones) and a fixed deal of three pairs per wave left the workgroup waiting for the unluckiest wave.  A
        // row's list is made by one lane in column order whichever wave runs it: the lists do not depend on the order.
        // With fewer block pairs than half the waves (one pair on eight workgroups: three block pairs for eight waves) a unit of work is ONE 64-row block: the
        // lane's second row is padding and is not swept, twice the waves take part, each with half the arithmetic and only the groups near its own 64 rows.
        const int halves = (CVO_CULL_SPLIT == 2 || (CVO_CULL_SPLIT && nblk2 * 2 <= (nthreads >> 6))) ? 2 : 1;   // (CVO_CULL_SPLIT=2, experiment builds: always)
        auto units = [&](auto single_t) {
        for (;;) {
            int unit = 0;
            if (lane == 0) unit = atomicAdd(&sh->cull_next, 1);
            unit = uni(unit);
            if (unit >= nblk2 * halves) break;
            constexpr bool single = decltype(single_t)::value;      // true: the unit is 64-row block `unit`, the lane's row 0; its row 1 is padding
            float x[SWEEP_R][3]; int li[SWEEP_R]; int cnt[SWEEP_R];
            float nthr[SWEEP_R];                                    // - (the row's list radius)^2
            float Rrow = Rb * inv_1ma;                              // the widest radius of the wave's rows
            unsigned long long buf[SWEEP_R];                        // the row's last, not yet full word of four columns
            float blo[4] = {INF, INF, INF, INF}, bhi[4] = {-INF, -INF, -INF, -INF};
#pragma unroll
            for (int r = 0; r < SWEEP_R; ++r) {
                li[r] = single ? unit * 64 + lane : (unit * SWEEP_R + r) * 64 + lane;
                nthr[r] = -(Rrow * Rrow * 1.00001f);
                if (li[r] < nrows && (!single || r == 0)) {
                    const float4 lo4 = ld4(c.fixed + lo_off(global_row(c, li[r])));
                    x[r][0] = lo4.x; x[r][1] = lo4.y; x[r][2] = lo4.z;
                    if (alpha > 0.f) {
                        const float xn = sqrtf(__builtin_fmaf(lo4.z, lo4.z, __builtin_fmaf(lo4.y, lo4.y, lo4.x * lo4.x)));
                        const float Ri = (Rb + alpha * xn * 1.0001f) * inv_1ma;
                        nthr[r] = -(Ri * Ri * 1.00001f);
                        xmax_l = fmaxf(xmax_l, xn);
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) { blo[q] = fminf(blo[q], x[r][q]); bhi[q] = fmaxf(bhi[q], x[r][q]); }
                    if (lo4.z > 1.0e-3f) { const float t = lo4.y / lo4.z; blo[3] = fminf(blo[3], t); bhi[3] = fmaxf(bhi[3], t); }
                    else { blo[3] = -INF; bhi[3] = INF; }
                } else {
                    x[r][0] = x[r][1] = x[r][2] = FAR_ROW;
                }
                cnt[r] = (t0 == 0) ? 0 : rowlen[li[r]];
                buf[r] = 0ull;
                if (t0 != 0 && (cnt[r] & 3) && cnt[r] < c.capn) {   // a later tile of a large cloud: pick the row's partial word up again
                    const v2u w = c.jT4[(size_t)(cnt[r] >> 2) * c.rows_pad + li[r]];
                    buf[r] = ((unsigned long long)w.y << 32) | w.x;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { blo[q] = wave_min(blo[q]); bhi[q] = wave_max(bhi[q]); }
            if (alpha > 0.f) Rrow = sqrtf(wave_max(-fminf(nthr[0], nthr[1])));
            const float thr_box = Rrow * Rrow * 1.00001f * 1.001f;  // box gaps are compared with a margin: a skipped group holds no hit
            // points p (a row), q (a column) within R of each other: |y_p/z_p - y_q/z_q| <= R (1 + |y_q/z_q|) / z_p
            const float slope_reach = (blo[2] > 1.0e-3f) ? Rrow * 1.01f / blo[2] : INF;
            for (int gb = 0; gb < ngr; gb += 64) {
                bool near = false;
                if (gb + lane < ngr) {
                    float gap2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const float glo = L.gbox[q * gplane + gb + lane], ghi = L.gbox[(4 + q) * gplane + gb + lane];
                        const float d = fmaxf(0.f, fmaxf(glo - bhi[q], blo[q] - ghi));
                        gap2 = __builtin_fmaf(d, d, gap2);
                    }
                    const float tlo = L.gbox[3 * gplane + gb + lane], thi = L.gbox[7 * gplane + gb + lane];
                    const float tgap = fmaxf(0.f, fmaxf(tlo - bhi[3], blo[3] - thi));
                    const float tabs = fmaxf(fabsf(tlo), fabsf(thi));
                    // comparisons are false for NaN (inf - inf of an all-padding group): skipped
                    near = (gap2 <= thr_box) && (tgap <= slope_reach * (1.0f + tabs) + 1.0e-6f);
                }
                unsigned long long mask = __ballot(near);
                while (mask) {
                    const int gi = gb + __builtin_ctzll(mask);
                    mask &= mask - 1ull;
                    uint32_t w[SWEEP_R];
                    if (single) sweep_group<1>(colx, coly, colz, gi * 32, x, nthr, w); else sweep_group<SWEEP_R>(colx, coly, colz, gi * 32, x, nthr, w);
                    const uint32_t col0 = (uint32_t)(t0 + gi * 32);
                    // bit 31 = first column of the group: ascending columns.  Up to four hits per row leave the words per trip (the loop runs as
                    // long as the fullest row of the wave needs: a trip per hit was more than half of the cull) and go into the rows' 8-byte
                    // words of four columns; a full word is stored (a 2-byte store per hit was the cull's first bottleneck).  The lane's rows take
                    // their trips together: the bit extraction is a chain of dependent operations, two rows = two chains side by side, and the
                    // loop runs max(trips) instead of their sum.
                    static_assert(SWEEP_R == 2, "the hit words of a lane's two rows are unpacked side by side");
                    uint32_t wq[SWEEP_R] = {w[0], w[1]};
                    while (wq[0] | wq[1]) {
                        int nn[SWEEP_R]; unsigned long long four[SWEEP_R];
#pragma unroll
                        for (int r = 0; r < SWEEP_R; ++r) {
                            const uint32_t ww = wq[r];
                            nn[r] = min(__popc(ww), 4);
                            const int k0 = __clz(ww) & 31; const uint32_t w1 = ww & ~(0x80000000u >> k0);
                            const int k1 = __clz(w1) & 31; const uint32_t w2 = w1 & ~(0x80000000u >> k1);
                            const int k2 = __clz(w2) & 31; const uint32_t w3 = w2 & ~(0x80000000u >> k2);
                            const int k3 = __clz(w3) & 31;
                            wq[r] = w3 & ~(0x80000000u >> k3);
                            const uint32_t lo = (col0 + (uint32_t)k0) | (nn[r] > 1 ? (col0 + (uint32_t)k1) << 16 : 0u);
                            const uint32_t hi = (nn[r] > 2 ? col0 + (uint32_t)k2 : 0u) | (nn[r] > 3 ? (col0 + (uint32_t)k3) << 16 : 0u);
                            four[r] = ((unsigned long long)hi << 32) | lo;
                        }
#pragma unroll
                        for (int r = 0; r < SWEEP_R; ++r) {
                            const int n = nn[r];
                            if (n > 0) {
                                const int fill = cnt[r] & 3;
                                if (cnt[r] < c.capn) {              // capn is a multiple of 4: a word that starts below it ends at or below it
                                    buf[r] |= four[r] << (16 * fill);
                                    if (fill + n >= 4) {
                                        v2u wv; wv.x = (unsigned)buf[r]; wv.y = (unsigned)(buf[r] >> 32);
                                        st_jt(&c.jT4[(size_t)(cnt[r] >> 2) * c.rows_pad + li[r]], wv);
                                        buf[r] = fill ? (four[r] >> (16 * (4 - fill))) : 0ull;
                                    }
                                }
                                cnt[r] += n;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < SWEEP_R; ++r) {
                if (single && r != 0) continue;                     // (padding)
                if ((cnt[r] & 3) && cnt[r] < c.capn) {              // the partial word (capn is a multiple of 4: a full list never leaves one)
                    v2u w; w.x = (unsigned)buf[r]; w.y = (unsigned)(buf[r] >> 32);
                    st_jt(&c.jT4[(size_t)(cnt[r] >> 2) * c.rows_pad + li[r]], w);
                }
                rowlen[li[r]] = (uint16_t)cnt[r];                   // rowlen has room for the padding rows of the last block pair
            }
        }
        };
        if (halves == 2) units(std::true_type{}); else units(std::false_type{});   // (two instances: the common path keeps its code and registers)
    }
    if (again) predicted = false;
    } while (again);
#ifdef CVO_KTRACE_CULL
    if (tid == 0) sh->kabs[2] = CVO_NOW();
    __syncthreads();
    if (tid == 0) sh->kabs[3] = CVO_NOW();
#endif
    // The margins that later list refinements may take (phase_refine, the filtering walk) depend on the farthest row.  A pair's members must agree on them -- a
    // member with near rows only would give itself a wider margin, find its lists stale at another iteration than the others and rebuild alone, with every other
    // member waiting for it in the exchange (seen: 67 us of one pair's 1.6 ms) -- so the farthest row is the PAIR's, whichever workgroup owns it.
    if (alpha > 0.f && c.G > 1 && CVO_PAIR_XMAX) {
        for (int i = tid; i < c.nf; i += nthreads) {
            const float4 lo4 = ld4(c.fixed + lo_off(i));
            xmax_l = fmaxf(xmax_l, sqrtf(__builtin_fmaf(lo4.z, lo4.z, __builtin_fmaf(lo4.y, lo4.y, lo4.x * lo4.x))));
        }
    }
    if (alpha > 0.f) xmax_l = block_max(xmax_l, sh, tid, nthreads >> 6);
    if (tid == 0) {
        sh->Rb = Rb; sh->alpha_build = alpha; sh->xmax = xmax_l; sh->ell_build = sh->ell; sh->list_valid = 1; sh->rebuilds += 1;
        for (int i = 0; i < 12; ++i) sh->Mb[i] = predicted ? Mp[i] : sh->M[i];
        if (predicted) sh->predicted += 1;
    }
    __syncthreads();
}

// ---- X: rows sorted by list length.  The candidate phase walks one row per lane, so a wave pays for the longest list of
// its 64 rows: rows are ordered by length class (descending, stable in row order: a counting sort whose every count is
// deterministic) into "slots", 64 consecutive slots form a block of near-equal lists, and the blocks (longest first) are
// dealt to the waves in serpentine order (0..7, 7..0, 0..7, ...), which keeps the waves' loads within one block of each other.
// Only the order in which rows are visited changes: each row's own sum still runs in column order.
__device__ __forceinline__ int len_class(int len) { return min(NCLS - 1, (len + PF - 1) / PF); }
// i-th block of a wave in the serpentine deal
__device__ __forceinline__ int wave_block(int i, int wave, int nwaves) { return i * nwaves + ((i & 1) ? nwaves - 1 - wave : wave); }

// rows -> slots: from L.rowlen[local row] to L.lenS[slot], L.row_of[slot].  Every thread; ends with a barrier.
__device__ __forceinline__ void sort_slots(const Lds& L, int nrows, int tid, int nthreads) {
    const int lane = tid & 63, nwaves = nthreads >> 6, wave = tid >> 6;
    const int nblk = (nrows + 63) >> 6;
    int* hist = L.hist; int* base = L.base;
    for (int i = tid; i < MAX_WAVES * NCLS; i += nthreads) hist[i] = 0;
    for (int sl = nrows + tid; sl < nblk * 64; sl += nthreads) { L.lenS[sl] = 0; L.row_of[sl] = 0; }   // padding slots: empty lists of row 0
    __syncthreads();
    // wave w sorts the contiguous rows [r0, r1)
    const int per = (((nrows + nwaves - 1) / nwaves) + 63) & ~63;
    const int r0 = min(nrows, wave * per), r1 = min(nrows, r0 + per);
    for (int li = r0 + lane; li < r1; li += 64) atomicAdd(&hist[wave * NCLS + len_class(L.rowlen[li])], 1);
    __syncthreads();
    if (tid < 64) {
        // lane l owns classes 2l and 2l+1: rows per class over all waves, then an exclusive SUFFIX sum (longer classes first)
        int t0 = 0, t1 = 0;
        for (int w = 0; w < nwaves; ++w) { t0 += hist[w * NCLS + 2 * lane]; t1 += hist[w * NCLS + 2 * lane + 1]; }
        int inc = t0 + t1;                                          // inclusive suffix over lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_down(inc, off, 64); if (lane + off < 64) inc += t; }
        const int after = inc - (t0 + t1);                          // rows in classes of higher lanes
        int run0 = after + t1, run1 = after;
        for (int w = 0; w < nwaves; ++w) {
            const int h0 = hist[w * NCLS + 2 * lane], h1 = hist[w * NCLS + 2 * lane + 1];
            base[w * NCLS + 2 * lane] = run0; base[w * NCLS + 2 * lane + 1] = run1;
            run0 += h0; run1 += h1;
        }
    }
    __syncthreads();
    for (int li0 = r0; li0 < r1; li0 += 64) {
        const int li = li0 + lane;
        const bool valid = li < r1;
        const int len = valid ? (int)L.rowlen[li] : 0;
        const int cls = valid ? len_class(len) : -1;
        // rank among the rows of the same class in this step and the class's size, by ballots alone; LDS is touched once after
        int rank = 0, csize = 0; bool leader = false;
        unsigned long long todo = __ballot(valid);
        while (todo) {                                              // one trip per distinct class among the 64 rows
            const int src = __builtin_ctzll(todo);
            const int c0 = __builtin_amdgcn_readlane(cls, src);
            const unsigned long long m = __ballot(cls == c0);
            if (cls == c0) { rank = __popcll(m & ((1ull << lane) - 1ull)); csize = __popcll(m); leader = lane == src; }
            todo &= ~m;
        }
        if (valid) {
            const int slot = base[wave * NCLS + cls] + rank;
            L.lenS[slot] = (uint16_t)len; L.row_of[slot] = (uint16_t)li;
        }
        __builtin_amdgcn_wave_barrier();
        if (leader) base[wave * NCLS + cls] += csize;               // one lane per class: no two lanes touch the same counter
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
}

// what follows a new slot order: the waves' blocks and totals, the segments of the nonzero records, the fixed points by slot.
// Every thread; ends with a barrier.
__device__ __forceinline__ void finish_slots(const Ctx& c, const Lds& L, Shared* sh, int tile, int tid, int nthreads, bool fresh_lists) {
    const int lane = tid & 63, nwaves = nthreads >> 6, wave = tid >> 6;
    const int nblk = (c.nrows + 63) >> 6;
    // every wave sums up the blocks the serpentine deal gives it
    int my_lmax = 0, my_tot = 0, my_nb = 0;
    for (int i = 0;; ++i) {
        const int b = wave_block(i, wave, nwaves);
        if (b >= nblk) break;
        const int len = L.lenS[b * 64 + lane];
        int lmaxb = len, ltot = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lmaxb = max(lmaxb, __shfl_xor(lmaxb, off, 64)); ltot += __shfl_xor(ltot, off, 64); }
        if (lane == 0) sh->blk_lmax[b] = (unsigned short)lmaxb;
        my_lmax = max(my_lmax, lmaxb); my_tot += ltot;
        if (lmaxb > 0) ++my_nb;                                      // slots are in descending length order: an empty block ends the wave's walk (at ell = 0.03 a sixth to a third of the rows have no neighbour left)
    }
    if (lane == 0) { sh->wsum[wave] = my_lmax; sh->wtot[wave] = my_tot; sh->wnb[wave] = my_nb; }
    __syncthreads();
    if (tid == 0) {
        int lmax_all = 0, run_w = 0;
        for (int w = 0; w < nwaves; ++w) { lmax_all = max(lmax_all, sh->wsum[w]); sh->wbase[w] = run_w; run_w += sh->wtot[w]; }
        sh->total = run_w; sh->lmax = lmax_all;
        if (fresh_lists) {
            // candidates beyond what the survivor planes hold, or a row longer than the lists: dense per-row fallback until the
            // next rebuild
            const int dense = ((run_w > c.flat_cap) || (lmax_all > c.capn)) ? 1 : 0;
            sh->dense_mode = dense;
            sh->dense_fallbacks += dense;
        }
    }
    // the cull tile is idle until the next cull: it keeps the fixed points in slot order for the candidate and line-search phases
    const int xl = (nblk * 64 <= tile) ? 1 : 0;
    if (xl) {
        for (int sl = tid; sl < nblk * 64; sl += nthreads) {
            const float4 lo = ld4(c.fixed + lo_off(global_row(c, (int)L.row_of[sl])));
            L.lx[sl] = lo.x; L.ly[sl] = lo.y; L.lz[sl] = lo.z;
        }
    }
    if (tid == 0) sh->x_lds = xl;
    __syncthreads();
}

static __device__ __noinline__ void phase_sort(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), tile = tgeo & 0x1FFF, y_lds = uni(y_lds_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh;
    const Ctx c = make_ctx(Dp, g, G);
    sort_slots(L, c.nrows, threadIdx.x, blockDim.x);
    finish_slots(c, L, sh, tile, threadIdx.x, blockDim.x, true);
}

// ---- C: exact kernel values + compute_flow row sums (cvo.cpp:202-231), reduced over the workgroup and the pair's workgroups.
// One lane per ROW (slot), walking the row's candidate list; the lanes of a wave read entry n of 64 consecutive slots as
// one coalesced run.  The point gather y_j comes from the LDS-resident cloud, x_i and the row sums live in registers, so
// a candidate costs no scattered global access and the f32 sums add up in column order (cvo.cpp:213-223) without a
// second pass.  Survivors {x_i,a},{y_j} are compacted per wave (ballot + prefix popcount) for the line-search phase.
struct RowSums { float sw[3], sv[3]; };

__device__ __forceinline__ void fold_entry(const float* xi, const float4 y4, float a, unsigned tag /* slot << 16 | column */, RowSums& rs, gv2u* sp /* the wave's segment of nonzero records: wave-uniform */,
                                           int& wcount, int lane) {
    {   // a == 0 for a non-member: it adds exact zeros, the sums keep their bits
        const float yv[3] = {y4.x, y4.y, y4.z};
        float cr[3]; cross3(xi, yv, cr);                            // cvo.cpp:216
        rs.sw[0] += a * cr[0]; rs.sw[1] += a * cr[1]; rs.sw[2] += a * cr[2];
        rs.sv[0] += a * (yv[0] - xi[0]); rs.sv[1] += a * (yv[1] - xi[1]); rs.sv[2] += a * (yv[2] - xi[2]);   // cvo.cpp:217
    }
    const unsigned long long mask = __ballot(a > 0.f);
    if (a > 0.f) {
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        v2u rec; rec.x = __float_as_uint(a); rec.y = tag;             // 8 bytes per nonzero: the line search finds x_i, y_j in LDS
        st_rec(at_off(sp, (unsigned)wcount + below), rec);            // scalar base + 32-bit lane offset
    }
    wcount += __popcll(mask);
}

// The wave's segment of nonzero records as a buffer resource: the record stores of the walks go through it with the lane's byte offset, and a lane
// that has no record to write gives an offset beyond the segment -- the range check of the buffer instruction drops it.  A store under `if (a > 0)` is a
// divergent branch (s_and_saveexec, s_cbranch_execz around the store), and behind four of those the compiler no longer knows how many stores are in flight
// when the walk claims its prefetched list entries at the end of the step: it waited for vmcnt(0) there -- every step of the walk ended by draining its own
// record stores, a round trip to L2 with nothing to overlap it.  Unconditional stores are counted exactly: the wait becomes vmcnt(4) and the stores of a step
// complete under the arithmetic of the next.
constexpr unsigned REC_DROP = 0x7FFFFFF0u;   // a byte offset beyond any segment (segments are far below 2 GB)
// a whole work buffer of the pair (lists, raw lists) as a buffer resource: offsets below REC_DROP are in range
__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_region(const void* base /* wave-uniform */) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(REC_DROP - 16u), 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t record_segment(gv2u* sp /* wave-uniform */, unsigned records) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)sp, 0, (int)(records * 8u), 0x00020000);
}
// the same with e = x_i - y_j already at hand (the flat evaluation computed it for d2): y_j - x_i = -e exactly, so
// sv - a*e has the bits of sv + a*(y_j - x_i)
__device__ __forceinline__ void fold_entry_e(const float* xi, const float4 y4, const float (&e)[3], float a, unsigned tag, RowSums& rs, gv2u* sp, __amdgpu_buffer_rsrc_t seg,
                                             int& wcount, int lane) {
    {
        const float yv[3] = {y4.x, y4.y, y4.z};
        float cr[3]; cross3(xi, yv, cr);                            // cvo.cpp:216
        rs.sw[0] += a * cr[0]; rs.sw[1] += a * cr[1]; rs.sw[2] += a * cr[2];
        rs.sv[0] -= a * e[0]; rs.sv[1] -= a * e[1]; rs.sv[2] -= a * e[2];     // cvo.cpp:217
    }
    const unsigned long long mask = __ballot(a > 0.f);
#ifndef CVO_BRANCHY_REC
    {
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        v2u rec; rec.x = __float_as_uint(a); rec.y = tag;
        const unsigned off = (a > 0.f) ? ((unsigned)wcount + below) * 8u : REC_DROP;
        __builtin_amdgcn_raw_buffer_store_b64(rec, seg, (int)off, 0, 0);
    }
#else
    if (a > 0.f) {
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        v2u rec; rec.x = __float_as_uint(a); rec.y = tag;
        st_rec(at_off(sp, (unsigned)wcount + below), rec);
    }
#endif
    wcount += __popcll(mask);
}

// every iteration but the first after a rebuild: entries {ck, j} stream in, PF per lane per step, the next step's in flight
// REFINE: this is the last iteration at the present ell (cvo.cpp:810-812 is a schedule by iteration count), and the walk also makes the lists of the NEXT ell: it has
// every entry and its squared distance at hand anyway, so the in-place filter of phase_refine (same rule: kept within the next radius of the current positions,
// row i with (Rn + alpha |x_i|) / (1 - alpha)) costs it a compare and a store per kept entry instead of a pass of its own over the lists.  Rn, alpha: phase_candidates.
template <int YM, bool FLAT, bool NT = true, bool REFINE = false>
__device__ __forceinline__ void cand_steady(const Ctx& c, const Lds& L, Shared* sh, const Gates& gates, int lane, int wave, int nwaves, float inv_c, float inv_d,
                                            double (&acc8)[8], float Rn = 0.f, float alpha_n = 0.f, bool do_shift = false, const float* shift_rt = nullptr) {
    gv2u* sp = uni_ptr(c.surv + c.fbase + sh->wbase[wave]);
    const __amdgpu_buffer_rsrc_t seg = record_segment(sp, (unsigned)uni(sh->wtot[wave]));   // (a wave has no more nonzeros than listed candidates)
    const bool x_lds = sh->x_lds != 0;
    int wcount = 0;
    const int nb = sh->wnb[wave];
    int kept_w = 0, nb_left = 0;
#ifdef CVO_EXP7
    const Exp7 e7 = make_exp7(gates);
#endif
    const float inv_1ma_n = 1.0f / (1.0f - alpha_n);
    const unsigned rp = (unsigned)c.rows_pad, estep = (PF / 2) * rp;      // in 16-byte words
    // the first entries of a block are fetched while the block before it is walked: in the late iterations a row holds only a
    // handful of entries, and a block would otherwise start with an exposed memory round trip
    static_assert(PF == 4, "a step is two 16-byte words of two entries");
    v4u ehead[PF / 2];
    if (nb > 0) {
        const gv4u* eb0 = uni_ptr((const gv4u*)c.ent + wave_block(0, wave, nwaves) * 64);
#pragma unroll
        for (int u = 0; u < PF / 2; ++u) ehead[u] = ld_ent2<(NT && ENT_NT(YM))>(at_off(eb0, (unsigned)lane + (unsigned)u * rp));
    }
    for (int bi = 0; bi < nb; ++bi) {
        prio_by_remaining(uni(nb) - 1 - bi);
        const int blk = wave_block(bi, wave, nwaves);
        const int slot = blk * 64 + lane;
        const int len = L.lenS[slot];
        const int lw = uni((int)sh->blk_lmax[blk]);                  // longest list of the block (phase_sort / refine_lists)
        float xi[3]; load_x(c, L, x_lds, slot, xi);
        RowSums rs = {{0, 0, 0}, {0, 0, 0}};
        const gv4u* eb = uni_ptr((const gv4u*)c.ent + (slot - lane)); // scalar base of the block's entries + 32-bit lane offsets
        unsigned eo = (unsigned)lane;
        const unsigned stag = (unsigned)slot << 16;
        int cnt = 0; float thr_n = 0.f; gv2u* wp = nullptr; unsigned woff = 0u;
        float xs[3] = {xi[0], xi[1], xi[2]};                        // the row as the next lists see it: moved by the inverse of the extrapolated motion (refine_lists, `shift`)
        if (REFINE) {
            const float xn = sqrtf(__builtin_fmaf(xi[2], xi[2], __builtin_fmaf(xi[1], xi[1], xi[0] * xi[0])));
            float Ri = (Rn + alpha_n * xn * 1.0001f) * inv_1ma_n;
            if (do_shift) {
                xs[0] = sum3f(shift_rt[0] * xi[0], shift_rt[1] * xi[1], shift_rt[2] * xi[2]) + shift_rt[9];
                xs[1] = sum3f(shift_rt[3] * xi[0], shift_rt[4] * xi[1], shift_rt[5] * xi[2]) + shift_rt[10];
                xs[2] = sum3f(shift_rt[6] * xi[0], shift_rt[7] * xi[1], shift_rt[8] * xi[2]) + shift_rt[11];
                Ri += 4.0e-6f * (1.0f + xn);
            }
            thr_n = Ri * Ri * 1.0001f;                              // (the margin covers the rounding of the un-fused sums against the cull's fused one many times over)
            wp = uni_ptr(c.ent + 2 * (slot - lane));                // the kept entries go to the front of the row (never ahead of the reads): scalar base of the block + 32-bit offsets
            woff = 2u * (unsigned)lane;                             // entry n of the lane's slot at 2 ((n >> 1) rows_pad + slot) + (n & 1)
        }
        const __amdgpu_buffer_rsrc_t wseg = whole_region((const void*)(REFINE ? wp : sp));
        v4u eq4[PF / 2];
#pragma unroll
        for (int u = 0; u < PF / 2; ++u) eq4[u] = ehead[u];
        if (bi + 1 < nb) {
            const gv4u* eb1 = uni_ptr((const gv4u*)c.ent + wave_block(bi + 1, wave, nwaves) * 64);
#pragma unroll
            for (int u = 0; u < PF / 2; ++u) ehead[u] = ld_ent2<(NT && ENT_NT(YM))>(at_off(eb1, (unsigned)lane + (unsigned)u * rp));
        }
        for (int n0 = 0; n0 < lw; n0 += PF) {
            v4u en4[PF / 2];
#ifndef CVO_ALWAYS_PREFETCH
            // the next step's entries -- when there is a next step.  (The last step of a block used to fetch entries nobody reads and to wait for them at its end:
            // in the light iterations, where a block is one step, every step ended with a round trip to memory.)
            if (n0 + PF < lw) {
                eo += estep;
#pragma unroll
                for (int u = 0; u < PF / 2; ++u) en4[u] = ld_ent2<(NT && ENT_NT(YM))>(at_off(eb, eo + (unsigned)u * rp));
            } else {
#pragma unroll
                for (int u = 0; u < PF / 2; ++u) en4[u] = eq4[u];
            }
#else
            if (n0 + 2 * PF <= c.capn) eo += estep;                 // the prefetch stays inside the lists (the last step re-reads its own entries)
#pragma unroll
            for (int u = 0; u < PF / 2; ++u) en4[u] = ld_ent2<(NT && ENT_NT(YM))>(at_off(eb, eo + (unsigned)u * rp));
#endif
            v2u eq[PF];
#pragma unroll
            for (int u = 0; u < PF / 2; ++u) { eq[2 * u].x = eq4[u].x; eq[2 * u].y = eq4[u].y; eq[2 * u + 1].x = eq4[u].z; eq[2 * u + 1].y = eq4[u].w; }
            float av[PF], ckv[PF]; float4 yv4[PF]; bool actv[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                actv[u] = n0 + u < len;
                const int j = actv[u] ? (int)eq[u].y : 0;           // slots past the row's end hold stale entries
                yv4[u] = load_y<YM>(c, L, j);
                ckv[u] = __uint_as_float(eq[u].x);
            }
            if (FLAT) {
                float ev[PF][3], d2v[PF];
#ifdef CVO_EXP7
                se_kernel_values_flat7<PF>(xi, yv4, ckv, actv, gates, e7, av, ev, REFINE ? d2v : nullptr);   // PF exp chains side by side
#else
                se_kernel_values_flat<PF>(xi, yv4, ckv, actv, gates, av, ev, REFINE ? d2v : nullptr);   // PF exp chains side by side
#endif
#pragma unroll
                for (int u = 0; u < PF; ++u) fold_entry_e(xi, yv4[u], ev[u], av[u], stag | (eq[u].y & 0xFFFFu), rs, sp, seg, wcount, lane);
                if (REFINE) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        float dn = d2v[u];
                        if (do_shift) { const float q0 = xs[0] - yv4[u].x, q1 = xs[1] - yv4[u].y, q2 = xs[2] - yv4[u].z; dn = __builtin_fmaf(q2, q2, __builtin_fmaf(q1, q1, q0 * q0)); }
#ifndef CVO_BRANCHY_REC
                        {   // (unconditional store, dropped by the range check for an entry that is not kept: see record_segment)
                            const bool keep = actv[u] && dn < thr_n;
                            __builtin_amdgcn_raw_buffer_store_b64(eq[u], wseg, (int)(keep ? woff * 8u : REC_DROP), 0, 0);
                            woff += keep ? ((cnt & 1) ? 2u * rp - 1u : 1u) : 0u; cnt += keep ? 1 : 0;
                        }
#else
                        if (actv[u] && dn < thr_n) { st_rf(at_off(wp, woff), eq[u]); woff += (cnt & 1) ? 2u * rp - 1u : 1u; ++cnt; }
#endif
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < PF; ++u) av[u] = actv[u] ? se_kernel_value_ck(xi, yv4[u], ckv[u], gates) : 0.f;
#pragma unroll
                for (int u = 0; u < PF; ++u) fold_entry(xi, yv4[u], av[u], stag | (eq[u].y & 0xFFFFu), rs, sp, wcount, lane);
            }
#pragma unroll
            for (int u = 0; u < PF / 2; ++u) eq4[u] = en4[u];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { acc8[q] += (double)(inv_c * rs.sw[q]); acc8[3 + q] += (double)(inv_d * rs.sv[q]); }   // cvo.cpp:222-223
        if (REFINE) {                                               // the block's lists as they are walked from the next iteration on (cf. refine_lists)
            L.lenS[slot] = (uint16_t)cnt;
            kept_w += cnt;
            int cmax = cnt;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off, 64));
            if (lane == 0) sh->blk_lmax[blk] = (unsigned short)cmax;
            if (cmax > 0) nb_left = bi + 1;
        }
    }
    if (lane == 0) { sh->wcnt[wave] = wcount; acc8[6] = (double)wcount; }
    if (REFINE) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) kept_w += __shfl_xor(kept_w, off, 64);
        if (lane == 0) { sh->wnb[wave] = nb_left; sh->wsum[wave] = kept_w; }
    }
}

// the first iteration on new lists: columns come from the cull's raw lists (by row), the colour gate and factor
// (cvo.cpp:169-173) are evaluated once and kept with the column in the slot-ordered entries.  Three stages in flight per
// lane: columns of step s+2 (gathered from the raw lists), second feature plane of the columns of step s+1, arithmetic
// of step s.
// GL: the moving cloud's second feature plane has been staged in LDS (the line-search table's room, dead until the line search of this iteration makes its
// table): the four 16-byte feature gathers of a step are LDS reads instead of vector-memory gathers from L2
template <int YM, bool FLAT, bool GL = false>
__device__ __forceinline__ void cand_fresh(const Ctx& c, const Lds& L, Shared* sh, const Gates& gates, int lane, int wave, int nwaves, float inv_c, float inv_d,
                                           double (&acc8)[8]) {
    gv2u* sp = uni_ptr(c.surv + c.fbase + sh->wbase[wave]);
    const __amdgpu_buffer_rsrc_t seg = record_segment(sp, (unsigned)uni(sh->wtot[wave]));
    const __amdgpu_buffer_rsrc_t jseg = whole_region((const void*)uni_ptr(c.jT4)), eseg = whole_region((const void*)uni_ptr(c.ent));
    int wcount = 0;
    const int nb = sh->wnb[wave];
#ifdef CVO_EXP7
    const Exp7 e7 = make_exp7(gates);
#endif
    for (int bi = 0; bi < nb; ++bi) {
        prio_by_remaining(uni(nb) - 1 - bi);
        const int blk = wave_block(bi, wave, nwaves);
        const int slot = blk * 64 + lane;
        const int len = L.lenS[slot];
        const int li = L.row_of[slot];
        const int lw = uni((int)sh->blk_lmax[blk]);
        const int gi = global_row(c, li);
        const float4 lo = ld4(c.fixed + lo_off(gi)), hi = ld4(c.fixed + hi_off(c.nf, gi));
        const float xi[3] = {lo.x, lo.y, lo.z};
        const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
        RowSums rs = {{0, 0, 0}, {0, 0, 0}};
        static_assert(PF == 4, "the cull packs four columns per word");
#ifdef CVO_BRANCHY_REC
        const gv2u* jp = c.jT4 + li;                                // entries 4q .. 4q+3 of this row: jp[q * rows_pad]
        gv2u* ep = c.ent + 2 * slot;
#endif
        const unsigned stag = (unsigned)slot << 16;
        auto cols = [&](int n0, int (&jo)[PF]) {
#pragma unroll
            for (int u = 0; u < PF; ++u) jo[u] = 0;
#ifndef CVO_BRANCHY_REC
            {   // (unconditional load: a lane whose row has ended reads beyond the range and gets zeros -- no branch, so the loads in flight stay countable)
                const v2u w = __builtin_amdgcn_raw_buffer_load_b64(jseg, (int)(n0 < len ? ((unsigned)(n0 >> 2) * (unsigned)c.rows_pad + (unsigned)li) * 8u : REC_DROP), 0, 0);
                const int q[PF] = {(int)(w.x & 0xFFFFu), (int)(w.x >> 16), (int)(w.y & 0xFFFFu), (int)(w.y >> 16)};
#pragma unroll
                for (int u = 0; u < PF; ++u) jo[u] = (n0 + u < len) ? q[u] : 0;
            }
#else
            if (n0 < len) {
                const v2u w = ld_jt(&jp[(size_t)(n0 >> 2) * c.rows_pad]);
                const int q[PF] = {(int)(w.x & 0xFFFFu), (int)(w.x >> 16), (int)(w.y & 0xFFFFu), (int)(w.y >> 16)};
#pragma unroll
                for (int u = 0; u < PF; ++u) jo[u] = (n0 + u < len) ? q[u] : 0;
            }
#endif
        };
        auto feats = [&](const int (&ji)[PF], float4 (&go)[PF]) {
#pragma unroll
            for (int u = 0; u < PF; ++u) go[u] = GL ? L.tab[ji[u]] : ld4(c.moving + hi_off(c.nm, ji[u]));
        };
        int j0[PF], j1[PF], j2[PF]; float4 g0[PF], g1[PF];
        cols(0, j0); cols(PF, j1);
        feats(j0, g0);
        for (int n0 = 0; n0 < lw; n0 += PF) {
            cols(n0 + 2 * PF, j2);
            feats(j1, g1);
            float av[PF], ckv[PF], d2c[PF]; float4 yv4[PF]; bool actv[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                actv[u] = n0 + u < len;
                const int j = j0[u];
                yv4[u] = load_y<YM>(c, L, j);
                const float f0 = (YM == 2) ? ld4(c.moving + lo_off(j)).w : yv4[u].w;   // the first channel rides with y, except in the 12-byte LDS layout
                const float fb[5] = {f0, g0[u].x, g0[u].y, g0[u].z, g0[u].w};
                d2c[u] = feat_d2(fi, fb);
            }
            colour_factors<PF>(d2c, gates, ckv);
#pragma unroll
#ifndef CVO_BRANCHY_REC
            for (int u = 0; u < PF; u += 2) {                         // two entries per 16-byte store (the second may lie beyond the row's end: stale there anyway)
                v4u e; e.x = __float_as_uint(ckv[u]); e.y = (unsigned)j0[u]; e.z = __float_as_uint(ckv[u + 1]); e.w = (unsigned)j0[u + 1];
                __builtin_amdgcn_raw_buffer_store_b128(e, eseg, (int)(actv[u] ? (2u * (unsigned)slot + (unsigned)ent_ix(n0 + u, (size_t)c.rows_pad)) * 8u : REC_DROP), 0, 2 /* nt */);
            }
#else
            for (int u = 0; u < PF; u += 2)                           // two entries per 16-byte store (the second may lie beyond the row's end: stale there anyway)
                if (actv[u]) {
                    v4u e; e.x = __float_as_uint(ckv[u]); e.y = (unsigned)j0[u]; e.z = __float_as_uint(ckv[u + 1]); e.w = (unsigned)j0[u + 1];
#ifndef CVO_PLAIN_ENT_ST
                    __builtin_nontemporal_store(e, reinterpret_cast<gv4u*>(&ep[ent_ix(n0 + u, (size_t)c.rows_pad)]));
#else
                    *reinterpret_cast<gv4u*>(&ep[ent_ix(n0 + u, (size_t)c.rows_pad)]) = e;
#endif
                }
#endif
            if (FLAT) {
                float ev[PF][3];
#ifdef CVO_EXP7
                se_kernel_values_flat7<PF>(xi, yv4, ckv, actv, gates, e7, av, ev);
#else
                se_kernel_values_flat<PF>(xi, yv4, ckv, actv, gates, av, ev);
#endif
#pragma unroll
                for (int u = 0; u < PF; ++u) fold_entry_e(xi, yv4[u], ev[u], av[u], stag | (unsigned)j0[u], rs, sp, seg, wcount, lane);
            } else {
#pragma unroll
                for (int u = 0; u < PF; ++u) av[u] = actv[u] ? se_kernel_value_ck(xi, yv4[u], ckv[u], gates) : 0.f;
#pragma unroll
                for (int u = 0; u < PF; ++u) fold_entry(xi, yv4[u], av[u], stag | (unsigned)j0[u], rs, sp, wcount, lane);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) { j0[u] = j1[u]; j1[u] = j2[u]; g0[u] = g1[u]; }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { acc8[q] += (double)(inv_c * rs.sw[q]); acc8[3 + q] += (double)(inv_d * rs.sv[q]); }   // cvo.cpp:222-223
    }
    if (lane == 0) { sh->wcnt[wave] = wcount; acc8[6] = (double)wcount; }
}

// ---- R: ell has dropped: the lists shrink to the new radius in place.  The test is the cull's (same fused arithmetic on the
// current positions), so the result is the list a dense cull would build now, with the colour factors it already carries.
// shift: the lists are filtered around positions dR^T (y_j - dT) a stretch ahead on the path instead of the current y_j (phase_cull, "where the cloud is heading");
// |x_i - dR^T (y_j - dT)| = |(dR x_i + dT) - y_j|, so the row moves once instead of every column (sh_rt = {dR row-major, dT}); the radius takes a slack for the
// rounding of that detour.
template <int YM>
__device__ __forceinline__ int refine_lists(const Ctx& c, const Lds& L, Shared* sh, float Rb, float alpha, bool shift, const float (&sh_rt)[12], int lane, int wave, int nwaves) {
    const float inv_1ma = 1.0f / (1.0f - alpha);
    int kept = 0, nb_left = 0;
    const int nb = sh->wnb[wave];
    for (int bi = 0; bi < nb; ++bi) {
        const int blk = wave_block(bi, wave, nwaves);
        const int slot = blk * 64 + lane;
        const int len = L.lenS[slot];
        const int lw = uni((int)sh->blk_lmax[blk]);
        float xi[3]; load_x(c, L, sh->x_lds != 0, slot, xi);
        // the row's new list radius (Rb + alpha |x_i|) / (1 - alpha): the cull's for this ell
        const float xn = sqrtf(__builtin_fmaf(xi[2], xi[2], __builtin_fmaf(xi[1], xi[1], xi[0] * xi[0])));
        float Ri = (Rb + alpha * xn * 1.0001f) * inv_1ma;
        if (shift) {
            const float s0 = sum3f(sh_rt[0] * xi[0], sh_rt[1] * xi[1], sh_rt[2] * xi[2]) + sh_rt[9], s1 = sum3f(sh_rt[3] * xi[0], sh_rt[4] * xi[1], sh_rt[5] * xi[2]) + sh_rt[10],
                        s2 = sum3f(sh_rt[6] * xi[0], sh_rt[7] * xi[1], sh_rt[8] * xi[2]) + sh_rt[11];
            xi[0] = s0; xi[1] = s1; xi[2] = s2;
            Ri += 4.0e-6f * (1.0f + xn);
        }
        const float nthr = -(Ri * Ri * 1.00001f);
        gv2u* wp = c.ent + 2 * slot;                                // the kept entries go to the front of the row: never ahead of the reads
        const gv2u* ep = wp;
        const size_t rp = (size_t)c.rows_pad;
        int cnt = 0, nr = 0;
        v2u eq[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) eq[u] = ld_rf(&ep[ent_ix(u, rp)]);
        for (int n0 = 0; n0 < lw; n0 += PF) {
            if (n0 + 2 * PF <= c.capn) nr += PF;                    // next step's entries: loaded before this step stores anything
            v2u en[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) en[u] = ld_rf(&ep[ent_ix(nr + u, rp)]);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const bool act = n0 + u < len;
                const int j = act ? (int)eq[u].y : 0;
                const float4 y = load_y<YM>(c, L, j);
                const float dx = xi[0] - y.x, dy = xi[1] - y.y, dz = xi[2] - y.z;
                const float t = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, nthr)));
                if (act && t < 0.f) { st_rf(&wp[ent_ix(cnt, rp)], eq[u]); ++cnt; }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) eq[u] = en[u];
        }
        L.lenS[slot] = (uint16_t)cnt;
        kept += cnt;
        int cmax = cnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off, 64));
        if (lane == 0) sh->blk_lmax[blk] = (unsigned short)cmax;   // read again by this wave only (same block, later phases)
        if (cmax > 0) nb_left = bi + 1;
    }
    if (lane == 0) sh->wnb[wave] = nb_left;                          // blocks the filter emptied at the end of the wave's list are not walked again
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    return kept;
}

__device__ __forceinline__ void lists_filtered_impl(const Ctx& c, Shared* sh, int nwaves, int nblk, float Rb, float alpha, float ell_of_lists, int k);
__device__ __forceinline__ void resort_lists_impl(const Ctx& c, const Lds& L, Shared* sh, int tile, int tid, int nthreads);
__device__ __forceinline__ void lists_filtered(const Ctx& c, Shared* sh, int nwaves, int nblk, float Rb, float alpha, float ell_of_lists, int k) { lists_filtered_impl(c, sh, nwaves, nblk, Rb, alpha, ell_of_lists, k); }
__device__ __forceinline__ void resort_lists(const Ctx& c, const Lds& L, Shared* sh, int tile, int tid, int nthreads) { resort_lists_impl(c, L, sh, tile, tid, nthreads); }
static __device__ __noinline__ void phase_refine(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), tile = tgeo & 0x1FFF, y_lds = uni(y_lds_in), k = uni(k_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh;
    const Ctx c = make_ctx(Dp, g, G);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x, nwaves = nthreads >> 6, wave = tid >> 6;
    const int nblk = (c.nrows + 63) >> 6;
    const float r_c = sqrtf(gate_d2_align(sh->ell, sh->P.sp_thres, sh->P.sigma * sh->P.sigma));
    const float Rb = r_c * (1.0f + sh->P.skin);
    // Where the filtered lists are centred: at the current positions, or (DevParams::predict) a stretch ahead on the path, as phase_cull does -- if the old lists
    // hold everything within the new radius of THOSE positions too: the reach of phase_transform's test, taken to the extrapolated positions, point by point.
    float reach = sh->reach;
    float sh_rt[12]; bool shift = false;
    float Mn[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Mn[i] = sh->M[i];
    if (sh->P.predict > 0.f && sh->twist_ok) {
        float om[3], vv[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { om[q] = sh->omega[q]; vv[q] = sh->v[q]; }
        const float wn = norm3f(om), vn = norm3f(vv);
        float t = sh->P.predict * fminf(vn > 1.0e-12f ? r_c * sh->P.skin / vn : 1.0e9f, wn > 1.0e-12f ? list_alpha(sh->P, sh->ell) / wn : 1.0e9f);
        t = fminf(t, sh->P.predict_steps * sh->step);
        if (t > 0.f && t < 1.0e8f) {
            float dR[9], dT[3];
            exp_sek3(om, vv, t, dR, dT);
            // the extrapolated transform dR^T (M p - dT), and how far it takes the points from where they were listed
            float Mp[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int q = 0; q < 3; ++q) Mp[r * 4 + q] = sum3f(dR[0 * 3 + r] * Mn[0 * 4 + q], dR[1 * 3 + r] * Mn[1 * 4 + q], dR[2 * 3 + r] * Mn[2 * 4 + q]);
                Mp[r * 4 + 3] = sum3f(dR[0 * 3 + r] * (Mn[3] - dT[0]), dR[1 * 3 + r] * (Mn[7] - dT[1]), dR[2 * 3 + r] * (Mn[11] - dT[2]));
            }
            float Mbo[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) Mbo[i] = sh->Mb[i];
            const float alpha_o = sh->alpha_build;
            float far_l = 0.f;
            for (int j = tid; j < c.nm; j += nthreads) {
                const float4 pj = ld4(c.moving + lo_off(j));
                float n0, n1, n2, b0, b1, b2;
                apply_transform(Mp, pj.x, pj.y, pj.z, n0, n1, n2);
                apply_transform(Mbo, pj.x, pj.y, pj.z, b0, b1, b2);
                const float e0 = n0 - b0, e1 = n1 - b1, e2 = n2 - b2;
                const float disp = sqrtf(__builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0))) * 1.0001f + 1.0e-5f;
                const float far = sqrtf(__builtin_fmaf(b2, b2, __builtin_fmaf(b1, b1, b0 * b0))) * 0.9999f;
                far_l = fmaxf(far_l, disp - alpha_o * far);
            }
            far_l = block_max(far_l, sh, tid, nwaves);
            if ((Rb + far_l) * 1.00001f <= sh->Rb) {
                // ... and the CURRENT positions have to be within the reach of lists centred there (this iteration walks them): phase_transform's test of the
                // lists about to be made, with the margin they will have
                const float alpha_p = fmaxf(0.f, fminf(list_alpha(sh->P, sh->ell), 0.999f * (sh->Rb - (Rb + far_l) * 1.00001f) / (sh->xmax * 1.0001f + sh->Rb)));
                float off_l = 0.f;
                for (int j = tid; j < c.nm; j += nthreads) {
                    const float4 pj = ld4(c.moving + lo_off(j));
                    float n0, n1, n2, y0, y1, y2;
                    apply_transform(Mp, pj.x, pj.y, pj.z, n0, n1, n2);
                    apply_transform(Mn, pj.x, pj.y, pj.z, y0, y1, y2);
                    const float e0 = y0 - n0, e1 = y1 - n1, e2 = y2 - n2;
                    const float disp = sqrtf(__builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0))) * 1.0001f + 1.0e-5f;
                    const float far = sqrtf(__builtin_fmaf(n2, n2, __builtin_fmaf(n1, n1, n0 * n0))) * 0.9999f;
                    off_l = fmaxf(off_l, disp - alpha_p * far);
                }
                off_l = block_max(off_l, sh, tid, nwaves);
                if ((r_c + off_l) * 1.00002f <= Rb) {
                    shift = true; reach = far_l;
#pragma unroll
                    for (int i = 0; i < 9; ++i) sh_rt[i] = dR[i];
#pragma unroll
                    for (int q = 0; q < 3; ++q) sh_rt[9 + q] = dT[q];
#pragma unroll
                    for (int i = 0; i < 12; ++i) Mn[i] = Mp[i];
                }
            }
        }
    }
    // the depth-proportional margin the old lists leave room for (see phase_transform's test)
    const float alpha = fmaxf(0.f, fminf(list_alpha(sh->P, sh->ell), 0.999f * (sh->Rb - (Rb + reach) * 1.00001f) / (sh->xmax * 1.0001f + sh->Rb)));
    const int kept = y_lds == 1 ? refine_lists<1>(c, L, sh, Rb, alpha, shift, sh_rt, lane, wave, nwaves) : (y_lds == 2 ? refine_lists<2>(c, L, sh, Rb, alpha, shift, sh_rt, lane, wave, nwaves) : refine_lists<0>(c, L, sh, Rb, alpha, shift, sh_rt, lane, wave, nwaves));
    if (lane == 0) sh->wsum[wave] = kept;
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < 12; ++i) sh->Mb[i] = Mn[i];             // displacements count from here again
        if (shift) sh->predicted += 1;
        lists_filtered(c, sh, nwaves, nblk, Rb, alpha, sh->ell, k);
    }
    __syncthreads();
    if (!sh->resort) return;
    resort_lists(c, L, sh, tile, tid, nthreads);
}

// what follows an in-place filter of the lists (phase_refine, or the candidate walk that made the next ell's lists on its way): totals, the radius and margin the
// lists now stand for, and whether the rows are worth re-sorting.  One thread; sh->wsum[w] = entries wave w kept.
__device__ __forceinline__ void lists_filtered_impl(const Ctx& c, Shared* sh, int nwaves, int nblk, float Rb, float alpha, float ell_of_lists, int k) {
        int tot = 0, lmax_new = 0;
        long long walked = 0;                                        // list slots a walk evaluates in the present order: 64 rows x the longest list of each block, in steps of PF
        for (int w = 0; w < nwaves; ++w) tot += sh->wsum[w];
        for (int bq = 0; bq < nblk; ++bq) { const int lm = (int)sh->blk_lmax[bq]; lmax_new = max(lmax_new, lm); walked += 64 * PF * ((lm + PF - 1) / PF); }
        sh->total = tot; sh->lmax = lmax_new; sh->Rb = Rb; sh->alpha_build = alpha; sh->ell_build = ell_of_lists; sh->refines += 1;
        // Re-sort (below) when it pays: it costs about 25 us + 1.7 ns per list entry (measured: 31 us at 13 k entries, 68 at 48 k, 240 at
        // 126 k -- the lists change columns, 64 cache lines per wave load) and saves 0.28 ns per list slot no longer walked, in every
        // iteration until the next rebuild: those left at this ell by the schedule of cvo.cpp:810-812, 24 assumed at the last one.
        // The staging area is the (idle) record buffer: it has to hold the new lists in list layout.
        const int left = k < 3 ? 3 - k : (k < 10 ? 10 - k : (k < 20 ? 20 - k : 24));
        const float gain_ns = 0.28f * (float)left * ((float)walked - 1.3f * (float)tot), cost_ns = 25000.f + 1.7f * (float)tot;
        const int mode = sh->P.resort;                               // CVO_HIP_RESORT: 0 never, 1 by the cost model, 2 always
        sh->resort = (mode && lmax_new > 0 && (long long)((lmax_new + 1) & ~1) * c.rows_pad <= (long long)c.flat_cap && (mode == 2 || gain_ns > cost_ns)) ? 1 : 0;
}

// Every thread; the record buffer must be idle (between the line search of one iteration and the candidate phase of the next).
__device__ __forceinline__ void resort_lists_impl(const Ctx& c, const Lds& L, Shared* sh, int tile, int tid, int nthreads) {
    const int lane = tid & 63, nwaves = nthreads >> 6, wave = tid >> 6;
    const int nblk = (c.nrows + 63) >> 6;
    // The filter keeps a third to a half of every list, unevenly: rows that were neighbours in the old length order now differ by a factor
    // of two and more, and a wave walks its 64 rows for as long as the longest of them lasts (at ell = 0.03 the walk would evaluate 2.8 list
    // slots per listed candidate).  So the rows are sorted again by their new lengths and the lists move to their new slots -- through the
    // record buffer, which is idle between the line search of one iteration and the candidate phase of the next.
    typedef CVO_GLOBAL uint16_t gu16;
    gu16* slot_of = reinterpret_cast<gu16*>(c.jT4);                 // old slot of every local row (the cull's raw lists are dead by now)
    for (int sl = tid; sl < c.nrows; sl += nthreads) {
        const int li = L.row_of[sl];
        L.rowlen[li] = L.lenS[sl]; slot_of[li] = (uint16_t)sl;
    }
    __syncthreads();
    sort_slots(L, c.nrows, tid, nthreads);
    gv2u* stage = c.surv + c.fbase;
    const size_t rp = (size_t)c.rows_pad;
    for (int pass = 0; pass < 2; ++pass) {                          // 0: old slot -> staging area at the new slot; 1: back into the lists
        for (int bq = wave; bq < nblk; bq += nwaves) {
            const int sn = bq * 64 + lane;
            const int len = L.lenS[sn];
            int lw = len;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lw = max(lw, __shfl_xor(lw, off, 64));
            const int so = (pass == 0 && len > 0) ? (int)slot_of[L.row_of[sn]] : sn;
            const gv2u* src = (pass == 0 ? c.ent : stage) + 2 * so;
            gv2u* dst = (pass == 0 ? stage : c.ent) + 2 * sn;
            constexpr int MV = 8;
            for (int n0 = 0; n0 < lw; n0 += MV) {
                v2u e[MV];
#pragma unroll
                for (int u = 0; u < MV; ++u) if (n0 + u < len) e[u] = src[ent_ix(n0 + u, rp)];
#pragma unroll
                for (int u = 0; u < MV; ++u) if (n0 + u < len) dst[ent_ix(n0 + u, rp)] = e[u];
            }
        }
        __syncthreads();
    }
    finish_slots(c, L, sh, tile, tid, nthreads, false);
}

// the re-sort a candidate walk has asked for (it filtered the lists for the next ell on its way and found the rows worth re-sorting: Shared::resort_pending)
static __device__ __noinline__ void phase_resort(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), tile = tgeo & 0x1FFF, y_lds = uni(y_lds_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh;
    const Ctx c = make_ctx(Dp, g, G);
    resort_lists(c, L, sh, tile, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) sh->resort_pending = 0;
    __syncthreads();
}

CVO_PHASE_FN(8) void phase_candidates(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), y_lds = uni(y_lds_in), k = uni(k_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh;
    const Ctx c = make_ctx(Dp, g, G);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x, nwaves = nthreads >> 6, wave = tid >> 6;
    const int nrows = c.nrows;
    const unsigned long long ts0 = CVO_NOW();
    const Gates gates = *reinterpret_cast<const Gates*>(sh->gates_store);   // make_gates(sh->ell, sh->P), kept by the transform that precedes every walk at a new ell
    const bool dense_mode = sh->dense_mode != 0, fresh_list = sh->rebuild == 1;
    const float inv_c = sh->inv_c, inv_d = sh->inv_d;               // 1 / P.c, 1 / P.d
    const float ell_now = sh->ell;
    bool fused = false, shifted = false; float Rn_f = 0.f, alpha_f = 0.f, ell_f = 0.f, Mb_f[12], shift_f[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { Mb_f[i] = 0.f; shift_f[i] = 0.f; }
    double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};                      // omega[3], v[3], nnz, candidates
    const unsigned long long ts1 = CVO_NOW();
    if (!dense_mode) {
#define CVO_CAND(fn, flat) do { if (y_lds == 1) fn<1, flat>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8); \
                                else if (y_lds == 2) fn<2, flat>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8); \
                                else fn<0, flat>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8); } while (0)
#ifndef CVO_NO_FEATURE_STAGING
        if (fresh_list && y_lds == 1 && sh->tab_cols >= c.nm) {
            for (int j = tid; j < c.nm; j += nthreads) L.tab[j] = ld4(c.moving + hi_off(c.nm, j));
            __syncthreads();
            if (gates.poly_ok) cand_fresh<1, true, true>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8);
            else cand_fresh<1, false, true>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8);
        } else
#endif
        if (fresh_list) { if (gates.poly_ok) CVO_CAND(cand_fresh, true); else CVO_CAND(cand_fresh, false); }
        else if (y_lds == 1 && gates.poly_ok && sh->total < sh->P.nt_min) cand_steady<1, true, false>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8);   // short lists: plain loads (DevParams::nt_min)
        else {
            // Is this the last iteration at the present ell (cvo.cpp:810-812: the schedule goes by the iteration count), and do the lists hold everything within the
            // NEXT ell's list radius of the current positions (phase_transform's condition for filtering them in place)?  Then this walk makes the next lists itself.
            float ell_next = ell_now;
            ell_next = (k > 2) ? (float)0.10 : ell_next; ell_next = (k > 9) ? (float)0.06 : ell_next; ell_next = (k > 19) ? (float)0.03 : ell_next;
            if (sh->P.fuse_refine && y_lds != 0 && gates.poly_ok && ell_next != ell_now && sh->list_valid && sh->ell_build == ell_now && k + 1 < sh->P.max_iter) {
                const float rcn = sqrtf(gate_d2_align(ell_next, sh->P.sp_thres, sh->P.sigma * sh->P.sigma));
                Rn_f = rcn * (1.0f + sh->P.skin);
                const float room = sh->Rb - (Rn_f + sh->reach_now) * 1.00001f;
                if (room >= 0.f) {
                    fused = true;
                    float reach_f = sh->reach_now;
                    ell_f = ell_next;
#pragma unroll
                    for (int i = 0; i < 12; ++i) Mb_f[i] = sh->M[i];
                    // the next lists centred a stretch ahead on the path (phase_refine does the same for a filter pass of its own): the twist of the previous iteration
                    // once more; the old lists must hold everything within the next radius of THOSE positions -- checked point by point.  (That the positions of the
                    // first use, one pose update from here, are within reach of lists centred there is phase_transform's test at the start of the next iteration.)
                    if (sh->P.predict > 0.f && sh->twist_ok) {
                        float om[3], vv[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) { om[q] = sh->omega[q]; vv[q] = sh->v[q]; }
                        const float wn = norm3f(om), vn = norm3f(vv);
                        float t = sh->P.predict * fminf(vn > 1.0e-12f ? rcn * sh->P.skin / vn : 1.0e9f, wn > 1.0e-12f ? list_alpha(sh->P, ell_next) / wn : 1.0e9f);
                        t = fminf(t, sh->P.predict_steps * sh->step);
                        if (t > 0.f && t < 1.0e8f) {
                            float dR[9], dT[3], Mp[12], Mbo[12];
                            exp_sek3(om, vv, t, dR, dT);
#pragma unroll
                            for (int r = 0; r < 3; ++r) {
#pragma unroll
                                for (int q = 0; q < 3; ++q) Mp[r * 4 + q] = sum3f(dR[0 * 3 + r] * Mb_f[0 * 4 + q], dR[1 * 3 + r] * Mb_f[1 * 4 + q], dR[2 * 3 + r] * Mb_f[2 * 4 + q]);
                                Mp[r * 4 + 3] = sum3f(dR[0 * 3 + r] * (Mb_f[3] - dT[0]), dR[1 * 3 + r] * (Mb_f[7] - dT[1]), dR[2 * 3 + r] * (Mb_f[11] - dT[2]));
                            }
#pragma unroll
                            for (int i = 0; i < 12; ++i) Mbo[i] = sh->Mb[i];
                            const float alpha_o = sh->alpha_build;
                            float far_l = 0.f;
                            for (int j = tid; j < c.nm; j += nthreads) {
                                const float4 pj = ld4(c.moving + lo_off(j));
                                float n0, n1, n2, b0, b1, b2;
                                apply_transform(Mp, pj.x, pj.y, pj.z, n0, n1, n2);
                                apply_transform(Mbo, pj.x, pj.y, pj.z, b0, b1, b2);
                                const float e0 = n0 - b0, e1 = n1 - b1, e2 = n2 - b2;
                                const float disp = sqrtf(__builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0))) * 1.0001f + 1.0e-5f;
                                const float far = sqrtf(__builtin_fmaf(b2, b2, __builtin_fmaf(b1, b1, b0 * b0))) * 0.9999f;
                                far_l = fmaxf(far_l, disp - alpha_o * far);
                            }
                            far_l = block_max(far_l, sh, tid, nwaves);
                            if ((Rn_f + far_l) * 1.00001f <= sh->Rb) {
                                shifted = true; reach_f = far_l;
#pragma unroll
                                for (int i = 0; i < 9; ++i) shift_f[i] = dR[i];
#pragma unroll
                                for (int q = 0; q < 3; ++q) shift_f[9 + q] = dT[q];
#pragma unroll
                                for (int i = 0; i < 12; ++i) Mb_f[i] = Mp[i];
                            }
                        }
                    }
                    alpha_f = fmaxf(0.f, fminf(list_alpha(sh->P, ell_next), 0.999f * (sh->Rb - (Rn_f + reach_f) * 1.00001f) / (sh->xmax * 1.0001f + sh->Rb)));
                }
            }
            if (fused) {
                float srt[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) srt[i] = uni_f(shift_f[i]);
                // (the shift as a flag of its own: the compiler folds a null test of a pointer to a promoted local array the wrong way)
                if (y_lds == 1) cand_steady<1, true, true, true>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8, Rn_f, alpha_f, shifted, srt);
                else cand_steady<2, true, true, true>(c, L, sh, gates, lane, wave, nwaves, inv_c, inv_d, acc8, Rn_f, alpha_f, shifted, srt);
            }
            else if (gates.poly_ok) CVO_CAND(cand_steady, true); else CVO_CAND(cand_steady, false);
        }
#undef CVO_CAND
        if (tid == 0) acc8[7] = (double)sh->total;
    } else {
        for (int li = tid; li < nrows; li += nthreads) {            // dense fallback: every column of the row
            const int i = global_row(c, li);
            const float4 lo = ld4(c.fixed + lo_off(i)), hi = ld4(c.fixed + hi_off(c.nf, i));
            const float xi[3] = {lo.x, lo.y, lo.z};
            const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
            float sw[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
            int nz = 0;
            for (int j = 0; j < c.nm; ++j) {
                float4 yj = load_y_rt(c, L, y_lds, j);
                if (y_lds == 2) yj.w = ld4(c.moving + lo_off(j)).w;
                const float a = se_kernel_value(xi, fi, yj, ld4(c.moving + hi_off(c.nm, j)), gates);
                if (a > 0.f) {
                    const float yv[3] = {yj.x, yj.y, yj.z};
                    float cr[3]; cross3(xi, yv, cr);
                    sw[0] += a * cr[0]; sw[1] += a * cr[1]; sw[2] += a * cr[2];
                    sv[0] += a * (yv[0] - xi[0]); sv[1] += a * (yv[1] - xi[1]); sv[2] += a * (yv[2] - xi[2]);
                    ++nz;
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) { acc8[q] += (double)(inv_c * sw[q]); acc8[3 + q] += (double)(inv_d * sv[q]); }
            acc8[6] += (double)nz;
            acc8[7] += (double)(y_lds == 2 ? c.nm : (int)L.rowlen[li]);   // statistics only (in the plane layout the row lengths have been overwritten by now)
        }
    }
    const unsigned long long ts2 = CVO_NOW();
#ifdef CVO_KTRACE_WAVES   // experiment builds: when every wave left its walk (ticks since the walk's start), for the trace row written by run_pair
    if (lane == 0) sh->kabs[wave] = ts2 - ts1;
#endif
    // (no barrier before the reduction: its own barrier is the one every wave reaches after its walk)
    const double mine = dense_mode ? block_reduce<8, 8, false>(acc8, sh, tid, nwaves) : block_reduce<8, 6, false>(acc8, sh, tid, nwaves);   // list mode: nnz and candidates are per-wave counts in lane 0
    const unsigned long long ts3 = CVO_NOW();
    if (G > 1) {
        if (tid < 64) { if (!group_exchange<8>(sh, c.xch, G, g, sh->launch_tag | (2u * (unsigned)k + 1u), lane)) sh->status = 6; }   // wave 0: reads the totals its own lanes wrote
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) {
            for (int q = 0; q < 3; ++q) { sh->omega[q] = (float)sh->vals[q]; sh->v[q] = (float)sh->vals[3 + q]; }   // cvo.cpp:234-235
            sh->nnz = (int)sh->vals[6]; sh->cand = (int)sh->vals[7];
        }
    } else {                                                        // one workgroup per pair: the summing lanes publish their totals themselves
        if (tid < 3) sh->omega[tid] = (float)mine;                  // cvo.cpp:234-235
        else if (tid < 6) sh->v[tid - 3] = (float)mine;
        else if (tid == 6) sh->nnz = (int)mine;
        else if (tid == 7) sh->cand = (int)mine;
    }
    if (tid == 0) {
        if (fused) {                                                 // (behind the reduction's barrier: every wave has left its kept count in sh->wsum)
            for (int i = 0; i < 12; ++i) sh->Mb[i] = Mb_f[i];       // the new lists are those of the current positions, or of the positions a stretch ahead
            if (shifted) sh->predicted += 1;
            lists_filtered(c, sh, nwaves, (c.nrows + 63) >> 6, Rn_f, alpha_f, ell_f, k + 1);
            sh->resort_pending = sh->resort; sh->fused += 1;
        }
        const unsigned long long ts4 = CVO_NOW();
        sh->sub[0] += ts1 - ts0; sh->sub[1] += ts2 - ts1; sh->sub[2] += ts3 - ts2; sh->sub[3] += ts4 - ts3;
    }
    __syncthreads();
}

// ---- L: compute_step_size sums (cvo.cpp:239-315); f64 terms, one lane per nonzero
template <bool EVEN>
__device__ __forceinline__ void linesearch_body(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), y_lds = uni(y_lds_in), k = uni(k_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh;
    const Ctx c = make_ctx(Dp, g, G);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x, nwaves = nthreads >> 6;
    float omega[3], v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { omega[q] = sh->omega[q]; v[q] = sh->v[q]; }
    LsConsts ls = make_ls(omega, v, sh->ell);
    {   // the same in every lane: keep the ~50 constants in scalar registers, not in each lane's vector registers
        float* f = reinterpret_cast<float*>(&ls);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(LsConsts) / sizeof(float)); ++q) f[q] = uni_f(f[q]);
    }
    double acc4[4] = {0, 0, 0, 0};
#ifdef CVO_KTRACE
    const unsigned long long kt0 = CVO_NOW();
#endif
    if (!sh->dense_mode) {
        // every wave walks the nonzeros it compacted itself: the candidate phase deals the rows so that the waves' shares are
        // near equal, and a wave's segment is one contiguous run
        const bool x_lds = sh->x_lds != 0;
        const int wave = tid >> 6;
        int cnt_wg = 0;
        for (int w = 0; w < nwaves; ++w) cnt_wg += sh->wcnt[w];
        // The workgroup's nonzeros lie in one segment per wave of the candidate walk (whose rows are dealt so that the segments are near equal -- when every wave
        // had blocks to walk: a member of a pair on eight workgroups owns six blocks for eight waves, the first twice as long as the last).  The sums are order-free
        // (cvo.cpp:309-314): every wave takes an EQUAL share of the records, wherever they lie -- a stretch of one segment, or the end of one and the start of the next.
        // (Only where the deal could not balance -- fewer than two blocks per wave -- and there is enough to share: with the waves' segments near equal, or a few
        // hundred records in all, a second stretch per wave costs more than it evens out: -1.5 % under load, +5 % on a tracker's frame, measured.)
        constexpr bool even = EVEN;                                   // (decided by the caller: ls_even_shares)
        const int share_lo = (int)(((long long)cnt_wg * wave) / nwaves), share_hi = (int)(((long long)cnt_wg * (wave + 1)) / nwaves);
        int cnt_w = 0; const gv2u* sp = c.surv + c.fbase;             // the stretch being walked (set per segment below)
        // With several nonzeros per column the last quarter of the point part ({2*z4, |z2|^2 + 2 z1.z3}: 33 of a nonzero's ~120
        // instructions) is tabulated once per iteration, 16 bytes per column in LDS over the rebuild scratch.  (Tabulating all of
        // the point part was measured: its four 16-byte gathers per nonzero make the LDS the bottleneck -- 88 against 118 cycles
        // per 64 nonzeros -- and two table passes need the records binned by column in the candidate phase; one gather does not.)
        const bool use_table = sh->tab_cols >= c.nm && cnt_wg > CVO_LS_TAB_FACTOR * c.nm && y_lds != 0;
        if (use_table) {
            for (int j = tid; j < c.nm; j += nthreads) {
                const float4 yj = y_lds == 1 ? load_y<1>(c, L, j) : load_y<2>(c, L, j);
                float4 t0, t1, t2, t3;
                ls_point<true>(yj, ls, t0, t1, t2, t3);
                L.tab[j] = make_float4(t3.x, t3.y, t3.z, t2.w);
            }
            __syncthreads();
        }
        // records stream from L2 / HBM: four steps of them are in flight per lane (under load one step's arithmetic is shorter
        // than a memory round trip)
#ifndef CVO_LS_RD
#define CVO_LS_RD 4
#endif
        constexpr int RD = CVO_LS_RD;
        auto walk = [&](auto ym, auto tb) {
            constexpr int YM = decltype(ym)::value;
            constexpr bool TAB = decltype(tb)::value;
            if (cnt_w <= 0) return;
            v2u ring[RD];
#pragma unroll
            for (int u = 0; u < RD; ++u) ring[u] = ld_rec(&sp[min(lane + 64 * u, cnt_w - 1)]);
            for (int q0 = lane; q0 < cnt_w; q0 += 64 * RD) {
                prio_by_remaining((cnt_w - uni(q0)) >> 11);        // units of eight trips (2048 records): the waves' shares are near equal, the one with more left goes first
#pragma unroll
                for (int u = 0; u < RD; ++u) {
                    const int q = q0 + 64 * u;
                    const v2u rec = ring[u];
                    ring[u] = ld_rec(&sp[min(q + 64 * RD, cnt_w - 1)]);
                    if (q < cnt_w) {
                        const int slot = (int)(rec.y >> 16), j = (int)(rec.y & 0xFFFFu);
                        float xi[3]; load_x(c, L, x_lds, slot, xi);
                        const float4 yj = load_y<YM>(c, L, j);
                        if (TAB) {
                            const float4 tq = L.tab[j];
                            float4 t0, t1, t2, t3;
                            ls_point<false>(yj, ls, t0, t1, t2, t3);
                            t2.w = tq.w; t3 = make_float4(tq.x, tq.y, tq.z, 0.f);
                            const float df[3] = {xi[0] - yj.x, xi[1] - yj.y, xi[2] - yj.z};            // cvo.cpp:286
                            ls_pair(df, __uint_as_float(rec.x), t0, t1, t2, t3, ls, acc4[0], acc4[1], acc4[2], acc4[3]);
                        } else {
                            ls_terms(xi, yj, __uint_as_float(rec.x), ls, acc4[0], acc4[1], acc4[2], acc4[3]);
                        }
                    }
                }
            }
        };
        using T1 = std::true_type; using T0 = std::false_type;
        auto dispatch = [&]() {
            if (use_table) { if (y_lds == 1) walk(std::integral_constant<int, 1>{}, T1{}); else walk(std::integral_constant<int, 2>{}, T1{}); }
            else if (y_lds == 1) walk(std::integral_constant<int, 1>{}, T0{}); else if (y_lds == 2) walk(std::integral_constant<int, 2>{}, T0{}); else walk(std::integral_constant<int, 0>{}, T0{});
        };
        if (!even) {                                                   // the wave's own segment (the common path: as it always was)
            cnt_w = sh->wcnt[wave]; sp = c.surv + c.fbase + (size_t)sh->wbase[wave];
            dispatch();
        } else {
            int seg_first = 0;                                         // records before segment s, in the order of the waves
            for (int s = 0; s < nwaves; ++s) {
                const int seg_n = uni((int)sh->wcnt[s]);
                const int lo = max(share_lo, seg_first) - seg_first, hi = min(share_hi, seg_first + seg_n) - seg_first;
                seg_first += seg_n;
                if (lo >= hi) continue;
                cnt_w = hi - lo; sp = c.surv + c.fbase + (size_t)sh->wbase[s] + lo;
                if (use_table) walk(std::integral_constant<int, 1>{}, T1{}); else walk(std::integral_constant<int, 1>{}, T0{});   // (the resident float4 layout only: two instances more, not five)
            }
        }
    } else {
        const Gates gates = make_gates(sh->ell, sh->P);
        for (int li = tid; li < c.nrows; li += nthreads) {
            const int i = global_row(c, li);
            const float4 lo = ld4(c.fixed + lo_off(i)), hi = ld4(c.fixed + hi_off(c.nf, i));
            const float xi[3] = {lo.x, lo.y, lo.z};
            const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
            double Bi = 0, Ci = 0, Di = 0, Ei = 0;
            for (int j = 0; j < c.nm; ++j) {
                float4 yj = load_y_rt(c, L, y_lds, j);
                if (y_lds == 2) yj.w = ld4(c.moving + lo_off(j)).w;
                const float A_ij = se_kernel_value(xi, fi, yj, ld4(c.moving + hi_off(c.nm, j)), gates);
                if (A_ij > 0.f) ls_terms(xi, yj, A_ij, ls, Bi, Ci, Di, Ei);
            }
            acc4[0] += Bi; acc4[1] += Ci; acc4[2] += Di; acc4[3] += Ei;
        }
    }
#ifdef CVO_KTRACE
    const unsigned long long kt1 = CVO_NOW();
#endif
    // no barrier before the reduction (its own is the one every wave reaches after its walk: the epilogue's transform may overwrite the
    // resident cloud only behind it) and none after it: the totals are read by thread 0 of wave 0 alone (exchange, epilogue), whose
    // own lanes wrote them; the next workgroup barrier is the epilogue's
    block_reduce<4, 4, false>(acc4, sh, tid, nwaves);
#ifdef CVO_KTRACE
#ifndef CVO_KTRACE_EPI
    if (tid == 0) { sh->ksub[0] = kt1 - kt0; sh->ksub[1] = CVO_NOW() - kt1; }
#endif
#endif
    if (G > 1) {
        if (tid < 64) { if (!group_exchange<4>(sh, c.xch, G, g, sh->launch_tag | (2u * (unsigned)k + 2u), lane)) sh->status = 6; }
        __syncthreads();                                             // sh->status is read by every thread right after the phase
    }
}

// every wave an equal share of the workgroup's nonzero records instead of the segment it compacted itself?  Where the deal could not balance the segments (fewer than
// two blocks per wave: a member of a pair on many workgroups) and there is enough to share; the resident float4 layout only.
__device__ __forceinline__ bool ls_even_shares(int y_lds) {
    const Shared* sh = reinterpret_cast<const Shared*>(cvo_smem);
    const int nwaves = (int)blockDim.x >> 6;
    if (!CVO_LS_EVEN || y_lds != 1 || sh->dense_mode || ((sh->ctx_nrows + 63) >> 6) >= 2 * nwaves) return false;
    int cnt_wg = 0;
    for (int w = 0; w < nwaves; ++w) cnt_wg += sh->wcnt[w];
    return cnt_wg >= CVO_LS_EVEN_MIN * nwaves;
}
CVO_PHASE_FN(1) void phase_linesearch(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in) { linesearch_body<false>(Dp_in, g_in, G_in, tile_in, y_lds_in, k_in); }
// (a function of its own whatever the build: the common path's function does not carry its code)
static __device__ __noinline__ void phase_linesearch_even(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in) { linesearch_body<true>(Dp_in, g_in, G_in, tile_in, y_lds_in, k_in); }

// ---- E: one lane finishes the iteration (every workgroup of the pair computes the same bits)
CVO_PHASE_FN(2) void phase_epilogue(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in, int max_iter_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), y_lds = uni(y_lds_in), k = uni(k_in), max_iter = uni(max_iter_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh;
    const Ctx c = make_ctx(Dp, g, G);
    // every lane fetches its share of the moving cloud for the NEXT iteration's transform while lane 0 does the scalar work
    // In the resident float4 layout waves 1 .. n-1 take all the points: they transform them while lane 0 of wave 0 is still at the second stop test (dist_se3,
    // a third of its scalar work), which the transform does not need -- only R and T.
    const bool next_T = k + 1 < max_iter;
    const int first_worker = ((y_lds == 1 || (y_lds == 2 && CVO_FUSE_PLANES)) && blockDim.x > 64 && sh->P.overlap_stop_test) ? 64 : 0;
    float4 pre[PRE_T];
    {
        int j = (int)threadIdx.x >= first_worker ? (int)threadIdx.x - first_worker : c.nm;
#pragma unroll
        for (int u = 0; u < PRE_T; ++u) { pre[u] = (j < c.nm) ? ld4(c.moving + lo_off(j)) : make_float4(0.f, 0.f, 0.f, 0.f); j += (int)blockDim.x - first_worker; }
    }
#ifdef CVO_KTRACE
    const unsigned long long ke0 = CVO_NOW();
#endif
    float4 preb[PRE_T];
    if (first_worker && next_T && (int)threadIdx.x >= first_worker && sh->list_valid) {   // (the waves that wait for lane 0 anyway)
        float Mb[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Mb[i] = sh->Mb[i];
        const float alpha_b = sh->alpha_build;
#pragma unroll
        for (int u = 0; u < PRE_T; ++u) {
            float b0, b1, b2;
            apply_transform(Mb, pre[u].x, pre[u].y, pre[u].z, b0, b1, b2);
            preb[u] = make_float4(b0, b1, b2, alpha_b * (fast_sqrt(__builtin_fmaf(b2, b2, __builtin_fmaf(b1, b1, b0 * b0))) * 0.9999f));
        }
    }
    // Lane 0's scalar work in two parts with a workgroup barrier between them (every wave passes it once): A = step, first stop test, pose update -- what the
    // next transform needs; B = second stop test (dist_se3), ell schedule, adoption word, trace.
    unsigned long long aword = 0;
    float step = 0.f, ell = 0.f, dR[9], dT[3], omega[3], v[3];
    double B = 0, C = 0, Dd = 0, E = 0;
    bool stop_a = false;
    if (threadIdx.x == 0) {
        const DevParams& P = sh->P;
        // has a finished workgroup of the launch asked to help with this pair?  (the load returns under the scalar work below)
        if (sh->adopt_word) aword = __hip_atomic_load((gu64*)sh->adopt_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        B = sh->vals[0]; C = sh->vals[1]; Dd = sh->vals[2]; E = sh->vals[3];
        for (int q = 0; q < 3; ++q) { omega[q] = sh->omega[q]; v[q] = sh->v[q]; }
        ell = sh->ell;
        const float c3 = (float)(4.0 * float(E)), c2 = (float)(3.0 * float(Dd)), c1 = (float)(2.0 * float(C)), c0 = float(B);   // cvo.cpp:318
#ifdef CVO_KTRACE_EPI
        const unsigned long long kq0 = CVO_NOW();
#endif
        step = cubic_step(c3, c2, c1, c0, P.min_step);
#ifdef CVO_KTRACE_EPI
        __builtin_amdgcn_sched_barrier(0); sh->ksub[0] = CVO_NOW() - kq0 + (step == 12345.f ? 1 : 0);
#endif
        stop_a = norm3f(omega) < P.eps && norm3f(v) < P.eps;                            // cvo.cpp:782
        if (!stop_a) {
            float R[9], T[3], RdT[3], Rn[9];
            for (int i = 0; i < 9; ++i) R[i] = sh->R[i];
            for (int i = 0; i < 3; ++i) T[i] = sh->T[i];
            exp_sek3(omega, v, step, dR, dT);                                           // cvo.cpp:793
            mat3_vec(R, dT, RdT);
            for (int q = 0; q < 3; ++q) sh->T[q] = RdT[q] + T[q];                        // cvo.cpp:800
            mat3_mul(R, dR, Rn);
            for (int i = 0; i < 9; ++i) sh->R[i] = Rn[i];                                // cvo.cpp:801
        }
        sh->stop = 0;
#ifdef CVO_KTRACE_EPI
        __builtin_amdgcn_sched_barrier(0); sh->ksub[1] = CVO_NOW() - kq0 - sh->ksub[0];
#if CVO_KTRACE_EPI == 2
        sh->kabs[0] = CVO_NOW();
#endif
#endif
    }
    __syncthreads();                                                 // R, T are out
#if defined(CVO_KTRACE_EPI) && CVO_KTRACE_EPI == 2
    if (threadIdx.x == 64) sh->kabs[4] = CVO_NOW();
#endif
    if (threadIdx.x == 0) {
#ifdef CVO_KTRACE_EPI
        const unsigned long long kq2 = CVO_NOW();
#endif
        const DevParams& P = sh->P;
        float dist = -1.f;
        int stop = 0;
        if (stop_a) stop = 1;
        else {
            dist = dist_se3(dR, dT);
            if (dist < P.eps_2) stop = 1;                                               // cvo.cpp:804
        }
        if (stop) sh->iter_at_break = k;
        else {
            float l = ell;                                                              // cvo.cpp:810-812
            l = (k > 2) ? (float)0.10 : l;
            l = (k > 9) ? (float)0.06 : l;
            l = (k > 19) ? (float)0.03 : l;
            sh->ell = l;
        }
        sh->stop = stop; sh->step = step; sh->dist = dist; sh->twist_ok = 1;
        sh->adopt_req = ((unsigned)(aword >> 32) == (sh->launch_tag | ADOPT_REQUEST)) ? 1 + (int)(unsigned)aword : 0;
        if (sh->adopt_word && (unsigned)(aword >> 32) == (sh->launch_tag | ADOPT_FREE) && !stop) {   // tell would-be helpers how far this pair has come (a CAS: an offer made meanwhile stays)
            unsigned long long e = aword;
            (void)__hip_atomic_compare_exchange_strong((gu64*)sh->adopt_word, &e, ((unsigned long long)(sh->launch_tag | ADOPT_FREE) << 32) | (unsigned)(k + 1),
                                                       __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (g == 0 && c.trace && k < c.trace_cap) {
            TraceRow& tr = c.trace[k];
            for (int q = 0; q < 3; ++q) { tr.omega[q] = omega[q]; tr.v[q] = v[q]; }
            tr.nnz = sh->nnz; tr.candidates = sh->cand; tr.B = B; tr.C = C; tr.D = Dd; tr.E = E;
            tr.step = step; tr.ell = ell; tr.dist = dist; tr.pad_ = 0;
            *c.trace_len = k + 1;
        }
#ifdef CVO_KTRACE_EPI
        __builtin_amdgcn_sched_barrier(0); sh->ksub[2] = CVO_NOW() - kq2 + (dist == 12345.f ? 1 : 0);
#if CVO_KTRACE_EPI == 2
        sh->kabs[1] = CVO_NOW();
#endif
#endif
    }
#ifdef CVO_KTRACE
    const unsigned long long ke1 = CVO_NOW();
#endif
    if (first_worker && next_T) {
        // T of iteration k+1 (cvo.cpp:770-771), begun before this iteration's second stop test is known: should it fire, the transformed cloud is simply not
        // used (the state written back is R, T; Shared::M keeps the transform of the last executed iteration).  Its barriers publish lane 0's part B.
        if (y_lds == 1) transform_body_t<1>(c, L, sh, pre, true, first_worker, true, sh->list_valid != 0, preb);
        else transform_body_t<2>(c, L, sh, pre, true, first_worker, true, sh->list_valid != 0, preb);   // the 12-byte planes of large clouds (round 5: fused as well)
    } else {
        __syncthreads();                                             // sh->stop and the rest of lane 0's results, for everyone
        if (!sh->stop && next_T) {                                   // T of iteration k+1 (cvo.cpp:770-771)
            if (y_lds == 1) transform_body_t<1>(c, L, sh, pre, true); else transform_large(Dp, g, G, tgeo, y_lds);
        }
    }
#ifdef CVO_KTRACE
#ifdef CVO_KTRACE_EPI
    if (threadIdx.x == 0) sh->ksub[3] = CVO_NOW() - ke0;
#if CVO_KTRACE_EPI == 2     // times from the end of lane 0's part A: part B done | thread 64 past the barrier | its points done | lane 0 has the staleness maximum | decision made | end
    if (threadIdx.x == 0) {
        const unsigned long long a = sh->kabs[0], e = CVO_NOW();
        sh->ksub[0] = a - ke0; sh->ksub[1] = sh->kabs[1] - a; sh->ksub[2] = sh->kabs[5] - a; sh->ksub[3] = sh->kabs[2] - a;
        sh->kabs[6] = sh->kabs[3] - a; sh->kabs[7] = e - a; sh->kabs[4] = sh->kabs[4] - a; sh->kabs[10] = sh->kabs[8] - a; sh->kabs[11] = sh->kabs[9] - a;
    }
#endif
#else
    if (threadIdx.x == 0) { sh->ksub[2] = ke1 - ke0; sh->ksub[3] = CVO_NOW() - ke1; }
#endif
#endif
}

#if CVO_INLINE_PHASES & 8
// one iteration's three regular phases as ONE function: a phase call costs its callee-saved saves (the line search keeps ~50 constants in callee-saved scalar
// registers, the candidate walk uses two dozen callee-saved vector registers) and, at its return, a wait for their reloads from scratch memory
static __device__ __noinline__ void phase_iteration(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in, int max_iter_in) {
    Shared* sh = reinterpret_cast<Shared*>(cvo_smem);
    const int tid = threadIdx.x;
    unsigned long long t_prev = CVO_NOW();
    phase_candidates(Dp_in, g_in, G_in, tile_in, y_lds_in, k_in);
    if (tid == 0) { atomicAdd(&sh->cand_total, (unsigned long long)sh->cand); atomicAdd(&sh->nnz_total, (unsigned long long)sh->nnz); }
    { const unsigned long long t_now = CVO_NOW(); if (tid == 0) atomicAdd(&sh->ticks[1], t_now - t_prev); t_prev = t_now; }
    if (sh->status != 0) return;
    if (ls_even_shares(uni(y_lds_in))) phase_linesearch_even(Dp_in, g_in, G_in, tile_in, y_lds_in, k_in); else phase_linesearch(Dp_in, g_in, G_in, tile_in, y_lds_in, k_in);
    { const unsigned long long t_now = CVO_NOW(); if (tid == 0) atomicAdd(&sh->ticks[3], t_now - t_prev); t_prev = t_now; }
    if (sh->status != 0) return;
    phase_epilogue(Dp_in, g_in, G_in, tile_in, y_lds_in, k_in, max_iter_in);
    { const unsigned long long t_now = CVO_NOW(); if (tid == 0) atomicAdd(&sh->ticks[5], t_now - t_prev); }
}
#endif

// ---- Tail: the tracker's score block (cvo::compute_innerproduct, cvo.cpp:475-503) for the pair this workgroup has just aligned, from
// what is resident anyway.  inn_post = fip(T moving, fixed) and se3_Hessian(T moving, fixed) (cvo.cpp:491, 500) are sums over the pairs
// within the radius at the ell the alignment left behind (Q1) -- a subset of the candidate lists, once the cloud has been transformed
// with the FINAL transform (cvo.cpp:485-487, 817) and the lists are still valid for it; no a > sp_thres test here (Q6).  inn_pre =
// fip(moving, fixed) (cvo.cpp:489) takes one cull of the untransformed cloud at that radius.  fip(fixed, fixed), fip(moving, moving)
// (cvo.cpp:496-497) come from the clouds' tables when they are there.  The pair arithmetic is the score kernel's (cvo_score_kernels.hip:
// un-fused d2, double exp with the division as written).  Whatever cannot be answered here (lists stale, a helped pair, a cloud not
// resident in LDS as float4, a cold table) is left to the host, which runs the score kernel for it.  A pair run by several workgroups
// (cooperative launch, helped pair) adds its members' sums up through the pair's exchange area.
__device__ __forceinline__ void score_pair_terms(const float (&pa)[3], const float (&fa)[5], const float (&pb)[3], const float (&fb)[5], float d2, float d2c_thres,
                                                 float sig2, float csig2, double den_l, double den_c, float il2, double& sumA, int& count, float (&H)[21], int& hcount) {
    float t[5];
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) { const float e = fa[cc] - fb[cc]; t[cc] = e * e; }
    const float d2c = (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
    if (!(d2c < d2c_thres)) return;                                                     // cvo.cpp:428 / 659
    const float k = (float)((double)sig2 * exp((double)(-d2) / den_l));                // cvo.cpp:429 / 661
    const float ck = (float)((double)csig2 * exp((double)(-d2c) / den_c));             // cvo.cpp:430
    sumA += ck * k; count += 1;                                                        // cvo.cpp:432-435
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) t[cc] = fa[cc] * fb[cc];
    const float cdot = (t[0] + t[1]) + (t[2] + (t[3] + t[4]));                          // cvo.cpp:662
    float cr[3]; cross3(pa, pb, cr);
    const float dot1 = pa[1] * pb[1] + pa[2] * pb[2], dot2 = pa[0] * pb[0] + pa[2] * pb[2], dot3 = pa[0] * pb[0] + pa[1] * pb[1];
    const float db[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    float Bq[21];
    Bq[0] = il2 * cr[0] * cr[0] - dot1;                                                // block A, cvo.cpp:670-675
    Bq[1] = (float)(il2 * cr[0] * cr[1] + 0.5 * (pa[0] * pb[1] + pa[1] * pb[0]));
    Bq[2] = (float)(il2 * cr[0] * cr[2] + 0.5 * (pa[0] * pb[2] + pa[2] * pb[0]));
    Bq[3] = il2 * cr[1] * cr[1] - dot2;
    Bq[4] = (float)(il2 * cr[1] * cr[2] + 0.5 * (pa[1] * pb[2] + pa[2] * pb[1]));
    Bq[5] = il2 * cr[2] * cr[2] - dot3;
    Bq[6] = il2 * cr[0] * db[0];          Bq[7] = -pa[2] + il2 * db[0] * cr[1];  Bq[8] = pa[1] + il2 * db[0] * cr[2];    // block C, cvo.cpp:680-688
    Bq[9] = pa[2] + il2 * db[1] * cr[0];  Bq[10] = il2 * cr[1] * db[1];          Bq[11] = -pa[0] + il2 * db[1] * cr[2];
    Bq[12] = -pa[1] + il2 * db[2] * cr[0]; Bq[13] = pa[0] + il2 * db[2] * cr[1]; Bq[14] = il2 * cr[2] * db[2];
    Bq[15] = il2 * db[0] * db[0] - 1; Bq[16] = il2 * db[0] * db[1]; Bq[17] = il2 * db[0] * db[2];                        // block D, cvo.cpp:692-697
    Bq[18] = il2 * db[1] * db[1] - 1; Bq[19] = il2 * db[1] * db[2]; Bq[20] = il2 * db[2] * db[2] - 1;
    const float wgt = il2 * cdot * k;                                                  // cvo.cpp:707
#pragma unroll
    for (int q2 = 0; q2 < 21; ++q2) H[q2] += wgt * Bq[q2];
    hcount += 1;
}
// eight partial sums over the workgroup and, for a pair run by several workgroups (a cooperative launch, a helped pair), over its members
__device__ __forceinline__ double tail_reduce8(double (&v)[8], Shared* sh, const Ctx& c, int G, int g, unsigned epoch, int tid, int nwaves) {
    double r = block_reduce<8>(v, sh, tid, nwaves);
    if (G > 1) {
        if (tid < 64) { if (!group_exchange<8>(sh, c.xch, G, g, epoch, tid & 63)) sh->status = 6; }
        __syncthreads();
        r = tid < 8 ? sh->vals[tid] : 0.0;
        __syncthreads();
    }
    return r;
}
static __device__ __noinline__ void phase_tail_scores(const PairDesc* Dp_in, int g_in, int G_in, int tile_in, int y_lds_in, int k_in) {
    const PairDesc* Dp = uni_ptr(Dp_in); const int g = uni(g_in), G = uni(G_in), tgeo = uni(tile_in), y_lds = uni(y_lds_in), k_done = uni(k_in);
    const Lds L = lds_layout(tgeo, y_lds); Shared* sh = L.sh;
    const Ctx c = make_ctx(Dp, g, G);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x, nwaves = nthreads >> 6, wave = tid >> 6;
    CVO_GLOBAL double* out = (CVO_GLOBAL double*)Dp->score_out;
    const unsigned ep0 = sh->launch_tag | (2u * (unsigned)k_done + 3u);                 // exchange epochs behind the loop's (2k+1, 2k+2 of its last iteration)
    int answered = 0;
    const DevParams P = sh->P;
    const float ell = sh->ell;                                                          // what align() left behind (Q1): cvo.cpp:395, 626
    const float d2_thres = gate_d2_score(ell, P.sp_thres, P.sigma), d2c_thres = gate_d2c(P.c_ell, P.sp_thres, P.c_sigma);
    const double den_l = 2.0 * ell * ell, den_c = 2.0 * P.c_ell * P.c_ell;
    const float sig2 = P.sigma * P.sigma, csig2 = P.c_sigma * P.c_sigma, il2 = 1 / (ell * ell);
    float4 none[PRE_T];
#pragma unroll
    for (int u = 0; u < PRE_T; ++u) none[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    float Ms[12];                                                                       // cvo::transform of the last executed iteration (cvo.cpp:815): the transforms below overwrite it
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) Ms[i] = sh->M[i];
    }
    // (the same decision on every workgroup of the pair: they exchange partial sums below.  A member in dense mode -- its rows' candidates did
    //  not fit its lists -- has no lists to walk and says so through the last reduction's spare slot)
    const unsigned long long tt0 = CVO_NOW();
    unsigned long long tt1 = tt0, tt2 = tt0, tt3 = tt0;
    const bool can = (y_lds == 1) && sh->list_valid && (G > 1 || sh->status == 0);
    const bool dense = sh->dense_mode != 0;
    const bool own = g == 0;                                                            // the pair's first workgroup writes the answers
    // ---- fip(T moving, fixed) and the Hessian terms from the candidate lists
    unsigned ep_pre = ep0 + 1u;                                                         // epoch of the last exchange below: consecutive exchanges alternate between the two buffers
    if (can) {
        transform_body_t<1>(c, L, sh, none, false);                                     // y = FINAL transform * p (cvo.cpp:485-487 with cvo.cpp:817); are the lists still valid for it?
        // The members of a pair do NOT all see the same answer: their lists may have been built under different margins (a member whose rows overflowed
        // the launch's margin rebuilt them with SKIN_DENSE_SCENE, the others did not) and at different iterations, so one member's lists can be stale for
        // the final transform while another's are not.  They exchange partial sums below, so they have to take the same branch: one more exchange, of the
        // "my lists are stale" flags, and everybody walks only when nobody's are.  (One workgroup per pair: its own flag.)
        bool lists_ok = sh->rebuild == 0;
        if (G > 1) {
            double fl[8] = {(tid == 0 && sh->rebuild != 0) ? 1.0 : 0.0, 0, 0, 0, 0, 0, 0, 0};
            const double stale = tail_reduce8(fl, sh, c, G, g, ep0, tid, nwaves);
            if (tid == 0) sh->cull_next = stale == 0.0 ? 1 : 0;                         // (a scratch word: idle outside a cull)
            __syncthreads();
            lists_ok = sh->cull_next != 0;
            __syncthreads();
        }
        if (lists_ok) {
            ep_pre = ep0 + 4u;
            double sumA = 0; int count = 0, hcount = 0; float H[21];
#pragma unroll
            for (int q = 0; q < 21; ++q) H[q] = 0.f;
            const bool x_lds = sh->x_lds != 0;
            const int nb = dense ? 0 : sh->wnb[wave];
            for (int bi = 0; bi < nb; ++bi) {
                const int blk = wave_block(bi, wave, nwaves);
                const int slot = blk * 64 + lane;
                const int len = L.lenS[slot];
                const int lw = uni((int)sh->blk_lmax[blk]);
                float xi[3]; load_x(c, L, x_lds, slot, xi);
                const int gi = global_row(c, (int)L.row_of[slot]);
                const float4 flo = ld4(c.fixed + lo_off(gi)), fhi = ld4(c.fixed + hi_off(c.nf, gi));
                const float fb[5] = {flo.w, fhi.x, fhi.y, fhi.z, fhi.w};
                const gv2u* ep = c.ent + 2 * slot;
                for (int n0 = 0; n0 < lw; n0 += PF) {
                    v2u en[PF];
#pragma unroll
                    for (int u = 0; u < PF; ++u) en[u] = ep[ent_ix(min(n0 + u, c.capn - 1), (size_t)c.rows_pad)];   // the step's entries in one round trip (stale beyond the row's end)
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        const bool act = n0 + u < len;
                        const int j = act ? (int)(en[u].y & 0xFFFFu) : 0;
                        const float4 y = L.ylds[j];
                        const float pa[3] = {y.x, y.y, y.z};
                        const float e0 = pa[0] - xi[0], e1 = pa[1] - xi[1], e2 = pa[2] - xi[2];
                        float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;       // nanoflann.hpp:403-406
                        if (act && d2 < d2_thres) {                                     // cvo.cpp:423 / 654
                            const float4 ghi = ld4(c.moving + hi_off(c.nm, j));
                            const float fa[5] = {y.w, ghi.x, ghi.y, ghi.z, ghi.w};
                            score_pair_terms(pa, fa, xi, fb, d2, d2c_thres, sig2, csig2, den_l, den_c, il2, sumA, count, H, hcount);
                        }
                    }
                }
            }
            double v[8];
            v[0] = sumA; v[1] = (double)count;
#pragma unroll
            for (int q = 0; q < 6; ++q) v[2 + q] = (double)H[q];
            double r0 = tail_reduce8(v, sh, c, G, g, ep0 + 1u, tid, nwaves);
            if (own && tid < 2) { out[24 + tid] = r0; out[4 * 24 + tid] = (tid == 1) ? r0 : 0.0; }
            if (own && tid >= 2 && tid < 8) out[4 * 24 + tid] = r0;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (double)H[6 + q];
            r0 = tail_reduce8(v, sh, c, G, g, ep0 + 2u, tid, nwaves);
            if (own && tid < 8) out[4 * 24 + 8 + tid] = r0;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = q < 7 ? (double)H[14 + q] : ((dense && tid == 0) ? 1.0 : 0.0);
            r0 = tail_reduce8(v, sh, c, G, g, ep0 + 3u, tid, nwaves);
            if (own && tid < 7) out[4 * 24 + 16 + tid] = r0;
            const double no_lists = __shfl(r0, 7, 64);                                  // (wave 0 holds the totals; thread 0 decides)
            if (tid == 0 && no_lists == 0.0) answered |= TAIL_POST | TAIL_HESSIAN;
        }
        tt1 = CVO_NOW();
        // ---- fip(moving, fixed): the untransformed cloud against the fixed one, one cull at this radius
        float Rs[9], Ts[3];
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) { Rs[i] = sh->R[i]; sh->R[i] = (i % 4 == 0) ? 1.f : 0.f; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { Ts[i] = sh->T[i]; sh->T[i] = 0.f; }
            sh->list_valid = 0;
            sh->P.skin = 0.f; sh->P.skin_alpha = 0.f; sh->twist_ok = 0;                                   // nothing moves any more: the cull's radius is r_c itself (restored below)
        }
        __syncthreads();
        transform_body_t<1>(c, L, sh, none, false);                                     // y = p
        const int rebuilds_before = sh->rebuilds;
        phase_cull(Dp, g, G, tgeo, y_lds);                                              // the rows' neighbours within (1 + skin) r_c of the untransformed cloud, by row
        tt2 = CVO_NOW();
        {
            double sumA = 0; int count = 0, hcount = 0; float H[21];
#pragma unroll
            for (int q = 0; q < 21; ++q) H[q] = 0.f;
            int overflow = 0;
            for (int li = tid; li < c.nrows; li += nthreads) {
                const int len = L.rowlen[li];
                if (len > c.capn) { overflow = 1; continue; }
                const int gi = global_row(c, li);
                const float4 flo = ld4(c.fixed + lo_off(gi)), fhi = ld4(c.fixed + hi_off(c.nf, gi));
                const float xi[3] = {flo.x, flo.y, flo.z};
                const float fb[5] = {flo.w, fhi.x, fhi.y, fhi.z, fhi.w};
                const gv2u* jp = c.jT4 + li;
                for (int n = 0; n < len; ++n) {
                    const v2u w = jp[(size_t)(n >> 2) * c.rows_pad];
                    const unsigned half = (n & 2) ? w.y : w.x;
                    const int j = (int)((n & 1) ? (half >> 16) : (half & 0xFFFFu));
                    const float4 y = L.ylds[j];
                    const float pa[3] = {y.x, y.y, y.z};
                    const float e0 = pa[0] - xi[0], e1 = pa[1] - xi[1], e2 = pa[2] - xi[2];
                    float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;
                    if (d2 < d2_thres) {
                        const float4 ghi = ld4(c.moving + hi_off(c.nm, j));
                        const float fa[5] = {y.w, ghi.x, ghi.y, ghi.z, ghi.w};
                        score_pair_terms(pa, fa, xi, fb, d2, d2c_thres, sig2, csig2, den_l, den_c, il2, sumA, count, H, hcount);
                    }
                }
            }
            double v[8] = {sumA, (double)count, (double)overflow, 0, 0, 0, 0, 0};
            const double r0 = tail_reduce8(v, sh, c, G, g, ep_pre, tid, nwaves);       // (with or without the lists of the first part: the members all took the same branch)
            if (own && tid < 2) out[tid] = r0;
            const double ovf = __shfl(r0, 2, 64);                                       // lanes 0..7 of wave 0 hold the totals; thread 0 decides
            if (tid == 0 && ovf == 0.0) answered |= TAIL_PRE;
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) sh->R[i] = Rs[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) sh->T[i] = Ts[i];
            sh->rebuilds = rebuilds_before;
            sh->P.skin = P.skin; sh->P.skin_alpha = P.skin_alpha;
        }
        __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) sh->M[i] = Ms[i];
    }
    // ---- fip(fixed, fixed), fip(moving, moving) from the clouds' tables
    if (tid == 0 && own) {
        const SelfCacheEntry* tabs[2] = {Dp->self_fixed, Dp->self_moving};
        for (int q = 0; q < 2; ++q) {
            if (!tabs[q]) continue;
            for (int e = 0; e < SELF_CACHE_N; ++e) {
                if (tabs[q][e].valid && tabs[q][e].ell == ell) { out[(2 + q) * 24] = tabs[q][e].sum; out[(2 + q) * 24 + 1] = tabs[q][e].count; answered |= (q == 0 ? TAIL_FIXED : TAIL_MOVING); }
            }
        }
        out[23] = (double)answered;
    }
    tt3 = CVO_NOW();
    if (tid == 0) { sh->tail_ticks[0] = tt1 - tt0; sh->tail_ticks[1] = tt2 - tt1; sh->tail_ticks[2] = tt3 - tt2; sh->tail_ticks[3] = tt3 - tt0; }
    __syncthreads();
}

// ---- Adoption: finished workgroups help with the pairs that are still running (one workgroup per pair, a slot per pair).
// Alignments take 33 ... 150 iterations: when a job runs out of queued work the last pairs drag on with most CUs idle.  A workgroup
// that has finished its pair, and finds nothing queued on the device, offers itself to a pair of its launch that is still running:
//   queue[1 + slot] = {launch tag | state, payload}: FREE (the pair runs alone) -> REQUEST (payload = helper's block; helper's CAS)
//   -> ACCEPT (payload = iteration of the join; owner's CAS, after it has written the pair's state to PairDesc::state) or back to FREE
//   (helper's CAS after a timeout); CLOSED when the pair ends (owner).  Both CAS on the same word: either both agree or neither.
// From the join on the pair runs as G = 2: rows dealt anew (pair_rows), lists rebuilt, partial sums exchanged -- the path G > 1
// launches always take.  A pair's results do not depend on G beyond the order of the double-precision partial sums (section 4.1).

// One pair from its start state to the state written back: everything a workgroup does between taking a pair up (its own slot's, a pull from the pair queue,
// a pair it has offered to help with) and going back for the next.  A function of its own so that the kernel's pull / adoption loop and the iteration loop do not
// share one register allocation: together they kept more than a hundred uniform values alive across every phase call, spilled to lanes of vector registers that
// were themselves spilled to scratch around the calls (the three-waves-per-SIMD build: 158 scratch accesses per iteration in the loop; -Rpass-analysis in DESIGN.md).
static __device__ __noinline__ void run_pair(const PairDesc* descs_in, int ps_in, int slot_in, int slots_in, int ge_in, int Ge_in, unsigned k_join_in, unsigned launch_tag_in,
                                            int tgeo_in, int y_lds_in, unsigned long long* queue_in, const float* const* raw_table_in, int adopt_launch_in) {
    const PairDesc* descs = uni_ptr(descs_in); const float* const* raw_table = uni_ptr(raw_table_in); gu64* queue = (gu64*)uni_ptr(queue_in);
    const int slot = uni(slot_in), slots = uni(slots_in), tgeo = uni(tgeo_in), y_lds = uni(y_lds_in);
    int ge = uni(ge_in), Ge = uni(Ge_in);
    const unsigned k_join = (unsigned)uni((int)k_join_in), launch_tag = (unsigned)uni((int)launch_tag_in);
    const bool adopt_launch = uni(adopt_launch_in) != 0;
    Shared* sh = reinterpret_cast<Shared*>(cvo_smem);
    const int tid = threadIdx.x;
    const int max_iter = sh->P.max_iter;
    int p = uni(ps_in);
    // `p` so far is a position of the launch (a slot, or a pull from the queue): the pair it stands for is the host's choice (densest clouds first)
    const int ps = p;
    p = descs[ps].run_pair;
    const PairDesc* Dp = descs + p;
    const int nf = Dp->nf, nm = Dp->nm;
    if (raw_table && !k_join) {
        // The pair's clouds as the caller handed them over (cvo_batch_set_pair(s): n x 3 positions AoS, data_type.h:30, then 5 channel-major
        // feature arrays, data_type.h:75), still in the host's pinned staging ring: this workgroup builds the two float4 planes itself --
        // the points cross PCIe here, once.  (A helper that joins later sees the planes behind the owner's release, like the pair's state.)
        const float* const raws[2] = {raw_table[4 * p], raw_table[4 * p + 2]};          // {positions, features (null: right behind the positions)} of the fixed, then of the moving cloud
        const float* const rawf[2] = {raw_table[4 * p + 1], raw_table[4 * p + 3]};
        float* const dsts[2] = {const_cast<float*>(Dp->fixed), const_cast<float*>(Dp->moving)};
        const int ns[2] = {nf, nm};
        // (Staging the positions through LDS -- one coalesced stream over PCIe instead of three strided 4-byte loads per point -- was measured and is 1 % slower:
        //  profiles/r05_upload_ab.txt.)
        for (int q = 0; q < 2; ++q) {
            const float* xyz = raws[q]; if (!xyz) continue;
            const int n = ns[q]; const float* feat = rawf[q] ? rawf[q] : xyz + 3 * (size_t)n;
            for (int i = tid; i < n; i += blockDim.x) {
                float4 lo, hi;
                lo.x = xyz[3 * (size_t)i]; lo.y = xyz[3 * (size_t)i + 1]; lo.z = xyz[3 * (size_t)i + 2]; lo.w = feat[i];
                hi.x = feat[(size_t)n + i]; hi.y = feat[2 * (size_t)n + i]; hi.z = feat[3 * (size_t)n + i]; hi.w = feat[4 * (size_t)n + i];
                *reinterpret_cast<float4*>(dsts[q] + lo_off(i)) = lo;
                *reinterpret_cast<float4*>(dsts[q] + hi_off(n, i)) = hi;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const int rows_per = ((((nf + ROW_DEAL - 1) / ROW_DEAL) + Ge - 1) / Ge) * ROW_DEAL;
    const PairState* st_from = k_join ? (const PairState*)Dp->state : Dp->state_in;   // a helper starts from what the pair's owner published
    if (tid < 25) {                                               // R[9], T[3], ell, transform[12]: the head of PairState, one lane per word
        const float v = __uint_as_float(__hip_atomic_load((const CVO_GLOBAL unsigned*)st_from + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (tid < 9) sh->R[tid] = v; else if (tid < 12) sh->T[tid - 9] = v; else if (tid == 12) sh->ell = v; else sh->M[tid - 13] = v;
    }
    if (tid == 32) {
        const PairState* st = st_from;
        int rp, nr; pair_rows(nf, ge, Ge, rp, nr); sh->ctx_rows_per = rp; sh->ctx_nrows = nr;
        sh->ws_slot = k_join ? ps : slot;                         // (one slot per pair when workgroups help each other)
        store_ctx(Dp, ge, Ge);
        sh->adopt_req = 0; sh->adopt_word = nullptr; sh->joined_at = 0; sh->retracted = 0;
        if (adopt_launch && !k_join) {                            // this pair may be helped: its word says so from now on
            __hip_atomic_store(&queue[1 + slot], (unsigned long long)(launch_tag | ADOPT_FREE) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh->adopt_word = (unsigned long long*)&queue[1 + slot];
        }
        sh->stop = 0; sh->status = 0; sh->nnz = 0; sh->cand = 0;
        sh->iter_at_break = k_join ? (int)__hip_atomic_load((const CVO_GLOBAL unsigned*)&st->iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : st->iter;
        for (int i = 0; i < 4; ++i) sh->sub[i] = 0;
        for (int i = 0; i < 10; ++i) sh->ticks[i] = 0;
        sh->cand_total = 0; sh->nnz_total = 0; sh->cull_mask = 0ull; sh->predict_mask = 0ull; for (int i = 0; i < 4; ++i) sh->tail_ticks[i] = 0ull;
        sh->P.skin = sh->skin0; sh->P.skin_alpha = sh->alpha0; sh->alpha_build = 0.f; sh->reach = 0.f; sh->xmax = 0.f; sh->twist_ok = 0; sh->predicted = 0; sh->resort_pending = 0; sh->fused = 0; sh->reach_now = 0.f; sh->list_valid = 0; sh->dense_mode = 0; sh->total = 0; sh->rebuilds = 0; sh->refines = 0; sh->dense_fallbacks = 0; sh->Rb = 0.f; sh->ell_build = -1.f;
    }
    __syncthreads();
    // a helper has read what the owner published for it: if the pair may grow further, its word takes offers again (not earlier --
    // the owner publishes the next newcomer's state in the same place)
    if (k_join && Ge < ADOPT_GMAX && tid == 0)
        __hip_atomic_store(&queue[1 + ps], ((unsigned long long)(launch_tag | ADOPT_FREE) << 32) | k_join, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    int k = (int)k_join;
    // phase timers and counters live in LDS, bumped by thread 0 with fire-and-forget ds_add: as registers of this function they were
    // saved and restored around every phase call (the phases are out of line), ~1 us of lane moves per iteration
    unsigned long long t_prev = CVO_NOW();
#ifdef CVO_KTRACE
    unsigned long long kt_prev[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ksub_prev[4] = {0, 0, 0, 0};
#endif
    const unsigned long long clk_t0 = t_prev, clk_c0 = __builtin_amdgcn_s_memtime();
#define CVO_PHASE(idx) do { const unsigned long long t_now = CVO_NOW(); if (tid == 0) atomicAdd(&sh->ticks[idx], t_now - t_prev); t_prev = t_now; } while (0)
    const bool ok_pair = (nf > 0 && nm > 0 && rows_per <= MAX_ROWS_PER_WG);
    if (!ok_pair && tid == 0) sh->status = (nf > 0 && nm > 0) ? 4 : 2;   // CVO_ERR_INVALID / CVO_ERR_EMPTY_CLOUD (reference: assert / UB, Q8)

    if (ok_pair && k < max_iter) phase_transform(Dp, ge, Ge, tgeo, y_lds);   // later iterations: done by the epilogue before them
    for (; ok_pair && k < max_iter; ++k) {
        if (sh->rebuild == 1) {
            const unsigned long long t_a = CVO_NOW();
            phase_cull(Dp, ge, Ge, tgeo, y_lds);
            const unsigned long long t_b = CVO_NOW();
            phase_sort(Dp, ge, Ge, tgeo, y_lds);
            if (sh->dense_mode && (sh->P.skin > SKIN_DENSE_SCENE || sh->P.skin_alpha > 0.f)) {   // the lists of this margin do not fit (a surface a few decimetres from the camera;
                __syncthreads();                                     // clouds far from the origin, whose depth-proportional margin is metres): once more with a narrow constant
                                                                     // margin, kept for the rest of the pair, before the rows fall back to dense sweeps
                if (tid == 0) { sh->P.skin = fminf(sh->P.skin, SKIN_DENSE_SCENE); sh->P.skin_alpha = 0.f; sh->dense_fallbacks -= 1; }
                __syncthreads();
                phase_cull(Dp, ge, Ge, tgeo, y_lds);
                phase_sort(Dp, ge, Ge, tgeo, y_lds);
            }
            if (tid == 0) { atomicAdd(&sh->ticks[6], t_b - t_a); atomicAdd(&sh->ticks[8], CVO_NOW() - t_b); sh->cull_mask |= 1ull << min(k, 63); if (sh->predicted) { sh->predict_mask |= 1ull << min(k, 63); sh->predicted = 0; } }
        } else if (sh->rebuild == 2) {
            phase_refine(Dp, ge, Ge, tgeo, y_lds, k);
            if (tid == 0 && sh->predicted) { sh->predict_mask |= 1ull << min(k, 63); sh->predicted = 0; }
        }
        if (sh->resort_pending) {
            if (sh->rebuild == 0) phase_resort(Dp, ge, Ge, tgeo, y_lds);
            else { __syncthreads(); if (tid == 0) sh->resort_pending = 0; __syncthreads(); }   // (the lists have been rebuilt or filtered again meanwhile)
        }
        CVO_PHASE(0);
#if CVO_INLINE_PHASES & 8
        phase_iteration(Dp, ge, Ge, tgeo, y_lds, k, max_iter);
        t_prev = CVO_NOW();
        if (sh->status != 0) break;
#else
        phase_candidates(Dp, ge, Ge, tgeo, y_lds, k);
        if (tid == 0) { atomicAdd(&sh->cand_total, (unsigned long long)sh->cand); atomicAdd(&sh->nnz_total, (unsigned long long)sh->nnz); }
        CVO_PHASE(1);
        if (sh->status != 0) break;
        if (ls_even_shares(y_lds)) phase_linesearch_even(Dp, ge, Ge, tgeo, y_lds, k); else phase_linesearch(Dp, ge, Ge, tgeo, y_lds, k);
        CVO_PHASE(3);
        if (sh->status != 0) break;
        phase_epilogue(Dp, ge, Ge, tgeo, y_lds, k, max_iter);
        CVO_PHASE(5);
#endif
#ifdef CVO_KTRACE   // experiment builds only: the trace row's B..E carry this iteration's phase times (100 MHz ticks) instead
        if (tid == 0 && ge == 0 && Dp->trace && k < Dp->trace_cap) {
            TraceRow& tr = Dp->trace[k];
            unsigned long long ticks[10];
            for (int q = 0; q < 10; ++q) ticks[q] = sh->ticks[q];
            tr.B = (double)(ticks[0] - kt_prev[0]); tr.C = (double)(ticks[1] - kt_prev[1]); tr.D = (double)(ticks[3] - kt_prev[3]); tr.E = (double)(ticks[5] - kt_prev[5]);
            // omega = candidate phase (prologue, row walk, wait + reduction), v = line-search walk, its reduction, epilogue scalar part; step = fused transform
            tr.omega[0] = (float)(sh->sub[0] - ksub_prev[0]); tr.omega[1] = (float)(sh->sub[1] - ksub_prev[1]); tr.omega[2] = (float)(sh->sub[2] - ksub_prev[2]);
            tr.v[0] = (float)sh->ksub[0]; tr.v[1] = (float)sh->ksub[1]; tr.v[2] = (float)sh->ksub[2]; tr.step = (float)sh->ksub[3];
#ifdef CVO_KTRACE_CULL
            tr.omega[0] = (float)(sh->kabs[1] - sh->kabs[0]); tr.omega[1] = (float)(sh->kabs[2] - sh->kabs[1]); tr.omega[2] = (float)(sh->kabs[3] - sh->kabs[2]); tr.v[0] = (float)(CVO_NOW() - sh->kabs[3]);
#endif
#ifdef CVO_KTRACE_WAVES
            {   // omega = min / mean / max over the waves of the walk's end, v[0] = wave 0's, v[1] = the latest wave's index
                unsigned long long mn = ~0ull, mx = 0, sm = 0; int wmx = 0; const int nw = (int)blockDim.x >> 6;
                for (int w = 0; w < nw; ++w) { const unsigned long long t = sh->kabs[w]; mn = t < mn ? t : mn; if (t > mx) { mx = t; wmx = w; } sm += t; }
                tr.omega[0] = (float)mn; tr.omega[1] = (float)(sm / (unsigned long long)nw); tr.omega[2] = (float)mx; tr.v[0] = (float)sh->kabs[0]; tr.v[1] = (float)wmx;
            }
#endif
#if defined(CVO_KTRACE_EPI) && CVO_KTRACE_EPI == 2
            tr.omega[0] = (float)sh->kabs[6]; tr.omega[1] = (float)sh->kabs[7]; tr.omega[2] = (float)sh->kabs[4]; tr.ell = (float)sh->kabs[10]; tr.dist = (float)sh->kabs[11];
#endif
            for (int q = 0; q < 4; ++q) ksub_prev[q] = sh->sub[q];
            for (int q = 0; q < 10; ++q) kt_prev[q] = ticks[q];
        }
#endif
        if (sh->stop) { ++k; break; }
        if (adopt_launch && (Ge > 1 || sh->adopt_req) && k + 1 < max_iter) {
            // Members of a pair that has (or is about to get) helpers agree on the member count of the next iteration: the owner
            // decides -- it accepts an offer it saw in its epilogue, if the pair may still grow -- and writes {iteration, members}
            // into the pair's control word; the helpers wait for that word.  On a change every member deals its rows anew
            // (pair_rows) and the lists are rebuilt; the newcomer starts from the state the owner published (the head of
            // PairState: R, T, ell, the current transform; iter) as member `old count`.
            if (tid == 0) {
                int g_next = Ge;
                gu64* ctrl = &queue[1 + slots + (ge == 0 ? slot : (int)sh->ws_slot)];
                if (ge == 0) {
                    if (sh->adopt_req && Ge < ADOPT_GMAX) {
                        CVO_GLOBAL unsigned* pub = (CVO_GLOBAL unsigned*)Dp->state;
                        for (int i = 0; i < 9; ++i) pub[i] = __float_as_uint(sh->R[i]);
                        for (int i = 0; i < 3; ++i) pub[9 + i] = __float_as_uint(sh->T[i]);
                        pub[12] = __float_as_uint(sh->ell);
                        for (int i = 0; i < 12; ++i) pub[13 + i] = __float_as_uint(sh->M[i]);
                        ((CVO_GLOBAL PairState*)Dp->state)->iter = sh->iter_at_break;
                        unsigned long long e = ((unsigned long long)(launch_tag | ADOPT_REQUEST) << 32) | (unsigned)(sh->adopt_req - 1);
                        const unsigned long long acc = ((unsigned long long)(launch_tag | ADOPT_ACCEPT) << 32) | ((unsigned)(Ge + 1) << 24) | ((unsigned)Ge << 16) | (unsigned)((k + 1) & 0xFFFF);
                        if (__hip_atomic_compare_exchange_strong((gu64*)sh->adopt_word, &e, acc, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            // The helper polls this word and confirms within a microsecond or two (ACCEPT -> CONFIRMED, its CAS).  Only then does the pair count on
                            // it: should no confirmation come (the helper is gone), the owner takes the acceptance back (ACCEPT -> FREE, its CAS -- one of the two
                            // wins) and the pair carries on with the members it has, instead of waiting for a member that never sends its partial sums.
                            bool joined = false;
                            const unsigned long long t_acc = __builtin_amdgcn_s_memrealtime();
                            for (;;) {
                                const unsigned long long x = __hip_atomic_load((gu64*)sh->adopt_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if ((unsigned)(x >> 32) != (launch_tag | ADOPT_ACCEPT)) { joined = true; break; }
                                if (__builtin_amdgcn_s_memrealtime() - t_acc > ADOPT_CONFIRM_TICKS) {
                                    unsigned long long e2 = acc;
                                    joined = !__hip_atomic_compare_exchange_strong((gu64*)sh->adopt_word, &e2, ((unsigned long long)(launch_tag | ADOPT_FREE) << 32) | (unsigned)(k + 1),
                                                                                    __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(1);
                            }
                            if (joined) {
                                g_next = Ge + 1;
                                if (!sh->joined_at) sh->joined_at = k + 1;
                                if (g_next >= ADOPT_GMAX) sh->adopt_word = nullptr;     // full: no more offers are looked at (the newcomer leaves the word as it is)
                            } else sh->retracted += 1;
                        }
                    }
                    if (g_next > 1) __hip_atomic_store(ctrl, ((unsigned long long)(launch_tag | (unsigned)((k + 1) & 0xFFFF)) << 32) | (unsigned)g_next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        const unsigned long long x = __hip_atomic_load(ctrl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(x >> 32) == (launch_tag | (unsigned)((k + 1) & 0xFFFF))) { g_next = (int)(unsigned)x; break; }
                        if (__builtin_amdgcn_s_memrealtime() - t_start > 300000000ull) { sh->status = 6; break; }   // 3 s: the owner is gone
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                sh->adopt_req = g_next;
                if (g_next != Ge) {
                    int rp, nr; pair_rows(nf, ge, g_next, rp, nr); sh->ctx_rows_per = rp; sh->ctx_nrows = nr;
                    store_ctx(Dp, ge, g_next);
                    sh->list_valid = 0; sh->rebuild = 1; sh->dense_mode = 0;
                }
            }
            __syncthreads();
            Ge = sh->adopt_req;
            __syncthreads();
            if (sh->status != 0) break;
        }
    }

    // ---- after the loop: the tracker's score block for this pair, when asked for (one workgroup per pair; a helped pair is left to the host)
    __syncthreads();
    if (Dp->score_out && ok_pair) phase_tail_scores(Dp, ge, Ge, tgeo, y_lds, k);   // every workgroup of the pair: each holds the lists of its own rows
    // ---- (cvo.cpp:815-817): write the pair's state back
    __syncthreads();
    if (tid == 0 && ge == 0) {
        if (adopt_launch) __hip_atomic_store(&queue[1 + slot], (unsigned long long)(launch_tag | ADOPT_CLOSED) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody joins any more
        PairState fin;
        float R[9], T[3], M[12];
        for (int i = 0; i < 9; ++i) { R[i] = sh->R[i]; fin.R[i] = R[i]; }
        for (int i = 0; i < 3; ++i) { T[i] = sh->T[i]; fin.T[i] = T[i]; }
        make_transform(R, T, M);                                                            // final update_tf, cvo.cpp:817
        for (int i = 0; i < 12; ++i) { fin.prev_transform[i] = sh->M[i]; fin.transform[i] = M[i]; }
        fin.ell = sh->ell;
        fin.iter = sh->iter_at_break;                                                       // unchanged (stale) if no break: Q4
        fin.A_nonzero = sh->nnz;
        fin.iterations_run = k;
        fin.status = sh->status;
        fin.rebuilds = sh->rebuilds;
        fin.joined_at = sh->joined_at; fin.adopt_retracted = sh->retracted;
        fin.dense_fallbacks = sh->dense_fallbacks;
        fin.candidates_total = (long long)sh->cand_total; fin.nonzeros_total = (long long)sh->nnz_total;
        for (int i = 0; i < 10; ++i) fin.phase_ticks[i] = sh->ticks[i];
        fin.phase_ticks[7] = sh->sub[0]; fin.phase_ticks[9] = sh->sub[1]; fin.phase_ticks[2] = sh->sub[2]; fin.phase_ticks[4] = sh->sub[3];
        fin.clk_cycles = __builtin_amdgcn_s_memtime() - clk_c0; fin.clk_ticks = __builtin_amdgcn_s_memrealtime() - clk_t0; fin.clk_t0 = clk_t0; fin.cull_mask = sh->cull_mask; fin.predict_mask = sh->predict_mask; for (int i = 0; i < 4; ++i) fin.tail_ticks[i] = sh->tail_ticks[i];
        *Dp->state = fin;                                          // device copy: the next launch may start from it
        *Dp->state_host = fin;                                     // pinned host mirror: visible to the host when the kernel has completed
        if (Dp->record) {                                          // the pair's 64-byte record of the cross-GPU gather (ints as floats: exact below 2^24)
            gv4f* rec = (gv4f*)Dp->record;
            v4f r0, r1, r2, r3;
            r0.x = M[0]; r0.y = M[1]; r0.z = M[2]; r0.w = M[3]; r1.x = M[4]; r1.y = M[5]; r1.z = M[6]; r1.w = M[7]; r2.x = M[8]; r2.y = M[9]; r2.z = M[10]; r2.w = M[11];
            r3.x = (float)fin.iter; r3.y = (float)fin.A_nonzero; r3.z = (float)fin.iterations_run; r3.w = (float)fin.status;
            rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3;
        }
    }
    __syncthreads();
}


// A workgroup whose pair is done looks for a pair of its launch that still runs alone and offers to help (cvo_align_kernel, adoption): wave 0 -- a lane per
// slot looks, lane 0 asks.  Leaves the slot found (or -1) in sh->cand and the owner's answer in sh->adopt_k.  (A function of its own: the kernel's loop around
// run_pair then keeps nothing in vector registers across its calls.)
static __device__ __noinline__ void adopt_search(gu64* queue, const unsigned* wgs_submitted, unsigned* wgs_started, int slots_in, int slot_in, unsigned launch_tag_in) {
    Shared* sh = reinterpret_cast<Shared*>(cvo_smem);
    const int tid = threadIdx.x, slots = uni(slots_in), slot = uni(slot_in);
    const unsigned launch_tag = (unsigned)uni((int)launch_tag_in);
    if (tid < 64) {                                           // wave 0: a lane per slot looks, lane 0 asks
        int found = -1; unsigned kj = 0;
        unsigned sub = __hip_atomic_load(wgs_submitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned sta = __hip_atomic_load(wgs_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sub == sta && sh->P.adopt_dwell > 0) {                // dry right now: is it the end of the job, or the moment between a completion and the caller's next launch?
            const unsigned long long t_dry = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t_dry < (unsigned long long)sh->P.adopt_dwell) {
                __builtin_amdgcn_s_sleep(64);
                sub = __hip_atomic_load(wgs_submitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                sta = __hip_atomic_load(wgs_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sub != sta) break;                        // new work has been queued: leave, its workgroups want this CU
            }
        }
        for (int attempt = 0; attempt < 4 && found < 0 && sub == sta; ++attempt) {
            // the pair that runs alone and has the most left to do, as far as one can tell: the one with the fewest iterations behind it
            unsigned key = 0xFFFFFFFFu;                     // iteration << 12 | slot
            for (int s2 = tid; s2 < slots; s2 += 64) {
                if (s2 == slot) continue;
                const unsigned long long w = __hip_atomic_load(&queue[1 + s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(w >> 32) == (launch_tag | ADOPT_FREE)) key = min(key, (min((unsigned)w, 0xFFFFFu) << 12) | (unsigned)s2);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, off, 64));
            if (key == 0xFFFFFFFFu) break;                  // nobody runs alone any more
            if ((key >> 12) >= (unsigned)sh->P.adopt_kmax) break; // every pair that runs alone is in its light late iterations: a second workgroup would cost it a list rebuild and gain it next to nothing
            const int s2 = (int)(key & 0xFFFu);
            int got = 0;                                    // lane 0: 1 accepted, 0 try again, -1 (unused)
            if (tid == 0) {
                unsigned long long w = __hip_atomic_load(&queue[1 + s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long want = ((unsigned long long)(launch_tag | ADOPT_REQUEST) << 32) | (unsigned)blockIdx.x;
                if ((unsigned)(w >> 32) == (launch_tag | ADOPT_FREE) &&
                    __hip_atomic_compare_exchange_strong(&queue[1 + s2], &w, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                    for (;;) {                              // the owner answers in its next epilogue
                        const unsigned long long x = __hip_atomic_load(&queue[1 + s2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned st = (unsigned)(x >> 32) - launch_tag;
                        if (st == ADOPT_ACCEPT) {              // payload: members after the join << 24 | this helper's index << 16 | iteration of the join
                            // confirm at once (a CAS: the owner takes the acceptance back when no confirmation comes, and only one of the two can win)
                            unsigned long long e = x;
                            if (sh->P.adopt_inject != 1 &&
                                __hip_atomic_compare_exchange_strong(&queue[1 + s2], &e, ((unsigned long long)(launch_tag | ADOPT_CONFIRMED) << 32) | (unsigned)x,
                                                                     __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { got = 1; kj = (unsigned)x; }
                            break;
                        }
                        if (st != ADOPT_REQUEST) break;     // the pair ended meanwhile
                        if (__builtin_amdgcn_s_memrealtime() - t_start > 200000ull) {   // 2 ms: take the offer back -- unless it has just been accepted
                            unsigned long long e = want;
                            const unsigned long long free_w = (unsigned long long)(launch_tag | ADOPT_FREE) << 32;
                            if (__hip_atomic_compare_exchange_strong(&queue[1 + s2], &e, free_w, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
            }
            got = __builtin_amdgcn_readfirstlane(got);
            if (got == 1) found = s2;
        }
        if (tid == 0) { sh->cand = found; sh->adopt_k = kj; }
    }
}

// The next pair of a launch with fewer slots than pairs (cvo_align_kernel, the in-kernel queue): the slot's first workgroup takes the next index and passes it on
// to the slot's other workgroups; leaves it in sh->cand.
static __device__ __noinline__ void queue_pull(gu64* queue, int slot_in, int g_in, int G_in, unsigned launch_tag_in, unsigned pull_in) {
    Shared* sh = reinterpret_cast<Shared*>(cvo_smem);
    const int tid = threadIdx.x, slot = uni(slot_in), g = uni(g_in), G = uni(G_in);
    const unsigned launch_tag = (unsigned)uni((int)launch_tag_in), pull = (unsigned)uni((int)pull_in);
    if (tid == 0) {
        unsigned long long got = 0;
        const unsigned long long seq = (unsigned long long)(launch_tag | (pull + 1u)) << 32;
        if (g == 0) {
            unsigned long long old = __hip_atomic_load(&queue[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), nw;
            do { nw = ((unsigned)(old >> 32) == launch_tag) ? old + 1ull : (((unsigned long long)launch_tag << 32) | 1ull); }
            while (!__hip_atomic_compare_exchange_strong(&queue[0], &old, nw, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            got = ((unsigned)(old >> 32) == launch_tag) ? (old & 0xFFFFFFFFull) : 0ull;
            if (G > 1) __hip_atomic_store(&queue[1 + slot], seq | got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 8 bytes: the data is its own flag
        } else {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned long long x = __hip_atomic_load(&queue[1 + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((x >> 32) == (seq >> 32)) { got = x & 0xFFFFFFFFull; break; }
                if (__builtin_amdgcn_s_memrealtime() - t_start > 300000000ull) { got = 0xFFFFFFFFull; break; }   // 3 s: give up, leave
                __builtin_amdgcn_s_sleep(2);
            }
        }
        sh->cand = (int)min(got, (unsigned long long)0x7FFFFFFF);   // (re-initialised below for the pair)
    }
}

__global__ __launch_bounds__(BLOCK_MAX, BLOCK_MAX > 512 ? 1 : CVO_WAVES_PER_SIMD) void cvo_align_kernel(const PairDesc* __restrict__ descs, int n_pairs, int G, int tile, int y_lds, int rows_cap, int y_cap,
                                                                         unsigned launch_tag, int tab_cols, unsigned long long* __restrict__ queue_in, DevParams P,
                                                                         const unsigned* wgs_submitted /* host-mapped */, unsigned* wgs_started,
                                                                         const float* const* __restrict__ raw_table /* pinned host memory, or null */) {
    Shared* sh = reinterpret_cast<Shared*>(cvo_smem);
    const int tid = threadIdx.x;
    const int slots = gridDim.x / G;
    int slot = blockIdx.x / G, g = blockIdx.x % G;
    if (P.colocate && G > 1 && (slots & 7) == 0) {                  // members of a slot at blocks congruent mod 8: one XCD, one L2 (DevParams::colocate)
        const int t = blockIdx.x >> 3;
        g = t % G; slot = (blockIdx.x & 7) + 8 * (t / G);
    }
    // "is anything queued on the device?" (adoption): every workgroup the library submits counts itself as started, whatever kind of launch it belongs to
    if (wgs_started != nullptr && tid == 0) atomicAdd(wgs_started, 1u);
    if (slot >= slots) return;                                      // gridDim.x is a multiple of G; defensive
    if (tid == 0) { sh->rc_ell = -1.f; sh->P = P; sh->skin0 = P.skin; sh->alpha0 = P.skin_alpha; sh->launch_tag = launch_tag; sh->rows_cap = rows_cap; sh->y_cap = y_cap; sh->tab_cols = tab_cols; }
    __syncthreads();                                                // (run_pair reads the parameters from there)
    const bool adopting = wgs_started != nullptr && P.adopt_on != 0;   // set by the host for launches of one workgroup and one slot per pair
    const int tgeo = pack_geometry(tile, rows_cap, y_cap);
    gu64* queue = (gu64*)queue_in;

    // A launch with fewer pair slots than pairs (large clouds: G workgroups per pair, a share of the device per launch) hands the pairs out
    // dynamically -- alignments have data-dependent iteration counts (33 ... 150), a static deal would leave slots idle behind the longest
    // pair.  queue[0] = {launch tag, next pair}: the slot's first workgroup takes the next index (a CAS loop: the tag makes a stale word of
    // an earlier launch start from 0, so nothing has to be cleared between launches) and passes it to the slot's other workgroups through
    // queue[1 + slot] = {launch tag | pull number, pair}.  With a slot per pair there is nothing to hand out.
    const bool dynamic = slots < n_pairs;
    const bool adopt_launch = adopting && !dynamic && G == 1;
    for (unsigned pull = 0;; ++pull) {
        int p;
        int ge = g, Ge = G;                                             // this workgroup's place in the pair it works on
        unsigned k_join = 0;                                            // > 0: it joins a running pair of another slot at that iteration
        if (!dynamic && pull == 0) p = slot;
        else if (!dynamic) {
            if (!adopt_launch) break;
            // this workgroup's pair is done.  With nothing queued on the device (every workgroup submitted so far has started), look for
            // a pair of the launch that still runs alone and offer to help; leave when there is none.
            adopt_search(queue, wgs_submitted, wgs_started, slots, slot, launch_tag);
            __syncthreads();
            p = sh->cand;
            { const unsigned kj = sh->adopt_k; k_join = kj & 0xFFFFu; ge = (int)((kj >> 16) & 0xFFu); Ge = (int)(kj >> 24); }
            __syncthreads();
            if (p < 0) break;
        } else {
            queue_pull(queue, slot, g, G, launch_tag, pull);
            __syncthreads();
            p = sh->cand;
            __syncthreads();
            if (p >= n_pairs) break;
        }
        run_pair(descs, p, slot, slots, ge, Ge, k_join, launch_tag, tgeo, y_lds, queue_in, raw_table, adopt_launch ? 1 : 0);
    }
}

// The 64-byte result records of a launch ({transform[12], iter, A_nonzero, iterations_run, status} as floats) are written by the align
// kernel itself (PairDesc::record).  What is left for the gather path: records that stand for no pair -- the padding of a rank whose
// block is shorter than the longest one (cvo_shard_range), or all of a rank's records when its launch could not be made -- carry a
// status and zeros; and a copy for callers that want the records in a buffer of their own.
__global__ void cvo_fill_records_kernel(float* __restrict__ rec, int from, int to, float status) {
    const int p = from + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= to) return;
    for (int i = 0; i < 15; ++i) rec[p * 16 + i] = 0.f;
    rec[p * 16 + 15] = status;
}
hipError_t launch_fill_records(float* rec, int from, int to, int status, hipStream_t stream) {
    if (to > from) hipLaunchKernelGGL(cvo_fill_records_kernel, dim3((to - from + 63) / 64), dim3(64), 0, stream, rec, from, to, (float)status);
    return hipGetLastError();
}
__global__ void cvo_copy_records_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}
hipError_t launch_copy_records(const float* src, float* dst, int n, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(cvo_copy_records_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), 4 * n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- adaptive-ell variant
// acvo::align of thirdparty/cvo/src/adaptive_cvo.cpp:490-555, behind cvo_adaptive_align.  The reference does not build that file and
// ships no caller (thirdparty/cvo/CMakeLists.txt:66,77-81); this is a plain, dense evaluation of it -- a thread per row, every column
// visited, with the pair arithmetic, the row sums and the scalar epilogue of the main kernel (same device functions) -- spread over the
// device: the rows of a sweep are dealt to one-wave workgroups (grid over rows), each iteration is five small kernels on one stream
//   transform | sweep 1 (rows of Axy + Axx, then the rows of Ayy) | sums -> omega, v, dl | sweep 2 (step-size terms) | sums -> step, pose, ell
// and the host queues a few iterations at a time, looking at the stop flag in between (kernels queued behind a stop return at once).
// Per iteration (compute_flow, :154-272): Axy gives omega, v and -2 sum(a d2)/ell^3; Axx (the fixed cloud against itself) +sum(a d2)/ell^3;
// of Ayy (the transformed moving cloud against itself) only the rows from num_fixed on add their sum -- the reference never fills
// sum_diff_yy_2 for the rows below (:218-226 against :246-262) -- while all of its nonzeros count in the denominator (:271).  Then
// compute_step_size (:275-365, the base sequence), the stop tests (:509, :531), the pose update and ell += dl_step*dl inside
// [ell_min, ell_max], ell_max shrinking by 0.7 when hit (:538-545).  Partial sums are added in a fixed order: reproducible run to run.
constexpr int ADP_ROWS = 64;       // rows (threads) per workgroup of a sweep
__device__ __forceinline__ void adp_block_partials(double (&v)[16], int K, double* __restrict__ out) {
    for (int k = 0; k < K; ++k) v[k] = wave_sum_all(v[k]);
    if (threadIdx.x == 0) for (int k = 0; k < 16; ++k) out[k] = k < K ? v[k] : 0.0;
}
__global__ __launch_bounds__(256) void cvo_adaptive_transform_kernel(AdaptiveArgs A) {
    if (A.state->stop) return;
    __shared__ float sM[12];
    if (threadIdx.x == 0) {
        float Mx[12]; make_transform(A.state->R, A.state->T, Mx);                       // update_tf, :497
        for (int i = 0; i < 12; ++i) sM[i] = Mx[i];
        if (blockIdx.x == 0) for (int i = 0; i < 12; ++i) A.state->M[i] = Mx[i];
    }
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < A.nm) {                                                                     // transform_pcd, :500
        const float4 p = ld4((const gfloat*)A.moving + lo_off(j));
        float y0, y1, y2; apply_transform(sM, p.x, p.y, p.z, y0, y1, y2);
        GF4 ybuf{(gv4f*)A.ybuf}; ybuf.set(j, make_float4(y0, y1, y2, p.w));
    }
}
// blocks [0, nbx): rows of Axy and Axx; blocks [nbx, nbx + nby): rows of Ayy
__global__ __launch_bounds__(ADP_ROWS) void cvo_adaptive_sweep1_kernel(AdaptiveArgs A, int nbx) {
    if (A.state->stop) return;
    const gfloat* fixed = (const gfloat*)A.fixed; const gfloat* moving = (const gfloat*)A.moving;
    GF4 ybuf{(gv4f*)A.ybuf};
    const int N = A.nf, M = A.nm;
    const float ell = A.state->ell;
    const Gates gates = make_gates(ell, A.P);
    const float inv_c = 1 / A.P.c, inv_d = 1 / A.P.d;
    const float ell_3 = ell * ell * ell, inv_l3 = 1 / ell_3;                            // :172
    double acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                  // omega[3], v[3], dl, nnz_xy, nnz_xx, nnz_yy
    if ((int)blockIdx.x < nbx) {
        const int i = blockIdx.x * ADP_ROWS + threadIdx.x;
        if (i < N) {                                                                    // rows of Axy and Axx, :175-240
            const float4 lo = ld4(fixed + lo_off(i)), hi = ld4(fixed + hi_off(N, i));
            const float xi[3] = {lo.x, lo.y, lo.z}; const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
            float sw[3] = {0, 0, 0}, sv[3] = {0, 0, 0}, s_yx = 0.f, s_xx = 0.f; int nxy = 0, nxx = 0;
            for (int j = 0; j < M; ++j) {
                const float4 yj = ybuf[j];
                const float a = se_kernel_value(xi, fi, yj, ld4(moving + hi_off(M, j)), gates);
                if (a > 0.f) {
                    const float yv[3] = {yj.x, yj.y, yj.z};
                    float cr[3]; cross3(xi, yv, cr);                                                   // :203
                    sw[0] += a * cr[0]; sw[1] += a * cr[1]; sw[2] += a * cr[2];
                    const float e0 = yv[0] - xi[0], e1 = yv[1] - xi[1], e2 = yv[2] - xi[2];            // :204
                    sv[0] += a * e0; sv[1] += a * e1; sv[2] += a * e2;
                    s_yx += (inv_l3 * a) * ((e0 * e0 + e1 * e1) + e2 * e2);                            // :205, :231
                    ++nxy;
                }
            }
            for (int j = 0; j < N; ++j) {
                const float4 xj = ld4(fixed + lo_off(j));
                const float a = se_kernel_value(xi, fi, xj, ld4(fixed + hi_off(N, j)), gates);
                if (a > 0.f) {
                    const float e0 = xj.x - xi[0], e1 = xj.y - xi[1], e2 = xj.z - xi[2];               // :213-214
                    s_xx += (inv_l3 * a) * ((e0 * e0 + e1 * e1) + e2 * e2);                            // :234
                    ++nxx;
                }
            }
            for (int q = 0; q < 3; ++q) { acc[q] += (double)(inv_c * sw[q]); acc[3 + q] += (double)(inv_d * sv[q]); }   // :227-228
            acc[6] -= double(2 * s_yx);                                                 // :231
            acc[6] += double(s_xx);                                                     // :234
            acc[7] += nxy; acc[8] += nxx;
        }
    } else {
        const int i = ((int)blockIdx.x - nbx) * ADP_ROWS + threadIdx.x;
        if (i < M) {                                                                    // rows of Ayy: all of them count, those from num_fixed on add to dl (:243-266)
            const float4 yi4 = ybuf[i]; const float4 gi = ld4(moving + hi_off(M, i));
            const float yi[3] = {yi4.x, yi4.y, yi4.z}; const float fi[5] = {yi4.w, gi.x, gi.y, gi.z, gi.w};
            float s_yy = 0.f; int nyy = 0;
            for (int j = 0; j < M; ++j) {
                const float4 yj = ybuf[j];
                const float a = se_kernel_value(yi, fi, yj, ld4(moving + hi_off(M, j)), gates);
                if (a > 0.f) {
                    const float e0 = yj.x - yi[0], e1 = yj.y - yi[1], e2 = yj.z - yi[2];
                    s_yy += (inv_l3 * a) * ((e0 * e0 + e1 * e1) + e2 * e2);
                    ++nyy;
                }
            }
            if (i >= N) acc[6] += double(s_yy);
            acc[9] += nyy;
        }
    }
    adp_block_partials(acc, 10, A.partials + (size_t)blockIdx.x * 16);
}
// one workgroup: the sums of K values over nb partial records, in record order
__device__ __forceinline__ void adp_sum_partials(const double* __restrict__ partials, int nb, int K, double* tot /* shared, 16 */) {
    __shared__ double part[4][16];
    const int tid = threadIdx.x, q = tid & 15, sub = tid >> 4;                           // 64 threads: four strided sub-sums per value
    double s = 0;
    if (q < K) for (int r = sub; r < nb; r += 4) s += partials[(size_t)r * 16 + q];
    part[sub][q] = s;
    __syncthreads();
    if (tid < 16) tot[tid] = tid < K ? ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) : 0.0;
    __syncthreads();
}
__global__ __launch_bounds__(64) void cvo_adaptive_mid_kernel(AdaptiveArgs A, int nb) {
    if (A.state->stop) return;
    __shared__ double tot[16];
    adp_sum_partials(A.partials, nb, 10, tot);
    if (threadIdx.x == 0) {
        AdaptiveState& st = *A.state;
        for (int q = 0; q < 3; ++q) { st.omega[q] = (float)tot[q]; st.v[q] = (float)tot[3 + q]; }            // :269-270
        st.dl = tot[6] / (double)((long long)tot[8] + (long long)tot[9] - 2ll * (long long)tot[7]);          // :271
        st.nnz_xy = (int)tot[7]; st.nnz_xx = (int)tot[8]; st.nnz_yy = (int)tot[9];
    }
}
// compute_step_size, :275-365 (the base sequence): the nonzeros of Axy once more
__global__ __launch_bounds__(ADP_ROWS) void cvo_adaptive_sweep2_kernel(AdaptiveArgs A) {
    if (A.state->stop) return;
    const gfloat* fixed = (const gfloat*)A.fixed; const gfloat* moving = (const gfloat*)A.moving;
    GF4 ybuf{(gv4f*)A.ybuf};
    const int N = A.nf, M = A.nm;
    const float ell = A.state->ell;
    const Gates gates = make_gates(ell, A.P);
    float omega[3], v[3];
    for (int q = 0; q < 3; ++q) { omega[q] = A.state->omega[q]; v[q] = A.state->v[q]; }
    const LsConsts ls = make_ls(omega, v, ell);
    double b4[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int i = blockIdx.x * ADP_ROWS + threadIdx.x;
    if (i < N) {
        const float4 lo = ld4(fixed + lo_off(i)), hi = ld4(fixed + hi_off(N, i));
        const float xi[3] = {lo.x, lo.y, lo.z}; const float fi[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};
        double Bi = 0, Ci = 0, Di = 0, Ei = 0;
        for (int j = 0; j < M; ++j) {
            const float4 yj = ybuf[j];
            const float a = se_kernel_value(xi, fi, yj, ld4(moving + hi_off(M, j)), gates);
            if (a > 0.f) ls_terms(xi, yj, a, ls, Bi, Ci, Di, Ei);
        }
        b4[0] = Bi; b4[1] = Ci; b4[2] = Di; b4[3] = Ei;
    }
    adp_block_partials(b4, 4, A.partials + (size_t)blockIdx.x * 16);
}
__global__ __launch_bounds__(64) void cvo_adaptive_end_kernel(AdaptiveArgs A, int nb) {
    if (A.state->stop) return;
    __shared__ double tot[16];
    adp_sum_partials(A.partials, nb, 4, tot);
    if (threadIdx.x != 0) return;
    AdaptiveState& st = *A.state;
    const DevParams P = A.P;
    const int k = st.iterations_run;
    const float ell = st.ell;
    float omega[3], v[3];
    for (int q = 0; q < 3; ++q) { omega[q] = st.omega[q]; v[q] = st.v[q]; }
    const float c3 = (float)(4.0 * float(tot[3])), c2 = (float)(3.0 * float(tot[2])), c1 = (float)(2.0 * float(tot[1])), c0 = float(tot[0]);
    const float step = cubic_step(c3, c2, c1, c0, P.min_step);
    st.step = step;
    if (A.trace && k < A.trace_cap) {
        AdaptiveRow& tr = A.trace[k];
        for (int q = 0; q < 3; ++q) { tr.omega[q] = omega[q]; tr.v[q] = v[q]; }
        tr.dl = (float)st.dl; tr.ell = ell; tr.step = step; tr.nnz_xy = st.nnz_xy; tr.nnz_xx = st.nnz_xx; tr.nnz_yy = st.nnz_yy;
        *A.trace_len = k + 1;
    }
    const double nw = sqrt((double)omega[0] * omega[0] + (double)omega[1] * omega[1] + (double)omega[2] * omega[2]);
    const double nv = sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]);
    int stop = 0;
    if (nw < P.eps && nv < P.eps) stop = 1;                                             // :509
    else {
        float dR[9], dT[3], RdT[3], Rn[9], sR[9], sT[3];
        for (int i = 0; i < 9; ++i) sR[i] = st.R[i];
        for (int i = 0; i < 3; ++i) sT[i] = st.T[i];
        exp_sek3(omega, v, step, dR, dT);                                               // :520
        mat3_vec(sR, dT, RdT);
        for (int q = 0; q < 3; ++q) st.T[q] = RdT[q] + sT[q];                            // :527
        mat3_mul(sR, dR, Rn);
        for (int i = 0; i < 9; ++i) st.R[i] = Rn[i];                                     // :528
        if (dist_se3(dR, dT) < P.eps_2) stop = 1;                                       // :531
        else {
            float l = (float)((double)ell + (double)A.dl_step * st.dl);                 // :538
            if (l >= st.ell_max) { l = (float)(st.ell_max * 0.7); st.ell_max = (float)(st.ell_max * 0.7); }   // :541-544
            l = (l < A.ell_min) ? A.ell_min : l;                                        // :545
            st.ell = l;
        }
    }
    if (stop) st.iter = k;
    st.iterations_run = k + 1;
    if (stop || k + 1 >= P.max_iter) {
        float Rf[9], Tf[3], Mx[12];
        for (int i = 0; i < 9; ++i) Rf[i] = st.R[i];
        for (int i = 0; i < 3; ++i) Tf[i] = st.T[i];
        make_transform(Rf, Tf, Mx);                                                     // the final update_tf, :550
        for (int i = 0; i < 12; ++i) st.transform[i] = Mx[i];
        st.status = 0; st.stop = 1;
    }
}
// `iterations` iterations of the variant queued on `stream`; A.state->stop tells the host afterwards whether more are wanted
hipError_t launch_adaptive(const AdaptiveArgs& A, int iterations, hipStream_t stream) {
    const int nbx = (A.nf + ADP_ROWS - 1) / ADP_ROWS, nby = (A.nm + ADP_ROWS - 1) / ADP_ROWS;
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(cvo_adaptive_transform_kernel, dim3((A.nm + 255) / 256), dim3(256), 0, stream, A);
        hipLaunchKernelGGL(cvo_adaptive_sweep1_kernel, dim3(nbx + nby), dim3(ADP_ROWS), 0, stream, A, nbx);
        hipLaunchKernelGGL(cvo_adaptive_mid_kernel, dim3(1), dim3(64), 0, stream, A, nbx + nby);
        hipLaunchKernelGGL(cvo_adaptive_sweep2_kernel, dim3(nbx), dim3(ADP_ROWS), 0, stream, A);
        hipLaunchKernelGGL(cvo_adaptive_end_kernel, dim3(1), dim3(64), 0, stream, A, nbx);
    }
    return hipGetLastError();
}
int adaptive_partial_records(int nf, int nm) { return (nf + ADP_ROWS - 1) / ADP_ROWS + (nm + ADP_ROWS - 1) / ADP_ROWS; }

// ---------------------------------------------------------------- device known-answer test of the pair arithmetic (cvo_selftest_pair_values)
// One lane per case: the fixed point at the origin with zero features, the moving point y and its features g as given, so that d2 = y0^2 + y1^2 + y2^2 and
// d2c = sum g^2 with the kernels' own association.  out[4 i ..] = a by the four routes the align kernel has for it:
//   [0] se_kernel_value (dense fallback, adaptive variant: branches, exp_small)   [1] colour_factors + se_kernel_value_ck (lists outside Gates::poly_ok)
//   [2] colour_factors + se_kernel_values_flat (the 12-term chain)                [3] colour_factors + se_kernel_values_flat7 (degree 7 + guard)
// and out_aux[2 i ..] = {d2, d2c} as the device formed them.  Routes 2, 3 need Gates::poly_ok (the default parameters have it): 0 otherwise.
__global__ void selftest_pairs_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ aux, int n, float ell, DevParams P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const Gates G = make_gates(ell, P);
    const float xi[3] = {0.f, 0.f, 0.f}, fi[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int ii = min(i, n - 1);
    const float4 yj = make_float4(in[ii * 8 + 0], in[ii * 8 + 1], in[ii * 8 + 2], in[ii * 8 + 3]);
    const float4 gj = make_float4(in[ii * 8 + 4], in[ii * 8 + 5], in[ii * 8 + 6], in[ii * 8 + 7]);
    const float r0 = se_kernel_value(xi, fi, yj, gj, G);
    const float fb[5] = {yj.w, gj.x, gj.y, gj.z, gj.w};
    const float d2c1[1] = {feat_d2(fi, fb)};
    float ck1[1];
    colour_factors<1>(d2c1, G, ck1);
    const float r1 = se_kernel_value_ck(xi, yj, ck1[0], G);
    float r2 = 0.f, r3 = 0.f, d2s[1] = {0.f};
    if (G.poly_ok) {
        const float4 yv[1] = {yj}; const bool act[1] = {true};
        float a1[1], e1[1][3];
        se_kernel_values_flat<1>(xi, yv, ck1, act, G, a1, e1, d2s);
        r2 = a1[0];
        const Exp7 E7 = make_exp7(G);
        se_kernel_values_flat7<1>(xi, yv, ck1, act, G, E7, a1, e1);
        r3 = a1[0];
    } else {
        const float e0 = -yj.x, e1 = -yj.y, e2 = -yj.z;
        float d2 = e0 * e0; d2 = d2 + e1 * e1; d2 = d2 + e2 * e2;
        d2s[0] = d2;
    }
    if (i < n) {
        out[i * 4 + 0] = r0; out[i * 4 + 1] = r1; out[i * 4 + 2] = r2; out[i * 4 + 3] = r3;
        aux[i * 2 + 0] = d2s[0]; aux[i * 2 + 1] = d2c1[0];
    }
}
hipError_t launch_selftest_pairs(const float* in, float* out, float* aux, int n, float ell, const DevParams& P, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(selftest_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, aux, n, ell, P);
    return hipGetLastError();
}

int align_blocks_per_cu() { return BLOCK_MAX > 512 ? 1 : CVO_WAVES_PER_SIMD / 2; }
int align_block_max() { return BLOCK_MAX; }
int align_adopt_gmax() { return ADOPT_GMAX; }

// LDS: Shared | slot/row tables (3 x rows_cap u16) | sort histograms | group boxes | cull tile (3*tile floats) | resident y cloud
// (y_mode 1: 16 B x y_cap, y_mode 2: 12 B x y_cap, y_mode 0: none)
// tab_cols: columns of the line-search table (16 bytes each, laid over the rebuild scratch behind the resident cloud; 0 = none).
// y_mode 2: the rebuild scratch lives in the cull tile itself (lds_layout), which therefore has a minimum size (align_min_tile).
size_t align_scratch_bytes(int tile, int rows_cap, int y_mode, int y_cap) {
    return (size_t)rows_cap * sizeof(uint16_t) + (size_t)2 * MAX_WAVES * NCLS * sizeof(int) + (size_t)8 * ((y_mode == 2 ? y_cap : tile) >> 5) * sizeof(float);
}
int align_min_tile(int rows_cap, int y_mode, int y_cap) {
    if (y_mode != 2) return 128;
    const size_t need = align_scratch_bytes(0, rows_cap, 2, y_cap);
    return (int)(((need + 3 * sizeof(float) - 1) / (3 * sizeof(float)) + 127) / 128 * 128);
}
size_t align_shared_bytes(int tile, int rows_cap, int y_mode, int y_cap, int tab_cols) {
    const size_t ybytes = y_mode == 1 ? (size_t)y_cap * sizeof(float4) : (y_mode == 2 ? (size_t)y_cap * 3 * sizeof(float) : 0);
    const size_t scratch = y_mode == 2 ? 0 : align_scratch_bytes(tile, rows_cap, y_mode, y_cap);
    return ((sizeof(Shared) + 15) & ~size_t(15)) + (size_t)2 * rows_cap * sizeof(uint16_t) + (size_t)3 * tile * sizeof(float) + ybytes +
           std::max(scratch, (size_t)tab_cols * sizeof(float4));
}
int align_tile_granule() { return 128; }                            // keeps every LDS section 16-byte aligned

hipError_t launch_align(int grid, int block, int tile, int rows_cap, int y_mode, int y_cap, int tab_cols, hipStream_t stream, const PairDesc* descs, int n_pairs, int G,
                        unsigned launch_tag, unsigned long long* queue, const DevParams& P, const unsigned* wgs_submitted, unsigned* wgs_started, const float* const* raw_table) {
    const size_t shmem = align_shared_bytes(tile, rows_cap, y_mode, y_cap, tab_cols);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cvo_align_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cvo_align_kernel, dim3(grid), dim3(block), shmem, stream, descs, n_pairs, G, tile, y_mode, rows_cap, y_cap, launch_tag, tab_cols, queue, P,
                       wgs_submitted, wgs_started, raw_table);
    return hipGetLastError();
}

}  // namespace CVO_KNS
