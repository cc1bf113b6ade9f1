// cvo_capi.hip -- host side of libcvo_hip.so: the C ABI of include/cvo_hip.h.
//
// Holds the state machine of the reference's cvo::cvo object (cloud slots,
// R/T/ell carried between calls, transform bookkeeping; thirdparty/cvo/src/
// cvo.cpp:345-386, 461-618) and drives the gfx950 kernels.  There is no CPU
// compute path: without a gfx950 device every entry point fails with
// CVO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <functional>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cvo_hip.h"
#include "cvo_device.h"
#include "cvo_math.hpp"

// the align kernel's translation unit is in the library twice (cvo_kernels.hip, head comment): cvohip = two waves per SIMD, cvohip_w3 = three
namespace cvohip_w3 {
using cvohip::PairDesc; using cvohip::DevParams;
size_t align_shared_bytes(int tile, int rows_cap, int y_mode, int y_cap, int tab_cols);
int align_min_tile(int rows_cap, int y_mode, int y_cap);
int align_tile_granule();
int align_block_max();
hipError_t launch_align(int grid, int block, int tile, int rows_cap, int y_mode, int y_cap, int tab_cols, hipStream_t stream, const PairDesc* descs, int n_pairs, int G,
                        unsigned launch_tag, unsigned long long* queue, const DevParams& P, const unsigned* wgs_submitted, unsigned* wgs_started, const float* const* raw_table);
}
namespace cvohip {
size_t align_shared_bytes(int tile, int rows_cap, int y_mode, int y_cap, int tab_cols);
int align_min_tile(int rows_cap, int y_mode, int y_cap);
int align_tile_granule();
int align_blocks_per_cu();
int align_block_max();
hipError_t launch_align(int grid, int block, int tile, int rows_cap, int y_mode, int y_cap, int tab_cols, hipStream_t stream, const PairDesc* descs, int n_pairs, int G,
                        unsigned launch_tag, unsigned long long* queue, const DevParams& P, const unsigned* wgs_submitted, unsigned* wgs_started, const float* const* raw_table);
int align_adopt_gmax();
hipError_t launch_fill_records(float* rec, int from, int to, int status, hipStream_t stream);
hipError_t launch_copy_records(const float* src, float* dst, int n, hipStream_t stream);
hipError_t launch_pack_clouds(const float* raw, const PackDesc* descs, int n_clouds, int n_max, hipStream_t s);
SelfCacheEntry* score_self_cache(float* gbox, int n);
hipError_t pcd_launch_pyramid(const uint8_t* bgr, int w, int h, float* I0, float* I1, float* I2, float* dx0, float* dy0, float* abs0, float* abs1, float* abs2,
                              hipStream_t s);
hipError_t pcd_launch_thresholds(const float* abs0, int w, int h, float* ths, float* ths_smoothed, hipStream_t s);
hipError_t pcd_launch_select(const float* abs0, const float* abs1, const float* abs2, const float* ths_smoothed, int w, int h, int pot, uint8_t* map, int* counts,
                             hipStream_t s);
int pcd_tiles(int w, int h);
hipError_t pcd_launch_subsample(uint8_t* map, const uint8_t* pattern, int subsample, int char_th, const uint16_t* depth, int w, int h, int* tile_counts, hipStream_t s);
hipError_t pcd_launch_cloud(const uint8_t* map, const uint16_t* depth, const uint8_t* bgr, const float* dx0, const float* dy0, int w, int h, const float cam[5],
                            const int* tile_counts, int n_points, float* cloud, uint16_t* px, hipStream_t s);
hipError_t pcd_launch_unpack(const float* cloud, int n, float* xyz, float* feat, hipStream_t s);
int score_nout();
int score_row_blocks(int na);
int score_groups(int n);
size_t score_box_bytes(int n);
hipError_t launch_cloud_boxes(const float* rec, int n, float* gbox, hipStream_t stream);
hipError_t launch_adaptive(const AdaptiveArgs& A, int iterations, hipStream_t stream);
int adaptive_partial_records(int nf, int nm);
hipError_t launch_selftest(int kind, const float* in, float* out, int n, hipStream_t s);
hipError_t launch_selftest_pairs(const float* in, float* out, float* aux, int n, float ell, const DevParams& P, hipStream_t s);
hipError_t launch_score(const ScoreBatch& B, const ScoreDesc* more, int nreq, int row_blocks, int chunks, const DevParams& P, double* partials,
                        double* out_pinned, hipStream_t stream, unsigned* wgs_started, bool* sweep_submitted);
}  // namespace cvohip

using namespace cvohip;

static_assert(sizeof(cvo_trace_row) == sizeof(TraceRow), "trace row layout");
static_assert(sizeof(cvo_adaptive_row) == sizeof(AdaptiveRow), "adaptive trace row layout");

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return fail(CVO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return CVO_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 4;
        HIP_TRY(hipMalloc(&p, want));
        bytes = want;
        return CVO_OK;
    }
    // grow to `need` bytes keeping the first `keep` bytes (copied on `s`, the old block freed when that copy has run)
    int grow_keep(size_t need, size_t keep, hipStream_t s) {
        if (need <= bytes) return CVO_OK;
        void* np = nullptr; const size_t want = need + need / 4;
        HIP_TRY(hipMalloc(&np, want));
        if (p && keep) {
            hipError_t e = hipMemcpyAsync(np, p, std::min(keep, bytes), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipFree(np); return fail(CVO_ERR_HIP, std::string("growing a device buffer: ") + hipGetErrorString(e)); }
        }
        if (p) (void)hipFree(p);
        p = np; bytes = want;
        return CVO_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};
struct PinBuf {
    void* p = nullptr; size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return CVO_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        HIP_TRY(hipHostMalloc(&p, need + need / 4, hipHostMallocDefault));
        bytes = need + need / 4;
        return CVO_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

// a point cloud resident in HBM: two planes of n float4, {x,y,z,f0} then {f1..f4} (cvo_device.h)
struct Cloud {
    DevBuf buf; int n = 0;
    DevBuf px; int n_px = 0;        // selected pixel (x, y) per point, when the cloud was generated from images
    // boxes of the 32-point groups for the score kernels, made on first use after the points were written
    mutable DevBuf boxes; mutable bool boxes_valid = false; mutable hipStream_t boxes_stream = nullptr;
    // a hand-over the device has not packed yet: the cloud's arrays as they came (n x 3 positions, then 5 channel-major feature arrays) in
    // the engine's pinned staging ring; the next align launch packs them itself, any other consumer runs the pack kernel first
    mutable const float* raw = nullptr;
    mutable const float* raw_feat = nullptr;   // with `raw`: the feature arrays when they do not lie right behind the positions (clouds handed over in caller-registered memory)
    float cost_hint = 0.f;          // mean 1/z^2 of a sample of the points (0 = unknown): what a pair costs per iteration follows the density of its clouds
    float* rec() const { return static_cast<float*>(buf.p); }
    ~Cloud() { buf.release(); px.release(); boxes.release(); }
};

// The images a cloud was generated from, as they were staged for the device (pinned host memory): kept after the generation so that a second handle given the
// SAME frame -- the tracker's cvo_odometry and cvo_keyframe objects build their MOVING cloud from one image with a deterministic selector (local_tracker.cpp:356,415;
// SURVEY appendix B) -- can recognise it byte for byte and take a device copy of the finished cloud instead of generating it again (cvo_set_pcd_images).  `stamp`
// is raised before the owner overwrites the buffer: a compare that saw the same stamp before and after has read one image.
struct ImageStage {
    PinBuf buf; std::atomic<unsigned long long> stamp{0};
    std::mutex mu;              // held while the owner rewrites `buf` and while anybody compares against it (the stamp alone would leave memcpy racing memcmp formally)
    ~ImageStage() { buf.release(); }
};
struct LastGenerated {          // per host thread: the handles of one thread are used one after the other, so whoever shares never races with the generator
    std::shared_ptr<ImageStage> stage; unsigned long long stamp = 0;
    int device = -1, w = 0, h = 0, num_want = 0; cvo_camera cam{};
    std::shared_ptr<Cloud> cloud;
};
LastGenerated& last_generated() { static thread_local LastGenerated g; return g; }
struct FrameStager;
FrameStager*& frame_stager_slot() { static thread_local FrameStager* s = nullptr; return s; }
bool share_generated_clouds() { static const bool on = [] { const char* e = std::getenv("CVO_HIP_SHARE_CLOUDS"); return !e || std::atoi(e) != 0; }(); return on; }

// Caller-registered host memory (cvo_host_register): clouds handed over from inside such a range are not copied into the staging ring at all -- the align launch that
// builds their device layout reads them where they are (registered memory is pinned and mapped into the device's address space, like the ring).
struct HostRange { const unsigned char* lo; const unsigned char* hi; };
struct HostRegistry { std::mutex mu; std::vector<HostRange> ranges; };
HostRegistry& host_registry() { static HostRegistry r; return r; }
bool in_registered_memory(const void* p, size_t bytes) {
    HostRegistry& r = host_registry();
    if (r.ranges.empty()) return false;
    std::lock_guard<std::mutex> lk(r.mu);
    const unsigned char* q = static_cast<const unsigned char*>(p);
    for (const HostRange& g : r.ranges) if (q >= g.lo && q + bytes <= g.hi) return true;
    return false;
}

// The hand-over's copy into the pinned ring: 12.6 MB per 64-pair step that the host never reads again.  Ordinary stores pull every destination line into the cache first
// (read-for-ownership) and push the caller's data out of it; non-temporal 16-byte stores write the lines straight through (CVO_HIP_UPLOAD_NT=1; off by default: measured, no difference).
inline void ring_copy(void* dst, const void* src, size_t bytes, bool nt) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (nt && bytes >= 4096) {
        typedef long long v2di __attribute__((vector_size(16)));
        unsigned char* d = static_cast<unsigned char*>(dst); const unsigned char* s = static_cast<const unsigned char*>(src);
        const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
        if (head) { std::memcpy(d, s, head); d += head; s += head; bytes -= head; }
        const size_t body = bytes & ~(size_t)63;
        for (size_t o = 0; o < body; o += 64) {
            v2di a, b, c, e;
            std::memcpy(&a, s + o, 16); std::memcpy(&b, s + o + 16, 16); std::memcpy(&c, s + o + 32, 16); std::memcpy(&e, s + o + 48, 16);
            __builtin_nontemporal_store(a, reinterpret_cast<v2di*>(d + o)); __builtin_nontemporal_store(b, reinterpret_cast<v2di*>(d + o + 16));
            __builtin_nontemporal_store(c, reinterpret_cast<v2di*>(d + o + 32)); __builtin_nontemporal_store(e, reinterpret_cast<v2di*>(d + o + 48));
        }
        if (bytes > body) std::memcpy(d + body, s + body, bytes - body);
        asm volatile("sfence" ::: "memory");
        return;
    }
#endif
    (void)nt;
    std::memcpy(dst, src, bytes);
}

// Copy threads of the hand-over (Engine::upload_many: a batch's host arrays into the pinned ring, 12.6 MB per 64-pair step).  They live as long as the process:
// starting three threads per hand-over cost 0.1 ms of the host's 0.5 per step, and a step's launch is resubmitted that much later (profiles/r04_upload_pool.txt).
class CopyPool {
public:
    static CopyPool& get() { static CopyPool* p = new CopyPool(); return *p; }   // never destroyed: its threads may outlive main()'s statics
    // runs fn(0) .. fn(parts - 1), part 0 on the caller; returns when all are done.  One job at a time (callers queue on the lock).
    template <class F> void run(int parts, F&& fn) {
        if (parts <= 1 || workers_.empty()) { for (int i = 0; i < parts; ++i) fn(i); return; }
        std::unique_lock<std::mutex> job(job_mutex_);
        {
            std::lock_guard<std::mutex> lk(m_);
            task_ = [&fn](int i) { fn(i); }; next_ = 1; parts_ = parts; left_ = parts - 1; ++epoch_;
        }
        cv_.notify_all();
        fn(0);
        for (;;) {                                                    // the caller takes parts too while it waits
            int mine = -1;
            { std::lock_guard<std::mutex> lk(m_); if (next_ < parts_) mine = next_++; }
            if (mine < 0) break;
            fn(mine);
            { std::lock_guard<std::mutex> lk(m_); --left_; }
        }
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return left_ == 0; });
        task_ = nullptr;
    }
private:
    CopyPool() {
        const int hw = (int)std::thread::hardware_concurrency();
        const int n = std::max(0, std::min(7, hw - 1));
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
        for (std::thread& t : workers_) t.detach();
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return epoch_ != seen && next_ < parts_; });
            if (next_ >= parts_) { seen = epoch_; continue; }
            const int mine = next_++;
            if (next_ >= parts_) seen = epoch_;
            auto task = task_;
            lk.unlock();
            task(mine);
            lk.lock();
            if (--left_ == 0) done_.notify_all();
        }
    }
    std::mutex m_, job_mutex_;
    std::condition_variable cv_, done_;
    std::function<void(int)> task_;
    int next_ = 0, parts_ = 0, left_ = 0;
    unsigned long long epoch_ = 0;
    std::vector<std::thread> workers_;
};

DevParams to_dev(const cvo_params& p) {
    DevParams d;
    d.sigma = p.sigma; d.sp_thres = p.sp_thres; d.c = p.c; d.d = p.d; d.c_ell = p.c_ell; d.c_sigma = p.c_sigma;
    d.min_step = p.min_step; d.eps = p.eps; d.eps_2 = p.eps_2; d.max_iter = p.max_iter;
    d.skin = 0.25f;
    d.skin_alpha = 0.f; d.alpha_gamma = 0.f; d.first_scale = 1.f; d.fuse_refine = 1;
    d.nt_min = 0;
    d.overlap_stop_test = 1;
    d.predict = 0.7f; d.predict_steps = 8.f;   // lists built / filtered 0.7 of every point's allowance ahead on the path: -10 % culls, +1 % (profiles/r04_predicted_list_centres.txt)
    d.resort = 1;
    d.adopt_kmax = 40; d.adopt_on = 0; d.adopt_inject = 0; d.adopt_dwell = 0;
    d.colocate = 1;
    return d;
}

int check_device(int device, int* num_cus) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(CVO_ERR_NO_DEVICE, "no HIP device visible: libcvo_hip has no CPU path");
    if (device < 0 || device >= count) return fail(CVO_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CVO_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", libcvo_hip is built for gfx950 only");
    if (num_cus) *num_cus = prop.multiProcessorCount;
    return CVO_OK;
}

// Co-residency of cooperating workgroups.  The G > 1 workgroups of a pair wait for each other inside the kernel (tagged granules
// through L2), so all of a launch's workgroups must be resident at the same time.  One launch alone is sized to the CU count, but
// launches of OTHER handles (tracker thread + back-end thread, keyframe_graph.cpp:212-240; a thread pool of loop-closure checks) share
// the CUs: with more cooperating workgroups in flight than the device holds, every launch can end up partially resident and spin until
// the in-kernel timeout.  So the library keeps a book, per device and process-wide, of the cooperative (G > 1) launches in flight and
// the workgroup slots they hold.  A launch that does not fit beside them takes fewer workgroups per pair -- down to G = 1, which waits
// for nobody and needs no slot; results do not depend on G.  When its clouds force G > 1 (more than MAX_ROWS_PER_WG rows per
// workgroup otherwise) it is queued BEHIND the launches in the book instead (hipStreamWaitEvent on their completion events: no host
// thread ever blocks), so it starts on an empty device.  Entries leave the book when their launch has been waited for or the handle
// goes away.  Cooperative launches are submitted under the book's lock (acquire -> kernel launch -> completion event), a few
// microseconds each.  Launches of other processes on the same GPU are outside this book.
struct SlotBook {
    struct Holder { const void* engine; int device; int slots; hipEvent_t done; };
    std::mutex mu;
    std::vector<Holder> holders;
};
SlotBook& slot_book() { static SlotBook b; return b; }

// Adoption (cvo_kernels.hip: finished workgroups help with pairs that still run): a workgroup only offers its help when nothing is
// queued on the device, i.e. when every workgroup the library has submitted there has started.  Two counters per device: submitted
// (host-written before each launch, pinned host memory the kernels read) and started (device memory, bumped by every workgroup).
// EVERY align launch of the process counts (with or without adoption, cooperative, queue mode), and so does every score launch: those are
// the kernels whose workgroups wait for a compute unit while persistent align workgroups hold them.  The two counters only ever move
// together: a launch adds its grid to `submitted` before it is submitted (and takes it back when the submission fails), and every one of
// its workgroups adds itself to `started` first thing -- no host read of the device counter on any healthy path; a launch call that reports an error re-synchronises
// them once the device has drained (resync_adopt_counters).
struct AdoptCounters { int device = -1; unsigned* submitted_host = nullptr; unsigned* submitted_dev = nullptr; unsigned* started_dev = nullptr; };
AdoptCounters* adopt_counters(int device) {                          // null when they cannot be made: launches then run without adoption
    static std::mutex mu; static std::vector<AdoptCounters*> all;
    std::lock_guard<std::mutex> lk(mu);
    for (AdoptCounters* a : all) if (a->device == device) return a->submitted_host ? a : nullptr;
    AdoptCounters* a = new AdoptCounters; a->device = device;
    void* h = nullptr; void* d = nullptr; void* s = nullptr;
    if (hipHostMalloc(&h, sizeof(unsigned), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess &&
        hipMalloc(&s, sizeof(unsigned)) == hipSuccess && hipMemset(s, 0, sizeof(unsigned)) == hipSuccess) {
        a->submitted_host = static_cast<unsigned*>(h); *a->submitted_host = 0u; a->submitted_dev = static_cast<unsigned*>(d); a->started_dev = static_cast<unsigned*>(s);
    } else { (void)hipGetLastError(); }
    all.push_back(a);
    return a->submitted_host ? a : nullptr;
}
std::mutex& adopt_submit_mutex() { static std::mutex m; return m; }
// A launch call has reported an error.  It may still have been enqueued (hipGetLastError can hand out an earlier, sticky error), so "take the grid back from
// `submitted`" could leave the two counters apart for the life of the process -- and the helpers silent (submitted != started reads "work is queued").  Errors are
// rare: let the device drain and set `submitted` to what has really started.  Called with the submit lock held (nothing counted is submitted meanwhile).
void resync_adopt_counters(AdoptCounters* a) {
    (void)hipDeviceSynchronize();
    unsigned st = 0;
    if (hipMemcpy(&st, a->started_dev, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess) {
        if (*a->submitted_host != st) std::fprintf(stderr, "cvo_hip: launch error on device %d: adoption counters re-synchronised (submitted %u, started %u)\n", a->device, *a->submitted_host, st);
        *a->submitted_host = st;
    }
    (void)hipGetLastError();
}

// Launch machinery shared by single-object handles and batches.
struct Engine {
    int device = 0, num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevParams P;
    bool wide_waves = true;          // CVO_HIP_WIDE=0: plane-layout launches run the two-waves-per-SIMD build as well
    bool wide_all = false;           // CVO_HIP_WIDE=2: the float4 layout runs the three-waves-per-SIMD build too (measured again in round 4: profiles/r04_three_waves_tum.txt)
    bool gamma_set = false;          // CVO_HIP_ALPHA_GAMMA given: else the depth-proportional margin falls with ell (gamma 1) in the float4 layout: +1 % (profiles/r04_list_margin_gamma.txt)
    bool alpha_auto = true;          // ... and its depth-proportional part (CVO_HIP_SKIN_ALPHA fixes it; CVO_HIP_SKIN alone = one margin for all rows, alpha 0)
    bool skin_auto = true;           // the list radius margin follows the layout: 0.35 with the cloud resident as 16-byte points (3 k-point shape), 0.30 otherwise -- measured with
                                     // the device full (profiles/r03_skin_sweep.txt): a cull costs LDS and issue time only, a longer list costs memory traffic, and that is dearer
                                     // with 256 workgroups streaming than alone (round 2's 0.25 was tuned on one launch); CVO_HIP_SKIN fixes it
    DevBuf d_descs, d_states, d_ybuf, d_jT, d_ent, d_surv, d_xch, d_queue, d_trace, d_tracelen, d_partials;
    // 64-byte result records, one per pair, written by the align kernel (PairDesc::record); the block a rank contributes to the
    // cross-GPU gather may be longer than its pairs (padding records: pad_from .. pad_to carry pad_status, see padded_records)
    DevBuf d_records;
    int rec_hint = 0;            // records to make room for beyond the pairs of a launch (a batch: its max_pairs + 1)
    int pad_from = 0, pad_to = 0, pad_status = 0;
    PinBuf h_descs, h_states, h_states_in, h_stage, h_partials;   // h_states: final states, written by the kernel itself (mapped pinned memory)
    std::vector<unsigned char> descs_uploaded;                     // what d_descs holds: unchanged descriptors are not sent again
    unsigned launch_seq = 0;
    size_t xch_zeroed_bytes = 0;
    int wg_request = 0;          // 0 = auto
    int tile_request = 0;        // 0 = auto
    int capf_request = 0;        // flat capacity per row (0 = auto)
    int block_request = 0;       // threads per workgroup (0 = auto)
    int per_cu = 1;              // workgroups resident per CU: 1 x 512 threads, or 2 x 256 threads (half the LDS each)
    int max_wgs = 0;             // cap on the workgroups of one launch (0 = none): a share of the device for launches that run side by side
    float last_ms = 0.f;
    bool launched = false;
    int last_G = 1;              // workgroups per pair of the last launch
    bool tail_scores = false;    // align launches also answer the tracker's score block for every pair in the kernel's tail (PairDesc::score_out)
    bool last_tail = false;      // ... and the last launch did
    PinBuf h_tail;               // n x 5 x 24 doubles, written by the kernel
    bool adopt = false;          // finished workgroups help with the pairs of their launch that still run (one workgroup and one slot per pair; CVO_HIP_ADOPT)

    void release_slots() {
        SlotBook& b = slot_book();
        std::lock_guard<std::mutex> lk(b.mu);
        for (size_t i = 0; i < b.holders.size();) { if (b.holders[i].engine == this) b.holders.erase(b.holders.begin() + i); else ++i; }
    }
    // How many workgroups per pair this launch may use (<= G_want, >= g_min) given the cooperative launches in flight.  Returns with
    // `lk` locked when the launch is cooperative itself (the caller records its completion event, then unlocks); `wait_for` receives
    // the completion events the launch has to be queued behind when it does not fit beside the launches in the book.
    int acquire_slots(int G_want, int g_min, int n_pairs, std::unique_lock<std::mutex>& lk, std::vector<hipEvent_t>& wait_for) {
        const int capacity = num_cus * per_cu;
        const int share = max_wgs > 0 ? std::min(max_wgs, capacity) : capacity;   // what this launch may take at most
        SlotBook& b = slot_book();
        lk = std::unique_lock<std::mutex>(b.mu);
        for (size_t i = 0; i < b.holders.size();) { if (b.holders[i].engine == this) b.holders.erase(b.holders.begin() + i); else ++i; }   // an earlier launch of this engine precedes this one on its stream
        int used = 0;
        for (const SlotBook::Holder& h : b.holders) if (h.device == device) used += h.slots;
        auto need = [&](int G) { return std::max(1, std::min(n_pairs, share / G)) * G; };
        int G = std::max(G_want, g_min);
        while (G > g_min && G > 1 && need(G) > capacity - used) G /= 2;
        G = std::max(G, g_min);
        if (G <= 1) { lk.unlock(); return 1; }                        // G = 1 waits for nobody: always safe, not in the book
        if (need(G) > capacity - used)                                 // forced cooperative launch on a busy device: run after the others
            for (const SlotBook::Holder& h : b.holders) if (h.device == device) wait_for.push_back(h.done);
        b.holders.push_back(SlotBook::Holder{this, device, need(G), ev1});
        return G;
    }

    int init(int dev, const cvo_params& prm) {
        device = dev;
        int rc = check_device(dev, &num_cus); if (rc) return rc;
        HIP_TRY(hipSetDevice(dev));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
        P = to_dev(prm);
        if (const char* e = std::getenv("CVO_HIP_FLAT_CAP")) capf_request = std::atoi(e);
        if (const char* e = std::getenv("CVO_HIP_BLOCK")) block_request = std::atoi(e);
        if (const char* e = std::getenv("CVO_HIP_WGS_PER_CU")) per_cu = std::max(1, std::min(2, std::atoi(e)));
        if (const char* e = std::getenv("CVO_HIP_SKIN")) { P.skin = (float)std::atof(e); skin_auto = false; }
        if (const char* e = std::getenv("CVO_HIP_SKIN_ALPHA")) { P.skin_alpha = std::max(0.f, std::min(0.2f, (float)std::atof(e))); alpha_auto = false; }
        if (const char* e = std::getenv("CVO_HIP_ALPHA_GAMMA")) { P.alpha_gamma = std::max(0.f, std::min(4.f, (float)std::atof(e))); gamma_set = true; }
        if (const char* e = std::getenv("CVO_HIP_FUSE_REFINE")) P.fuse_refine = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_FIRST_SCALE")) { P.first_scale = std::max(0.f, std::min(16.f, (float)std::atof(e))); first_scale_set = true; }
        if (const char* e = std::getenv("CVO_HIP_PREDICT")) P.predict = std::max(0.f, std::min(0.99f, (float)std::atof(e)));
        if (const char* e = std::getenv("CVO_HIP_PREDICT_STEPS")) P.predict_steps = std::max(0.f, (float)std::atof(e));
        if (const char* e = std::getenv("CVO_HIP_OVERLAP_STOP")) P.overlap_stop_test = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_NT_MIN")) P.nt_min = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("CVO_HIP_RESORT")) P.resort = std::max(0, std::min(2, std::atoi(e)));
        if (const char* e = std::getenv("CVO_HIP_COLOCATE")) P.colocate = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_WIDE")) { wide_waves = std::atoi(e) != 0; wide_all = std::atoi(e) >= 2; }
        if (const char* e = std::getenv("CVO_HIP_ADOPT_KMAX")) P.adopt_kmax = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("CVO_HIP_ADOPT_INJECT")) P.adopt_inject = std::atoi(e);
        if (const char* e = std::getenv("CVO_HIP_ADOPT_DWELL_US")) P.adopt_dwell = std::max(0, std::min(100000, std::atoi(e))) * 100;
        if (const char* e = std::getenv("CVO_HIP_ADOPT")) adopt = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_TILE")) tile_request = std::atoi(e);
        if (const char* e = std::getenv("CVO_HIP_WGS")) wg_request = std::atoi(e);
        if (const char* e = std::getenv("CVO_HIP_SELF_CACHE")) self_cache_on = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_UPLOAD_COPY")) upload_copy = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_ORDER_PAIRS")) { order_mode = std::atoi(e); order_pairs = order_mode != 0; }
        if (const char* e = std::getenv("CVO_HIP_INKERNEL_PACK")) inkernel_pack = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_RING_MIRROR")) ring_mirror = std::atoi(e) != 0;
        if (const char* e = std::getenv("CVO_HIP_UPLOAD_THREADS")) upload_threads = std::max(1, std::min(16, std::atoi(e)));
        upload_threads = std::max(1, std::min(upload_threads, (int)std::thread::hardware_concurrency()));
        if (const char* e = std::getenv("CVO_HIP_UPLOAD_NT")) upload_nt = std::atoi(e) != 0;
        return CVO_OK;
    }
    void destroy() {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (last_stream && last_stream != stream) (void)hipStreamSynchronize(last_stream);
        release_slots();
        d_scoredescs.release(); h_scoredescs.release(); h_counts.release();
        for (DevBuf* b : {&d_bgr, &d_depth, &d_I0, &d_I1, &d_I2, &d_dx0, &d_dy0, &d_abs0, &d_abs1, &d_abs2, &d_ths, &d_thsS, &d_map, &d_pattern, &d_counts, &d_tiles}) b->release();
        for (DevBuf* b : {&d_descs, &d_states, &d_ybuf, &d_jT, &d_ent, &d_surv, &d_xch, &d_queue, &d_trace, &d_tracelen, &d_partials, &d_raw, &d_ring, &d_records}) b->release();
        for (PinBuf* b : {&h_descs, &h_states, &h_states_in, &h_stage, &h_partials, &h_tail, &h_packdesc, &h_rawtab}) b->release();
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev_ring) (void)hipEventDestroy(ev_ring);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr; ev0 = ev1 = nullptr;
    }

    // Host arrays in the reference layout -> the clouds' float4 planes in HBM.  All clouds of one hand-over travel together: their
    // arrays are copied as they are (memcpy, no per-point work on the host) into one block of the pinned staging ring behind a table of
    // PackDesc, ONE host-to-device copy brings block and table over, ONE kernel builds the planes (cvo_pack_clouds_kernel).  The ring
    // is only waited for when it wraps.  Large hand-overs (a batch of 64 pairs = 12.6 MB) are copied into the ring by a few threads.
    struct UploadItem { Cloud* c; const float* xyz; const float* feat; int n; };
    DevBuf d_raw;
    bool first_scale_set = false; float coop_first_scale = 0.f;   // (cvo_batch_create sets the latter)
    int upload_threads = 8;
    bool upload_nt = false;          // measured: no difference in the hand-over loop (profiles/r04_upload_nt_ab.txt) -- the host copy is not what that loop waits for
    bool upload_copy = true;
    bool defer_pack = false;          // batches (cvo_batch_create)
    bool order_pairs = true;          // CVO_HIP_ORDER_PAIRS=0: positions take the pairs in index order
    int order_mode = 1;
    bool inkernel_pack = true;        // CVO_HIP_INKERNEL_PACK=0: deferred clouds always go through the pack kernel
    // Experiment (CVO_HIP_RING_MIRROR=1): a batch's block of the ring is also copied to a device mirror by the copy engine, on the engine's stream, at the hand-over,
    // and the align launch behind it builds the clouds' planes from the mirror instead of reading the pinned ring over PCIe from inside the kernel.  Measured
    // interleaved on one lease: with_host_upload / value 0.928-0.965 on the 20-step command against 0.951-0.959 without, 0.959-0.965 against 0.954-0.957 on the
    // 256-step run: within the run-to-run spread, not kept (profiles/r04_upload_mirror_ab.txt).
    bool ring_mirror = false;
    DevBuf d_ring;                    // device mirror of h_stage (same offsets)
    hipEvent_t ev_ring = nullptr;     // recorded behind the last mirror copy
    std::vector<Cloud*> pending;      // clouds with a hand-over not packed yet (Cloud::raw)
    PinBuf h_packdesc, h_rawtab;
    hipStream_t packdesc_stream = nullptr;
    std::vector<hipStream_t> ring_readers;   // every stream with work queued that reads the staging ring (pack kernels, launches that pull raw clouds): all of them before the ring is written over
    void add_ring_reader(hipStream_t s) { if (std::find(ring_readers.begin(), ring_readers.end(), s) == ring_readers.end()) ring_readers.push_back(s); }
    int sync_ring_readers() {
        for (hipStream_t r : ring_readers) if (r != stream) HIP_TRY(hipStreamSynchronize(r));
        ring_readers.clear();
        return CVO_OK;
    }
    // pack kernel over every pending cloud, on stream s (zero-copy from the ring)
    int flush_pending(hipStream_t s) {
        if (pending.empty()) return CVO_OK;
        if (packdesc_stream) { HIP_TRY(hipStreamSynchronize(packdesc_stream)); packdesc_stream = nullptr; }   // an earlier flush may still read the table
        int rc = h_packdesc.ensure(sizeof(PackDesc) * pending.size()); if (rc) return rc;
        PackDesc* pd = static_cast<PackDesc*>(h_packdesc.p);
        const float* base = static_cast<const float*>(h_stage.p);
        int n_max = 0, q = 0;
        for (Cloud* c : pending) {
            if (!c->raw || c->n <= 0) { c->raw = nullptr; continue; }
            pd[q].raw_off = (unsigned long long)(c->raw - base); pd[q].dst = c->rec(); pd[q].n = c->n; pd[q].pad_ = 0;   // (a cloud in registered memory: the difference wraps, base + it is the cloud again)
            pd[q].feat_off = c->raw_feat ? (unsigned long long)(c->raw_feat - base) : 0ull; ++q;
            n_max = std::max(n_max, c->n); c->raw = nullptr; c->raw_feat = nullptr;
        }
        pending.clear();
        if (q == 0) return CVO_OK;
        const hipError_t e = launch_pack_clouds(base, pd, q, n_max, s);
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("cloud pack kernel launch: ") + hipGetErrorString(e));
        packdesc_stream = s; add_ring_reader(s);
        return CVO_OK;
    }
    // before the ring is written over: nothing may still have to read it
    int drain_ring() {
        // a launch of this engine on a caller's stream may still read the clouds the pack kernel below writes, or pull clouds from the ring itself
        if (launched && last_stream && last_stream != stream) HIP_TRY(hipStreamSynchronize(last_stream));
        int rc = flush_pending(stream); if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(stream));
        return sync_ring_readers();
    }
    int upload_many(const UploadItem* it, int count) {
        HIP_TRY(hipSetDevice(device));
        int live = 0, n_max = 0; size_t raw_floats = 0;
        for (int k = 0; k < count; ++k) {
            if (!it[k].c) return fail(CVO_ERR_INVALID, "null cloud");
            if (it[k].n < 0) return fail(CVO_ERR_INVALID, "negative point count");
            if (it[k].n > 65535) return fail(CVO_ERR_INVALID, "more than 65535 points per cloud is not supported (16-bit column indices)");
            if (it[k].n > 0 && (!it[k].xyz || !it[k].feat)) return fail(CVO_ERR_INVALID, "null cloud pointer");
            if (it[k].n > 0) { ++live; n_max = std::max(n_max, it[k].n); raw_floats += (size_t)it[k].n * REC; }
        }
        for (int k = 0; k < count; ++k) {
            Cloud& c = *it[k].c;
            c.n = it[k].n; c.boxes_valid = false; c.cost_hint = 0.f;
            if (c.n > 0) { int rc = c.buf.ensure((size_t)c.n * REC * sizeof(float)); if (rc) return rc; }
            if (c.n > 0) {                                          // every 16th point: enough to rank the pairs of a batch (launch_impl)
                double acc = 0; int m = 0;
                for (int i = 0; i < c.n; i += 16) { const float z = it[k].xyz[(size_t)i * 3 + 2]; if (z > 1e-3f) { acc += 1.0 / ((double)z * z); ++m; } }
                c.cost_hint = m ? (float)(acc / m) : 0.f;
            }
        }
        uploads_pending = true;
        if (live == 0) return CVO_OK;
        const size_t desc_bytes = (sizeof(PackDesc) * (size_t)live + 255) & ~(size_t)255;
        const size_t bytes = desc_bytes + raw_floats * sizeof(float);
        const size_t want = std::max(bytes, (size_t)16 << 20);
        int rc;
        if (h_stage.bytes < want) {
            rc = drain_ring(); if (rc) return rc;                    // the old buffer may still feed a copy or a kernel
            rc = h_stage.ensure(want); if (rc) return rc;
            stage_used = 0;
        }
        if (stage_used + bytes > h_stage.bytes) { rc = drain_ring(); if (rc) return rc; stage_used = 0; }
        if (!defer_pack && upload_copy && d_raw.bytes < bytes) { HIP_TRY(hipStreamSynchronize(stream)); rc = d_raw.ensure(bytes); if (rc) return rc; }   // an earlier pack kernel may still read it
        unsigned char* blk = static_cast<unsigned char*>(h_stage.p) + stage_used;
        stage_used += (bytes + 255) & ~(size_t)255;
        PackDesc* pd = reinterpret_cast<PackDesc*>(blk);             // (read by the pack kernel of the single-object path below; a batch's table is made by flush_pending)
        float* raw = reinterpret_cast<float*>(blk + desc_bytes);
        struct Piece { const float* src; float* dst; size_t bytes; };
        std::vector<Piece> pieces; pieces.reserve(2 * (size_t)live);
        size_t off = 0; int q = 0;
        std::vector<char> in_place(count, 0);                          // clouds of a batch that lie in caller-registered memory: read where they are
        for (int k = 0; k < count; ++k) {
            const int n = it[k].n; if (n <= 0) continue;
            in_place[k] = defer_pack && inkernel_pack && !ring_mirror && in_registered_memory(it[k].xyz, sizeof(float) * 3 * (size_t)n) && in_registered_memory(it[k].feat, sizeof(float) * 5 * (size_t)n);
            pd[q].raw_off = off; pd[q].dst = it[k].c->rec(); pd[q].n = n; pd[q].pad_ = 0; pd[q].feat_off = 0; ++q;
            if (!in_place[k]) {
                pieces.push_back(Piece{it[k].xyz, raw + off, sizeof(float) * 3 * (size_t)n});
                pieces.push_back(Piece{it[k].feat, raw + off + 3 * (size_t)n, sizeof(float) * 5 * (size_t)n});
            }
            off += (size_t)n * REC;
        }
        const int nthreads = (bytes >= ((size_t)2 << 20) && !pieces.empty()) ? std::max(1, std::min(upload_threads, (int)pieces.size())) : 1;
        const bool nt = upload_nt && defer_pack;                     // (a single object's block is copied on by the DMA engine right away: let it come from the cache)
        auto copy_range = [&pieces, nt](size_t a, size_t b) { for (size_t i = a; i < b; ++i) ring_copy(pieces[i].dst, pieces[i].src, pieces[i].bytes, nt); };
        if (nthreads == 1) copy_range(0, pieces.size());
        else {                                                        // in 4 x nthreads parts, taken by the pool's threads and the caller as they come free
            const int parts = std::min((int)pieces.size(), 4 * nthreads);
            const size_t per = (pieces.size() + parts - 1) / parts;
            CopyPool::get().run(parts, [&](int t) { copy_range(std::min(pieces.size(), (size_t)t * per), std::min(pieces.size(), (size_t)(t + 1) * per)); });
        }
        // Batches: the clouds stay in the ring as they came; the next align launch packs each pair's clouds itself, reading the ring where
        // it lies (pinned host memory is mapped into the device's address space): no copy-engine transfer and no kernel of its own between
        // the hand-over and the launch -- both cost a tenth of the throughput beside eight persistent launches (copies of different streams
        // share the DMA engines, small kernels wait for a compute unit no persistent workgroup occupies).  Anything else that wants the
        // clouds first (a score block, a cooperative launch) runs the pack kernel on them (flush_pending).
        if (defer_pack) {
            if (ring_mirror && inkernel_pack) {
                if (d_ring.bytes < h_stage.bytes) { HIP_TRY(hipStreamSynchronize(stream)); for (hipStream_t r : ring_readers) if (r != stream) HIP_TRY(hipStreamSynchronize(r)); rc = d_ring.ensure(h_stage.bytes); if (rc) return rc; }
                HIP_TRY(hipMemcpyAsync(static_cast<unsigned char*>(d_ring.p) + (blk - static_cast<unsigned char*>(h_stage.p)), blk, bytes, hipMemcpyHostToDevice, stream));
                if (!ev_ring) HIP_TRY(hipEventCreateWithFlags(&ev_ring, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(ev_ring, stream));
            }
            size_t o2 = 0;
            for (int k = 0; k < count; ++k) {
                const int n = it[k].n; if (n <= 0) continue;
                if (!it[k].c->raw) pending.push_back(it[k].c);
                if (in_place[k]) { it[k].c->raw = it[k].xyz; it[k].c->raw_feat = it[k].feat; }
                else { it[k].c->raw = raw + o2; it[k].c->raw_feat = nullptr; }
                o2 += (size_t)n * REC;
            }
            return CVO_OK;
        }
        // Single objects: one host-to-device copy of the block, one pack kernel.  (CVO_HIP_UPLOAD_COPY=0: the kernel reads the ring itself.)
        const unsigned char* src = blk;
        if (upload_copy) {
            HIP_TRY(hipMemcpyAsync(d_raw.p, blk, bytes, hipMemcpyHostToDevice, stream));
            src = static_cast<const unsigned char*>(d_raw.p);
        }
        const hipError_t e = launch_pack_clouds(reinterpret_cast<const float*>(src + desc_bytes), reinterpret_cast<const PackDesc*>(src), live, n_max, stream);
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("cloud pack kernel launch: ") + hipGetErrorString(e));
        return CVO_OK;
    }
    int upload(Cloud& c, const float* xyz, const float* feat, int n) {
        const UploadItem it{&c, xyz, feat, n};
        return upload_many(&it, 1);
    }

    // ---- pcd_generator on the GPU (cvo_pcd_kernels.hip).  Image-sized scratch lives with the engine.
    DevBuf d_bgr, d_depth, d_I0, d_I1, d_I2, d_dx0, d_dy0, d_abs0, d_abs1, d_abs2, d_ths, d_thsS, d_map, d_pattern, d_counts, d_tiles;
    PinBuf h_counts;
    std::shared_ptr<ImageStage> img_stage;                          // the frame's images as staged for the copy to the device (ImageStage above)
    int pattern_len = 0;
    // glibc srand(seed); rand() & 0xFF, n times (PixelSelector2.cpp:36-38): TYPE_3 additive feedback generator
    static void rand_pattern(unsigned seed, unsigned char* out, size_t n) {
        std::vector<uint32_t> r(344 + n);
        r[0] = seed ? seed : 1;
        for (int i = 1; i < 31; ++i) {
            const long long hi = (int32_t)r[i - 1] / 127773, lo = (int32_t)r[i - 1] % 127773;
            long long word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            r[i] = (uint32_t)word;
        }
        for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
        for (size_t i = 34; i < 344 + n; ++i) r[i] = r[i - 31] + r[i - 3];
        for (size_t k = 0; k < n; ++k) out[k] = (unsigned char)((r[344 + k] >> 1) & 0xFF);
    }
    int generate_pcd(Cloud& c, const unsigned char* bgr8, const unsigned short* depth16, int w, int h, const cvo_camera& cam, int num_want) {
        HIP_TRY(hipSetDevice(device));
        static const bool timing = std::getenv("CVO_HIP_PCD_TIMING") != nullptr;   // development aid: where a generation's host time goes (stderr)
        const auto tq0 = std::chrono::steady_clock::now(); auto tq = tq0; double tms[6] = {0, 0, 0, 0, 0, 0}; int tqi = 0;
        auto lap = [&]() { if (timing && tqi < 6) { const auto now = std::chrono::steady_clock::now(); tms[tqi++] = std::chrono::duration<double, std::micro>(now - tq).count(); tq = now; } };
        if (!bgr8 || !depth16) return fail(CVO_ERR_INVALID, "null image pointer");
        if (w < 64 || h < 64 || (size_t)w * h > (size_t)1 << 26) return fail(CVO_ERR_INVALID, "image size out of range");
        if (num_want <= 0) return fail(CVO_ERR_INVALID, "num_want must be positive");
        const size_t n = (size_t)w * h;
        const int w1 = w / 2, h1 = h / 2, w2 = w1 / 2, h2 = h1 / 2, w32 = w / 32, h32 = h / 32;
        int rc;
        if ((rc = d_bgr.ensure(3 * n))) return rc;
        if ((rc = d_depth.ensure(2 * n))) return rc;
        for (DevBuf* b : {&d_I0, &d_dx0, &d_dy0, &d_abs0}) if ((rc = b->ensure(sizeof(float) * n))) return rc;
        for (DevBuf* b : {&d_I1, &d_abs1}) if ((rc = b->ensure(sizeof(float) * (size_t)w1 * h1))) return rc;
        for (DevBuf* b : {&d_I2, &d_abs2}) if ((rc = b->ensure(sizeof(float) * (size_t)w2 * h2))) return rc;
        const size_t nths = (size_t)w32 * h32 + 100;                    // +100 zeroed slack, read for rows past h/32 (PixelSelector2.cpp:44-45)
        if ((rc = d_ths.ensure(sizeof(float) * nths))) return rc;
        if ((rc = d_thsS.ensure(sizeof(float) * nths))) return rc;
        if ((rc = d_map.ensure(n))) return rc;
        if ((rc = d_counts.ensure(sizeof(int) * 8))) return rc;
        if ((rc = h_counts.ensure(sizeof(int) * 8))) return rc;
        if (!img_stage) img_stage = std::make_shared<ImageStage>();     // (the copies of the previous generation have run: it ended with a synchronize)
        img_stage->stamp.fetch_add(1, std::memory_order_acq_rel);       // whoever compares against the old content sees that it is going
        if ((rc = img_stage->buf.ensure(5 * n))) return rc;
        if (pattern_len != (int)n) {                                    // the byte pattern only depends on w*h: made once
            if ((rc = d_pattern.ensure(n))) return rc;
            std::vector<unsigned char> pat(n);
            rand_pattern(3141592u, pat.data(), n);
            HIP_TRY(hipMemcpy(d_pattern.p, pat.data(), n, hipMemcpyHostToDevice));
            pattern_len = (int)n;
        }
        unsigned char* st = static_cast<unsigned char*>(img_stage->buf.p);
        // every return from here on leaves nothing of this call in flight: the next generation overwrites the pinned stage the copies read from
        struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{stream};
        std::unique_lock<std::mutex> stage_lock(img_stage->mu);
        // the colour image first: its copy and the gradient / threshold kernels (which need nothing else) run while the host stages the depth image
        std::memcpy(st, bgr8, 3 * n);
        HIP_TRY(hipMemcpyAsync(d_bgr.p, st, 3 * n, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_ths.p, 0, sizeof(float) * nths, stream));
        HIP_TRY(hipMemsetAsync(d_thsS.p, 0, sizeof(float) * nths, stream));
        float* I0 = (float*)d_I0.p; float* dx0 = (float*)d_dx0.p; float* dy0 = (float*)d_dy0.p; float* abs0 = (float*)d_abs0.p;
        hipError_t e = pcd_launch_pyramid((const uint8_t*)d_bgr.p, w, h, I0, (float*)d_I1.p, (float*)d_I2.p, dx0, dy0, abs0, (float*)d_abs1.p, (float*)d_abs2.p, stream);
        if (e == hipSuccess) e = pcd_launch_thresholds(abs0, w, h, (float*)d_ths.p, (float*)d_thsS.p, stream);
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("pcd kernels: ") + hipGetErrorString(e));
        lap();                                                          // [0] buffers, colour staging + copy + kernels queued
        std::memcpy(st + 3 * n, depth16, 2 * n);
        stage_lock.unlock();
        HIP_TRY(hipMemcpyAsync(d_depth.p, st + 3 * n, 2 * n, hipMemcpyHostToDevice, stream));
        lap();                                                          // [1] depth staging + copy queued
        // PixelSelector::makeMaps (PixelSelector2.cpp:136-282), a fresh selector per frame: potential 3, one re-selection allowed
        int pot = 3, recursions_left = 1, ideal = 3;
        float num_have = 0, quotia = 0;
        const float num_want_f = (float)num_want;
        int* hc = static_cast<int*>(h_counts.p);
        for (;;) {
            HIP_TRY(hipMemsetAsync(d_map.p, 0, n, stream));
            HIP_TRY(hipMemsetAsync(d_counts.p, 0, sizeof(int) * 8, stream));
            e = pcd_launch_select(abs0, (const float*)d_abs1.p, (const float*)d_abs2.p, (const float*)d_thsS.p, w, h, pot, (uint8_t*)d_map.p, (int*)d_counts.p, stream);
            if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("pcd select: ") + hipGetErrorString(e));
            HIP_TRY(hipMemcpyAsync(hc, d_counts.p, sizeof(int) * 3, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            lap();                                                      // [2] (and [3] after a re-selection) everything up to the selection's counts
            num_have = (float)(hc[0] + hc[1] + hc[2]);                  // :191
            quotia = num_want_f / num_have;                             // :192
            const float K = num_have * (pot + 1) * (pot + 1);           // :195
            ideal = (int)(sqrtf(K / num_want_f) - 1);                   // :196
            if (ideal < 1) ideal = 1;
            if (recursions_left > 0 && quotia > 1.25 && pot > 1) {      // :199-213
                if (ideal >= pot) ideal = pot - 1;
                pot = ideal; --recursions_left; continue;
            }
            if (recursions_left > 0 && quotia < 0.25) {                 // :214-229
                if (ideal <= pot) ideal = pot + 1;
                pot = ideal; --recursions_left; continue;
            }
            break;
        }
        const int subsample = (quotia < 0.95) ? 1 : 0;                  // :252-268
        const int char_th = subsample ? (int)(unsigned char)(255 * quotia) : 255;
        const float camv[5] = {cam.scaling_factor, cam.fx, cam.fy, cam.cx, cam.cy};
        const int nt = pcd_tiles(w, h);
        if ((rc = d_tiles.ensure(sizeof(int) * 3 * (size_t)nt))) return rc;
        if ((rc = h_counts.ensure(sizeof(int) * std::max(8, 3 * nt)))) return rc;
        hc = static_cast<int*>(h_counts.p);
        e = pcd_launch_subsample((uint8_t*)d_map.p, (const uint8_t*)d_pattern.p, subsample, char_th, (const uint16_t*)d_depth.p, w, h, (int*)d_tiles.p, stream);
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("pcd sub-sampling: ") + hipGetErrorString(e));
        HIP_TRY(hipMemcpyAsync(hc, d_tiles.p, sizeof(int) * 3 * (size_t)nt, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        lap();                                                          // sub-sampling + tile counts
        int npts = 0;
        for (int t = 0; t < nt; ++t) npts += hc[nt + t];                 // kept pixels with a valid depth = points (pcd_generator.cpp:471)
        if (npts > 65535) return fail(CVO_ERR_INVALID, "more than 65535 points per cloud is not supported (16-bit column indices)");
        c.n = npts; c.n_px = npts; c.boxes_valid = false;
        if (npts == 0) return CVO_OK;
        if ((rc = c.buf.ensure((size_t)npts * REC * sizeof(float)))) return rc;
        if ((rc = c.px.ensure((size_t)npts * 2 * sizeof(uint16_t)))) return rc;
        e = pcd_launch_cloud((const uint8_t*)d_map.p, (const uint16_t*)d_depth.p, (const uint8_t*)d_bgr.p, dx0, dy0, w, h, camv, (const int*)d_tiles.p, npts, c.rec(),
                             (uint16_t*)c.px.p, stream);
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("pcd cloud: ") + hipGetErrorString(e));
        HIP_TRY(hipStreamSynchronize(stream));
        lap();                                                          // the cloud itself
        if (timing) std::fprintf(stderr, "[cvo_hip] generate_pcd laps (us): %.0f %.0f %.0f %.0f %.0f %.0f, total %.0f\n", tms[0], tms[1], tms[2], tms[3], tms[4], tms[5],
                                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tq0).count());
        return CVO_OK;
    }

    struct PairIn { const Cloud* fixed; const Cloud* moving; };

    int pick_workgroups(int n_pairs, int nf_max) const {
        int G = wg_request;
        if (G <= 0) {
            // lowest latency of this launch alone: as many workgroups per pair as the CUs allow, but not below ~384 rows per
            // workgroup -- a row's list is walked by one lane, so beyond one 64-row block per wave nothing gets shorter while the
            // partial-sum exchanges keep costing (measured at 3072 points: 2.37 ms at 4, 2.33 at 8, 2.44 at 16, 2.78 at 32, 4.9 at 1)
            // (round 5: at most ~384 rows per workgroup, not at least -- a cloud of 2 817 points, what the generator makes of a 640 x 480 frame, took 4 workgroups
            // of 704 rows where 8 of 352 align it in 1.20 ms instead of 1.37: scripts/r05/probe_tracker_g.py)
            const int rows_max = 384 * (align_block_max() / 512);
            const int want = std::max(1, std::min(32, (nf_max + rows_max - 1) / rows_max));
            G = 1; while (G * 2 * n_pairs <= num_cus * per_cu && G < want) G *= 2;
        }
        const int g_min = (((nf_max + 127) / 128) + (MAX_ROWS_PER_WG / 128) - 1) / (MAX_ROWS_PER_WG / 128);   // a workgroup owns at most MAX_ROWS_PER_WG rows, dealt in blocks of 128
        G = std::max(std::max(1, g_min), std::min(G, num_cus));
        return G;
    }

    // One persistent launch aligning n pairs.  states (host, n entries) are uploaded
    // first when upload_states is set; results land in h_states after wait().
    int launch(const std::vector<PairIn>& pairs, const PairState* states_in, bool upload_states, hipStream_t on_stream,
               bool want_trace, int trace_cap) {
        const int rc = launch_impl(pairs, states_in, upload_states, on_stream, want_trace, trace_cap);
        if (rc != CVO_OK) release_slots();                            // nothing of this call is on the device
        return rc;
    }
    int launch_impl(const std::vector<PairIn>& pairs, const PairState* states_in, bool upload_states, hipStream_t on_stream,
                    bool want_trace, int trace_cap) {
        HIP_TRY(hipSetDevice(device));
        const int n = (int)pairs.size();
        if (n <= 0) return fail(CVO_ERR_INVALID, "no pairs to align");
        hipStream_t s = on_stream ? on_stream : stream;
        { int rcs = settle_uploads(s); if (rcs) return rcs; }
        int nf_max = 0, nm_max = 0;
        for (const PairIn& p : pairs) { nf_max = std::max(nf_max, p.fixed ? p.fixed->n : 0); nm_max = std::max(nm_max, p.moving ? p.moving->n : 0); }
        const int g_floor = std::max(1, (((nf_max + 127) / 128) + (MAX_ROWS_PER_WG / 128) - 1) / (MAX_ROWS_PER_WG / 128));
        std::unique_lock<std::mutex> book_lock; std::vector<hipEvent_t> run_after;
        const int G = acquire_slots(pick_workgroups(n, nf_max), g_floor, n, book_lock, run_after);   // fewer workgroups per pair beside other handles' cooperative launches
        last_G = G;
        const int slots = std::max(1, std::min(n, (max_wgs > 0 ? std::min(max_wgs, num_cus * per_cu) : num_cus * per_cu) / G));   // every workgroup of the grid must be resident
        const int grid = slots * G;
        const int nf_pad = round_up(std::max(nf_max, 1), 64), nm_pad = round_up(std::max(nm_max, 1), 64);
        int capf = capf_request;
        // room for the nonzero records: nm/6 per row on average (512 at 3 k points; a fronto-parallel wall half a metre from the
        // camera gives ~300 neighbours per row at the first ell), beyond that the dense per-row fallback takes over
        if (capf <= 0) capf = std::max(128, nm_max / 6);
        // Which build of the kernel runs the launch, and its LDS plan.  Clouds that only fit in the plane layout (9 k points) run with three waves per SIMD:
        // their walks wait for memory more than they issue (+5.8 %; the 3 k-point shape loses 3 % to the waves' uneven shares and stays with two).
        struct KSet { size_t (*shared_bytes)(int, int, int, int, int); int (*min_tile)(int, int, int); int (*tile_granule)(); int (*block_max)();
                      hipError_t (*launch)(int, int, int, int, int, int, int, hipStream_t, const PairDesc*, int, int, unsigned, unsigned long long*, const DevParams&, const unsigned*, unsigned*, const float* const*); };
        static const KSet KS2 = {align_shared_bytes, align_min_tile, align_tile_granule, align_block_max, launch_align};
        static const KSet KS3 = {cvohip_w3::align_shared_bytes, cvohip_w3::align_min_tile, cvohip_w3::align_tile_granule, cvohip_w3::align_block_max, cvohip_w3::launch_align};
        struct Plan { int y_mode = 0, tile = 0, tab_cols = 0, block = 0; bool err = false; };
        const int rows_per_w = ((((std::max(nf_max, 1) + 127) / 128) + G - 1) / G) * 128;   // rows are dealt to the workgroups in blocks of 128 (ROW_DEAL)
        const int rows_cap = round_up(std::max(rows_per_w, 1), 128) + 64;
        auto plan = [&](const KSet& K) -> Plan {
        Plan pl;
        const int tgran = K.tile_granule();
        // LDS budget of one workgroup.  The transformed moving cloud is kept resident if at all possible: as float4 {y, g0}
        // (mode 1) beside a cull tile that holds the whole cloud; else as three float planes (mode 2, 12 B/point) beside a
        // tile just large enough to keep the fixed points in slot order; else it stays in HBM/L2 (mode 0).
        const size_t lds_cap = (size_t)(160 / per_cu - 4) * 1024;
        const int tile_full = std::min(round_up(std::max(nm_max, tgran), tgran), 4096);
        const int tile_rows = round_up(std::max(round_up(std::max(rows_per_w, 1), 64), 512), tgran);   // >= the slots: x_i by slot fits the idle tile
        const bool allow_lds = !std::getenv("CVO_HIP_NO_YLDS");
        int& y_mode = pl.y_mode; int& tile = pl.tile; y_mode = 0; tile = tile_request > 0 ? round_up(tile_request, tgran) : tile_full;
        if (const char* e = std::getenv("CVO_HIP_Y_MODE")) {                                // test knob: force a layout (must fit)
            y_mode = std::max(0, std::min(2, std::atoi(e)));
            if (tile_request <= 0) { tile = y_mode == 1 ? tile_full : std::min(tile_full, std::max(tile_rows, 512)); while (tile > tgran && K.shared_bytes(tile, rows_cap, y_mode, nm_pad, 0) > lds_cap) tile -= tgran; }
            tile = std::max(tile, K.min_tile(rows_cap, y_mode, nm_pad));
            if (K.shared_bytes(tile, rows_cap, y_mode, nm_pad, 0) > lds_cap) { pl.err = true; return pl; }
        } else if (tile_request > 0) {
            if (allow_lds && K.shared_bytes(tile, rows_cap, 1, nm_pad, 0) <= lds_cap) y_mode = 1;
            else if (allow_lds && K.shared_bytes(std::max(tile, K.min_tile(rows_cap, 2, nm_pad)), rows_cap, 2, nm_pad, 0) <= lds_cap) { y_mode = 2; tile = std::max(tile, K.min_tile(rows_cap, 2, nm_pad)); }
            else while (tile > tgran && K.shared_bytes(tile, rows_cap, 0, 0, 0) > lds_cap) tile -= tgran;
        } else {
            int t1 = tile_full; while (t1 > 512 && K.shared_bytes(t1, rows_cap, 1, nm_pad, 0) > lds_cap) t1 -= tgran;
            // plane layout: the tile only holds the fixed points by slot (and the rebuild scratch during a rebuild); it is not used for columns
            const int t2min = K.min_tile(rows_cap, 2, nm_pad);
            int t2 = std::max(std::min(tile_full, tile_rows), t2min); while (t2 > t2min && K.shared_bytes(t2, rows_cap, 2, nm_pad, 0) > lds_cap) t2 -= tgran;
            if (allow_lds && K.shared_bytes(t1, rows_cap, 1, nm_pad, 0) <= lds_cap) { y_mode = 1; tile = t1; }
            else if (allow_lds && K.shared_bytes(t2, rows_cap, 2, nm_pad, 0) <= lds_cap) { y_mode = 2; tile = t2; }
            else { tile = std::min(tile_full, 2048); while (tile > tgran && K.shared_bytes(tile, rows_cap, 0, 0, 0) > lds_cap) tile -= tgran; }
        }
        // line-search table (cvo_kernels.hip, phase L): 16 bytes per moving point behind the resident cloud, when that fits
        int& tab_cols = pl.tab_cols; tab_cols = 0;
        if (!std::getenv("CVO_HIP_NO_TABLE") && y_mode != 0 && K.shared_bytes(tile, rows_cap, y_mode, nm_pad, nm_pad) <= (size_t)(160 / per_cu) * 1024 - 512) tab_cols = nm_pad;
        const int bmax = K.block_max();
        int& block = pl.block; block = rows_per_w > bmax / 2 ? bmax : std::max(64, round_up(rows_per_w, 64));
        if (per_cu > 1) block = std::min(block, 256);
        if (block_request > 0) block = std::max(64, std::min(bmax, round_up(block_request, 64)));

        return pl;
        };
        const KSet* K = &KS2;
        Plan pl = plan(KS2);
        if (pl.err) return fail(CVO_ERR_INVALID, "CVO_HIP_Y_MODE: the requested LDS layout does not fit");
        if ((pl.y_mode == 2 || (wide_all && pl.y_mode == 1)) && wide_waves && per_cu == 1) {
            const Plan p3 = plan(KS3);
            if (!p3.err && p3.y_mode == pl.y_mode) { pl = p3; K = &KS3; }
        }
        const int y_mode = pl.y_mode, tile = pl.tile, tab_cols = pl.tab_cols, block = pl.block, rows_per = rows_per_w;

        int rc;
        if ((rc = d_descs.ensure(sizeof(PairDesc) * n))) return rc;
        if ((rc = h_descs.ensure(sizeof(PairDesc) * n))) return rc;
        if ((rc = d_states.ensure(sizeof(PairState) * n))) return rc;
        if ((rc = d_records.grow_keep(sizeof(float) * CVO_RESULT_FLOATS * (size_t)std::max(n, rec_hint), 0, s))) return rc;
        if (n > pad_from) pad_from = pad_to = 0;                     // this launch writes over (some of) the padding records
        if ((rc = h_states.ensure(sizeof(PairState) * n))) return rc;
        if ((rc = h_states_in.ensure(sizeof(PairState) * n))) return rc;
        if (2 * (long long)P.max_iter + 8 >= 65536) return fail(CVO_ERR_INVALID, "max_iter must be below 32764");
        if ((rc = d_ybuf.ensure(sizeof(float4) * (size_t)slots * G * nm_pad))) return rc;   // work buffers: per pair SLOT of the launch
        // Adoption (a pair can grow to four workgroups while it runs): only for one workgroup and one slot per pair with the cloud resident
        // in LDS.  Every member keeps its lists and records in a region of its own, sized for the rows it owns when it joins (make_ctx:
        // 1 + 1/2 + 1/3 + 1/4 of the rows, each rounded up to blocks of 128), and the exchange area has room for four members
        // (cvo_kernels.hip: ADOPT_GMAX).
        AdoptCounters* const qc = adopt_counters(device);                   // what is queued on the device: every launch counts (null: the counters could not be made)
        AdoptCounters* const ac = (adopt && G == 1 && slots == n && y_mode != 0) ? qc : nullptr;
        const int gmax = align_adopt_gmax();                                 // the kernel's limit (ADOPT_GMAX)
        const int Gx = ac ? gmax : G;                                        // members a pair's exchange area has room for
        size_t member_rows = 0;                                              // rows of all member regions of a slot under adoption
        if (ac) { const int nbk = (std::max(nf_max, 1) + 127) / 128; for (int q = 1; q <= gmax; ++q) member_rows += (size_t)((nbk + q - 1) / q) * 128; }
        const size_t plane = ac ? member_rows * capf : (size_t)G * rows_per * capf;   // workgroup g's nonzero records start at g * rows_per * capf
        const int rows_pad = round_up(std::max(rows_per, 1), 128);          // the cull walks pairs of 64-row blocks
        int capn = 64; while (capn < nm_max / 3 && capn < 4096) capn *= 2;   // longest list a row may have: 1024 at 3 k points, 4096 at 10 k
        if (const char* e = std::getenv("CVO_HIP_ROW_CAP")) capn = std::max(1, std::atoi(e));
        capn = std::max(8, round_up(capn, 4));                               // the candidate phase reads entries four at a time, one step ahead
        const size_t tplane = ac ? member_rows * capn : (size_t)G * capn * rows_pad;
        if ((rc = d_jT.ensure(sizeof(uint16_t) * (size_t)slots * tplane))) return rc;
        if ((rc = d_ent.ensure(sizeof(uint2) * (size_t)slots * tplane))) return rc;
        if ((rc = d_surv.ensure(sizeof(uint2) * (size_t)slots * plane))) return rc;
        const size_t xch_bytes = sizeof(unsigned long long) * (size_t)n * 2 * Gx * XCH_WORDS;
        if ((rc = d_xch.ensure(xch_bytes))) return rc;
        if (want_trace) {
            if ((rc = d_trace.ensure(sizeof(TraceRow) * (size_t)std::max(1, trace_cap)))) return rc;
            if ((rc = d_tracelen.ensure(sizeof(int) * 4))) return rc;
            HIP_TRY(hipMemsetAsync(d_tracelen.p, 0, sizeof(int) * 4, s));
        }
        // the tracker's score block in the kernel's tail: the clouds' tables of cached self inner products live behind their group boxes
        bool tails = tail_scores && !want_trace;
        for (int i = 0; i < n && tails; ++i) tails = pairs[i].fixed && pairs[i].moving && pairs[i].fixed->n > 0 && pairs[i].moving->n > 0;
        // clouds handed over since the last launch: packed by this launch's workgroups (one per pair, nothing that needs the clouds before the
        // kernel runs), else by the pack kernel now
        const float** rawtab = nullptr;
        if (!pending.empty()) {
            if (inkernel_pack && G == 1 && !tails) {
                if (launched && last_stream) HIP_TRY(hipStreamSynchronize(last_stream));   // an earlier launch of this engine may still read the table
                if ((rc = h_rawtab.ensure(sizeof(const float*) * 4 * (size_t)n))) return rc;
                rawtab = static_cast<const float**>(h_rawtab.p);
                // (the clouds' raw pointers are host addresses in the ring; with the mirror the kernel gets the same offsets in the device copy)
                const bool mir = ring_mirror && d_ring.p && d_ring.bytes >= h_stage.bytes && ev_ring;
                const ptrdiff_t shift = mir ? (static_cast<const unsigned char*>(d_ring.p) - static_cast<const unsigned char*>(h_stage.p)) : 0;
                auto dev = [&](const float* r) -> const float* { return r ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(r) + shift) : nullptr; };
                for (int i = 0; i < n; ++i) {                        // {positions, features (null: right behind the positions)} x {fixed, moving}
                    const Cloud* cf = pairs[i].fixed; const Cloud* cm = pairs[i].moving;
                    rawtab[4 * i] = dev(cf ? cf->raw : nullptr); rawtab[4 * i + 1] = (cf && cf->raw) ? cf->raw_feat : nullptr;
                    rawtab[4 * i + 2] = dev(cm ? cm->raw : nullptr); rawtab[4 * i + 3] = (cm && cm->raw) ? cm->raw_feat : nullptr;
                }
                if (mir && s != stream) HIP_TRY(hipStreamWaitEvent(s, ev_ring, 0));
            } else if ((rc = flush_pending(s))) return rc;
        }
        if (tails) {
            if ((rc = h_tail.ensure(sizeof(double) * (size_t)n * 5 * 24))) return rc;
            for (int i = 0; i < n; ++i) { if ((rc = ensure_boxes(*pairs[i].fixed, s))) return rc; if ((rc = ensure_boxes(*pairs[i].moving, s))) return rc; }
        }
        std::vector<PairDesc> hdv(n);
        std::memset(hdv.data(), 0, sizeof(PairDesc) * n);            // padding bytes take part in the comparison below
        PairDesc* hd = hdv.data();
        // Which pair a position of the launch (a slot; a pull from the pair queue) works on.  A pair's time is its cost per iteration -- which
        // follows the number of neighbours, i.e. the density of its clouds: mean 1/z^2 explains 0.9 of it on the bench's pairs, the iteration
        // count next to nothing -- so the pairs are ranked by that (Cloud::cost_hint, a sample of the points taken at the hand-over), and
        //  * a launch with a slot per pair gives the positions b, b + 8, b + 16, ... -- workgroups that share an XCD: block p of a launch runs on XCD
        //    (h + p) mod 8, h fixed per stream and different from stream to stream (scripts/micro/xcc_map.hip) -- pairs of SIMILAR density: the c-th
        //    eighth of the ranking goes to the positions = c (mod 8).  With eight streams the launches then form a Latin square over the XCDs: every
        //    XCD hosts every density class, each from another stream, eight pairs of one class at a time -- equal work per XCD, and the eight CUs a
        //    class leaves come free together for the next launch's eight blocks on that XCD.  Measured: +5.3 ... +6.4 % over index order on four sets
        //    of 64 pairs, +11 % with 512 different pairs in flight; evenly mixed orders (densest first, eight heavy / eight light) lose 2 ... 13 %,
        //    2 / 4 / 16 / 32 classes instead of 8 lose 4 / 2 / 13 / 22 %, and launches that disagree on the classes' positions lose 35 %
        //    (profiles/r03_pair_order_*.txt, r03_distinct_sets_ab.txt, r03_class_count_sweep.txt, r03_class_shift_across_handles.txt);
        //  * a launch with fewer slots than pairs hands them out densest first: the long pairs start first and the launch's last ones are
        //    short (+5 ... 13 % on the config-5 shape, 64 pairs on 10 slots).
        // Results stay indexed by pair.  CVO_HIP_ORDER_PAIRS: 0 index order, 1 this rule (default); 2 ... experiment orders (below).
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        if (order_pairs && n > 1) {
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                const float ca = (pairs[a].fixed ? pairs[a].fixed->cost_hint : 0.f) + (pairs[a].moving ? pairs[a].moving->cost_hint : 0.f);
                const float cb = (pairs[b].fixed ? pairs[b].fixed->cost_hint : 0.f) + (pairs[b].moving ? pairs[b].moving->cost_hint : 0.f);
                return ca > cb; });
            const bool queue_launch = slots < n;
            // The XCD-class rule presumes that the positions = c (mod 8) of the launch are the blocks = c (mod 8): one workgroup per pair, or a pair's workgroups
            // co-located on the blocks of one class (DevParams::colocate with the slots a multiple of eight).  A cooperative launch with consecutive blocks per pair
            // spreads every pair over several XCDs whatever its position: densest first there, like the queue launches.  The density ranking itself (Cloud::cost_hint,
            // the mean of 1/z^2 over a sample of the points) presumes clouds in the camera frame with +z the viewing direction, as pcd_generator makes them
            // (pcd_generator.cpp:471-476); clouds in another frame get an arbitrary but harmless ranking -- CVO_HIP_ORDER_PAIRS=0 keeps index order.
            const bool class_is_xcd = G == 1 || (P.colocate && (slots & 7) == 0);
            const int mode = order_mode == 1 ? ((queue_launch || !class_is_xcd) ? 2 : 5) : order_mode;      // 2 densest first, 5 eighths of the ranking per XCD class
            if (mode == 3) std::reverse(order.begin(), order.end());                      // lightest first
            if (mode == 4) { std::vector<int> o2; for (int i = 0, j = n - 1; i <= j; ++i, --j) { o2.push_back(order[i]); if (i != j) o2.push_back(order[j]); } order = o2; }   // heavy, light, heavy, ...
            if (mode == 5 || mode == 10 || mode == 11) {                                   // position p <- chunk p % 8 of the ranking, its (p / 8)-th pair
                std::vector<int> first(9, 0), o2(n);
                for (int c = 0; c < 8; ++c) first[c + 1] = first[c] + n / 8 + ((mode == 11 ? 7 - c : c) < n % 8 ? 1 : 0);   // a chunk has as many pairs as its class has positions
                for (int p = 0; p < n; ++p) {
                    const int c = mode == 11 ? 7 - p % 8 : p % 8, len = first[c + 1] - first[c];
                    const int k = p / 8;                                                  // experiment orders: 10 lightest of the chunk first, 11 chunks in reverse
                    o2[p] = order[first[c] + (mode == 10 ? len - 1 - k : k)];
                }
                order = o2;
            }
            if (mode >= 20 && mode < 100) {                                                // experiment: K = mode - 20 classes instead of 8 (position p <- chunk p % K, its (p / K)-th pair)
                const int Kc = std::max(1, std::min(n, mode - 20));
                std::vector<int> first(Kc + 1, 0), o2(n);
                for (int c = 0; c < Kc; ++c) first[c + 1] = first[c] + n / Kc + (c < n % Kc ? 1 : 0);
                for (int p = 0; p < n; ++p) o2[p] = order[first[p % Kc] + p / Kc];
                order = o2;
            }
            if (mode == 6) { std::vector<int> o2; const int nb8 = (n + 7) / 8; for (int i = 0, j = nb8 - 1; i <= j; ++i, --j) { for (int q = 8 * i; q < std::min(n, 8 * i + 8); ++q) o2.push_back(order[q]); if (i != j) for (int q = 8 * j; q < std::min(n, 8 * j + 8); ++q) o2.push_back(order[q]); } order = o2; }   // eight heavy, eight light, ...
            if (mode >= 7 && mode < 10) { unsigned st = 12345u * (unsigned)mode; for (int i = n - 1; i > 0; --i) { st = st * 1664525u + 1013904223u; std::swap(order[i], order[(st >> 8) % (unsigned)(i + 1)]); } }   // shuffles
        }
        for (int i = 0; i < n; ++i) {
            PairDesc& D = hd[i];
            D.fixed = pairs[i].fixed ? pairs[i].fixed->rec() : nullptr;
            D.moving = pairs[i].moving ? pairs[i].moving->rec() : nullptr;
            D.nf = pairs[i].fixed ? pairs[i].fixed->n : 0;
            D.nm = pairs[i].moving ? pairs[i].moving->n : 0;
            D.nf_pad = nf_pad; D.rows_pad = rows_pad; D.capf = capf; D.nm_pad = nm_pad;
            D.ybuf = static_cast<float4*>(d_ybuf.p);
            D.ws_y_stride = (long long)G * nm_pad; D.ws_list_stride = (long long)tplane; D.ws_surv_stride = (long long)plane;
            D.capn = capn;
            D.jT = static_cast<uint16_t*>(d_jT.p);
            D.ent = static_cast<uint2*>(d_ent.p);
            D.surv = static_cast<uint2*>(d_surv.p);
            D.xch = static_cast<unsigned long long*>(d_xch.p) + (size_t)i * 2 * Gx * XCH_WORDS;
            D.state = static_cast<PairState*>(d_states.p) + i;
            D.state_in = upload_states ? static_cast<const PairState*>(h_states_in.p) + i : D.state;
            D.state_host = static_cast<PairState*>(h_states.p) + i;
            D.trace = (want_trace && i == 0) ? static_cast<TraceRow*>(d_trace.p) : nullptr;
            D.trace_cap = want_trace ? trace_cap : 0;
            D.trace_len = want_trace ? static_cast<int*>(d_tracelen.p) : nullptr;
            D.member_regions = ac ? 1 : 0;
            D.run_pair = order[i];
            D.record = static_cast<float*>(d_records.p) + (size_t)i * CVO_RESULT_FLOATS;
            if (tails) {
                D.score_out = static_cast<double*>(h_tail.p) + (size_t)i * 5 * 24;
                D.self_fixed = score_self_cache(static_cast<float*>(pairs[i].fixed->boxes.p), pairs[i].fixed->n);
                D.self_moving = score_self_cache(static_cast<float*>(pairs[i].moving->boxes.p), pairs[i].moving->n);
            }
        }
        // Steady state = no copy-engine work at all: copies queued on different streams share the DMA engines and a copy behind
        // another stream's running kernel would serialise the launches.  Descriptors go up only when they changed, start states
        // are read by the kernel from pinned host memory, final states are written by it to pinned host memory, and the
        // exchange area is cleared only when it is new (tags carry the launch number).
        if (descs_uploaded.size() != sizeof(PairDesc) * n || std::memcmp(descs_uploaded.data(), hd, sizeof(PairDesc) * n) != 0) {
            HIP_TRY(hipStreamSynchronize(s));                         // an earlier launch on this stream may still read d_descs / h_descs
            std::memcpy(h_descs.p, hd, sizeof(PairDesc) * n);
            HIP_TRY(hipMemcpyAsync(d_descs.p, h_descs.p, sizeof(PairDesc) * n, hipMemcpyHostToDevice, s));
            descs_uploaded.assign(reinterpret_cast<const unsigned char*>(hd), reinterpret_cast<const unsigned char*>(hd) + sizeof(PairDesc) * n);
        }
        if (upload_states) {
            if (launched) HIP_TRY(hipStreamSynchronize(last_stream));  // a queued launch of this engine may not have read its start states yet
            std::memcpy(h_states_in.p, states_in, sizeof(PairState) * n);
        }
        launch_seq = (launch_seq + 1) & 0xFFFFu;
        {   // pair queue of launches with fewer slots than pairs: head word + one mailbox per slot; tags carry the launch number
            const size_t qbytes = sizeof(unsigned long long) * (size_t)(1 + 2 * num_cus * per_cu);   // head, a mailbox / adoption word per slot, a control word per slot
            const bool fresh = d_queue.bytes < qbytes;
            if ((rc = d_queue.ensure(qbytes))) return rc;
            if (fresh || launch_seq == 0) HIP_TRY(hipMemsetAsync(d_queue.p, 0, d_queue.bytes, s));
        }
        if (Gx > 1 && (xch_zeroed_bytes != d_xch.bytes || launch_seq == 0)) {
            HIP_TRY(hipMemsetAsync(d_xch.p, 0, d_xch.bytes, s));
            xch_zeroed_bytes = d_xch.bytes;
        }
        for (hipEvent_t ev : run_after) HIP_TRY(hipStreamWaitEvent(s, ev, 0));
        HIP_TRY(hipEventRecord(ev0, s));
        hipError_t e;
        DevParams Pl = P;
        // The list margin: a moving point may travel skin * r + skin_alpha * (its distance from the camera) before the lists are stale (DevParams::skin_alpha).  Round 3 had one
        // margin for all rows (0.35 r / 0.30 r by layout); with the depth-proportional part the constant part shrinks to what a translation needs, the lists of the near (dense) rows
        // get shorter and the pairs cull 2.0-2.3 times instead of 3.3-3.8: +10 % at 3 k points, +6 % at 9 k (profiles/r04_list_margin_sweep.txt; other motion mixes: r04_list_margin_motion.txt)
        if (skin_auto && alpha_auto) { Pl.skin = y_mode == 1 ? 0.05f : 0.15f; Pl.skin_alpha = y_mode == 1 ? 0.0125f : 0.01f; if (!gamma_set) Pl.alpha_gamma = y_mode == 1 ? 1.0f : 0.f; }
        else if (skin_auto) Pl.skin = y_mode == 1 ? 0.35f : 0.30f;   // CVO_HIP_SKIN_ALPHA given alone: the constant part as round 3 had it
        // Wider FIRST lists for the pairs of a BATCH on four or more cooperating workgroups each (a handful of loop-closure candidates, keyframe_graph.cpp:693-717: few rows per
        // workgroup, a cull costs what it always did, and the frames are far apart): 1.75 x the margin, most pairs then cull once instead of twice -- 10 candidates 2.53 -> 2.35 ms, one
        // of the bench's pairs alone -2.2 %.  Not for single handles: a tracker's consecutive frames cull once either way and the wider lists ADD 5 % to their alignments
        // (profiles/r04_single_pair_phases.txt).  CVO_HIP_FIRST_SCALE sets it for every launch.
        if (coop_first_scale > 0.f && !first_scale_set && G >= 4 && y_mode == 1) Pl.first_scale = coop_first_scale;
        Pl.adopt_on = ac ? 1 : 0;
        if (qc) {                                                     // count the workgroups as submitted, then submit them: in that order, under one lock per process
            std::lock_guard<std::mutex> lk(adopt_submit_mutex());
            *qc->submitted_host += (unsigned)grid;
            e = K->launch(grid, block, tile, rows_cap, y_mode, nm_pad, tab_cols, s, static_cast<const PairDesc*>(d_descs.p), n, G, launch_seq << 16, static_cast<unsigned long long*>(d_queue.p), Pl,
                             qc->submitted_dev, qc->started_dev, rawtab);
            if (e != hipSuccess) resync_adopt_counters(qc);
        } else {
            e = K->launch(grid, block, tile, rows_cap, y_mode, nm_pad, tab_cols, s, static_cast<const PairDesc*>(d_descs.p), n, G, launch_seq << 16, static_cast<unsigned long long*>(d_queue.p), Pl, nullptr, nullptr, rawtab);
        }
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("align kernel launch: ") + hipGetErrorString(e));
        HIP_TRY(hipEventRecord(ev1, s));
        launched = true;
        last_stream = s;
        last_tail = tails;
        if (rawtab) {                                                 // those clouds are the kernel's now; the ring has to stay as it is until the launch is over
            for (int i = 0; i < n; ++i) { if (pairs[i].fixed) { pairs[i].fixed->raw = nullptr; pairs[i].fixed->raw_feat = nullptr; } if (pairs[i].moving) { pairs[i].moving->raw = nullptr; pairs[i].moving->raw_feat = nullptr; } }
            pending.erase(std::remove_if(pending.begin(), pending.end(), [](Cloud* c) { return c->raw == nullptr; }), pending.end());
            add_ring_reader(s);
        }
        return CVO_OK;
    }
    hipStream_t last_stream = nullptr;

    // The block of n_block records this engine contributes to a cross-GPU gather, complete on *s_out in stream order: records
    // [0, n_valid) are the last launch's (written by the align kernel), [n_valid, n_block) stand for no pair and carry CVO_ERR_PADDING --
    // blocks differ by one pair between ranks when the pairs do not divide evenly (cvo_shard_range), the collective needs equal counts.
    // launch_status != CVO_OK: this rank's launch could not be made; all n_block records carry that status, so the rank still takes part
    // in the collective and every rank learns of the failure instead of waiting for it forever.
    int padded_records(int n_valid, int n_block, int launch_status, int last_n, hipStream_t* s_out, float** send) {
        HIP_TRY(hipSetDevice(device));
        if (n_block <= 0 || n_valid < 0 || n_valid > n_block) return fail(CVO_ERR_INVALID, "gather: need 0 <= n_valid <= n_block, n_block > 0");
        if (launch_status == CVO_OK && n_valid > 0 && (!launched || n_valid > last_n)) return fail(CVO_ERR_INVALID, "gather: more records than the last launch aligned");
        hipStream_t s = launched ? last_stream : stream;
        const int keep = launch_status == CVO_OK ? n_valid : 0;
        int rc = d_records.grow_keep(sizeof(float) * CVO_RESULT_FLOATS * (size_t)n_block, sizeof(float) * CVO_RESULT_FLOATS * (size_t)keep, s); if (rc) return rc;
        const int from = keep, status = launch_status == CVO_OK ? CVO_ERR_PADDING : launch_status;
        if (from < n_block && !(pad_from == from && pad_to >= n_block && pad_status == status)) {
            const hipError_t e = launch_fill_records(static_cast<float*>(d_records.p), from, n_block, status, s);
            if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("record fill kernel launch: ") + hipGetErrorString(e));
            pad_from = from; pad_to = n_block; pad_status = status;
        }
        *s_out = s; *send = static_cast<float*>(d_records.p);
        return CVO_OK;
    }

    int wait() {
        HIP_TRY(hipSetDevice(device));
        if (!launched) return fail(CVO_ERR_INVALID, "no launch to wait for");
        const hipError_t es = hipStreamSynchronize(last_stream);
        release_slots();                                             // the launch has left the device, whatever it returned
        if (es != hipSuccess) return fail(CVO_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(es));
        HIP_TRY(hipEventElapsedTime(&last_ms, ev0, ev1));
        return CVO_OK;
    }
    // the align kernel has finished (its states are in h_states, which it writes itself); work queued behind it on the stream may still run
    int wait_align_only() {
        HIP_TRY(hipSetDevice(device));
        if (!launched) return fail(CVO_ERR_INVALID, "no launch to wait for");
        const hipError_t es = hipEventSynchronize(ev1);
        release_slots();
        if (es != hipSuccess) return fail(CVO_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(es));
        HIP_TRY(hipEventElapsedTime(&last_ms, ev0, ev1));
        return CVO_OK;
    }
    const PairState* results() const { return static_cast<const PairState*>(h_states.p); }

    // function_inner_product / se3_Hessian: out[0]=sum_A, out[1]=count, out[2..22]=Hessian terms
    // A score block: any number of function_inner_product / se3_Hessian evaluations in one launch.
    // out[r][0] = sum_A, out[r][1] = pair count, out[r][2..22] = Hessian terms.
    struct ScoreReq {
        const Cloud* a; const float* tran; const Cloud* b; bool hessian; float ell;
        int from = -1;                    // >= 0: ell comes from pair `from`'s device-resident state (the align() result, Q1) ...
        bool tran_from_state = false;     // ... and so does the transform applied to cloud a
    };
    DevBuf d_scoredescs; PinBuf h_scoredescs;
    bool self_cache_on = true;        // CVO_HIP_SELF_CACHE=0: every fip(cloud, cloud) is swept again (tests compare the two)
    std::vector<unsigned char> scoredescs_uploaded;
    hipStream_t score_stream = nullptr; int score_pending = 0;
    size_t stage_used = 0;            // bytes of the pinned staging ring handed to copies that may still be in flight
    bool uploads_pending = false;     // clouds were written on `stream`: a launch on another stream waits for them once
    int settle_uploads(hipStream_t s) {
        if (uploads_pending && s != stream) HIP_TRY(hipStreamSynchronize(stream));
        uploads_pending = false;
        return CVO_OK;
    }
    int ensure_boxes(const Cloud& c, hipStream_t s) {
        if (!c.boxes_valid) {
            int rc = c.boxes.ensure(score_box_bytes(c.n)); if (rc) return rc;
            hipError_t e = launch_cloud_boxes(c.rec(), c.n, static_cast<float*>(c.boxes.p), s);
            if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("cloud box kernel launch: ") + hipGetErrorString(e));
            c.boxes_valid = true; c.boxes_stream = s;
        } else if (c.boxes_stream != s) {
            HIP_TRY(hipStreamSynchronize(c.boxes_stream));           // made on another stream of this engine: complete before use here
            c.boxes_stream = s;
        }
        return CVO_OK;
    }
    // Queue the launch on stream s; the sums land in pinned memory when s has drained (score_collect).
    int score_enqueue(const ScoreReq* rq, int n, hipStream_t s) {
        HIP_TRY(hipSetDevice(device));
        if (n <= 0) return fail(CVO_ERR_INVALID, "bad score request count");
        { int rcs = flush_pending(s); if (rcs) return rcs; }
        { int rcs = settle_uploads(s); if (rcs) return rcs; }
        std::vector<ScoreDesc> descs(n);
        int row_blocks = 1;
        for (int r = 0; r < n; ++r) {
            if (!rq[r].a || !rq[r].b || rq[r].a->n <= 0 || rq[r].b->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "empty cloud");
            ScoreDesc& D = descs[r];
            std::memset(&D, 0, sizeof(D));
            int rcb = ensure_boxes(*rq[r].b, s); if (rcb) return rcb;
            D.a = rq[r].a->rec(); D.b = rq[r].b->rec(); D.na = rq[r].a->n; D.nb = rq[r].b->n; D.ell = rq[r].ell;
            D.bbox = static_cast<const float*>(rq[r].b->boxes.p); D.nbox = score_groups(D.nb);
            D.want_hessian = rq[r].hessian ? 1 : 0; D.out = nullptr;
            // fip(cloud, cloud), untransformed: a function of the cloud and ell alone (cvo.cpp:496-497), kept with the cloud
            if (self_cache_on && rq[r].a == rq[r].b && !rq[r].hessian && !rq[r].tran && !rq[r].tran_from_state)
                D.self_cache = score_self_cache(static_cast<float*>(rq[r].b->boxes.p), rq[r].b->n);
            D.from = rq[r].from >= 0 ? static_cast<const PairState*>(d_states.p) + rq[r].from : nullptr;
            if (rq[r].tran_from_state && !D.from) return fail(CVO_ERR_INVALID, "score request: transform from a state that is not named");
            D.use_tran = rq[r].tran_from_state ? 2 : (rq[r].tran ? 1 : 0);
            for (int i = 0; i < 12; ++i) D.tran[i] = (rq[r].tran && !rq[r].tran_from_state) ? rq[r].tran[i] : 0.f;
            row_blocks = std::max(row_blocks, score_row_blocks(D.na));
        }
        const int nout = score_nout();
        const int chunks = std::max(1, std::min(16, 4 * num_cus / std::max(1, row_blocks * n)));   // ~4 workgroups (waves) per CU
        int rc;
        if ((rc = d_partials.ensure(sizeof(double) * (size_t)n * row_blocks * chunks * nout))) return rc;
        if ((rc = h_partials.ensure(sizeof(double) * (size_t)std::max(n, SCORE_MAXREQ) * nout))) return rc;
        ScoreBatch B; std::memset(&B, 0, sizeof(B));
        const ScoreDesc* more = nullptr;
        if (n <= SCORE_MAXREQ) { B.n = n; for (int r = 0; r < n; ++r) B.d[r] = descs[r]; }
        else {
            // like the align descriptors: uploaded only when they changed, so a steady stream of score blocks over the same pairs
            // (transforms taken from the device-resident states) queues no copy
            const size_t bytes = sizeof(ScoreDesc) * (size_t)n;
            if (scoredescs_uploaded.size() != bytes || std::memcmp(scoredescs_uploaded.data(), descs.data(), bytes) != 0) {
                if ((rc = d_scoredescs.ensure(bytes))) return rc;
                if ((rc = h_scoredescs.ensure(bytes))) return rc;
                HIP_TRY(hipStreamSynchronize(s));
                std::memcpy(h_scoredescs.p, descs.data(), bytes);
                HIP_TRY(hipMemcpyAsync(d_scoredescs.p, h_scoredescs.p, bytes, hipMemcpyHostToDevice, s));
                scoredescs_uploaded.assign(reinterpret_cast<const unsigned char*>(descs.data()), reinterpret_cast<const unsigned char*>(descs.data()) + bytes);
            }
            more = static_cast<const ScoreDesc*>(d_scoredescs.p);
        }
        hipError_t e;
        if (AdoptCounters* const qc = adopt_counters(device)) {      // the score workgroups are queued work too: helpers of align launches in flight hold back for them
            std::lock_guard<std::mutex> lk(adopt_submit_mutex());
            const unsigned wgs = (unsigned)row_blocks * (unsigned)chunks * (unsigned)n;
            *qc->submitted_host += wgs;
            bool sweep_submitted = false;
            e = launch_score(B, more, n, row_blocks, chunks, P, static_cast<double*>(d_partials.p), static_cast<double*>(h_partials.p), s, qc->started_dev, &sweep_submitted);
            if (!sweep_submitted) *qc->submitted_host -= wgs;       // nothing of it will start (the sweep was never handed to the runtime)
            else if (e != hipSuccess) resync_adopt_counters(qc);
        } else e = launch_score(B, more, n, row_blocks, chunks, P, static_cast<double*>(d_partials.p), static_cast<double*>(h_partials.p), s, nullptr, nullptr);
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("score kernel launch: ") + hipGetErrorString(e));
        score_stream = s; score_pending = n;
        return CVO_OK;
    }
    int score_collect(int n, double (*out)[24]) {
        HIP_TRY(hipSetDevice(device));
        if (n <= 0 || n != score_pending) return fail(CVO_ERR_INVALID, "no queued score block of that size");
        HIP_TRY(hipStreamSynchronize(score_stream));
        const int nout = score_nout();
        const double* hp = static_cast<const double*>(h_partials.p);
        for (int r = 0; r < n; ++r) for (int q = 0; q < 24; ++q) out[r][q] = q < nout ? hp[(size_t)r * nout + q] : 0.0;
        score_pending = 0;
        return CVO_OK;
    }
    int score_many(const ScoreReq* rq, int n, double (*out)[24]) {
        int rc = score_enqueue(rq, n, stream); if (rc) return rc;
        return score_collect(n, out);
    }
};

// ---- host-side pieces of the reference's state helpers ----------------------
struct Aff { float m[12]; };
Aff aff_identity() { Aff a; std::memset(a.m, 0, sizeof(a.m)); a.m[0] = a.m[5] = a.m[10] = 1.f; return a; }
Aff aff_mul(const Aff& a, const Aff& b) {   // Affine3f * Affine3f
    Aff c;
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k)
            c.m[r * 4 + k] = (a.m[r * 4 + 0] * b.m[0 * 4 + k] + a.m[r * 4 + 1] * b.m[1 * 4 + k]) + (a.m[r * 4 + 2] * b.m[2 * 4 + k] + a.m[r * 4 + 3] * 0.f);
        c.m[r * 4 + 3] = (a.m[r * 4 + 0] * b.m[0 * 4 + 3] + a.m[r * 4 + 1] * b.m[1 * 4 + 3]) + (a.m[r * 4 + 2] * b.m[2 * 4 + 3] + a.m[r * 4 + 3] * 1.f);
    }
    return c;
}
Aff aff_inverse(const Aff& a) {             // Affine3f::inverse(), Affine mode: cofactor inverse of the linear part
    float L[9]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L[r * 3 + c] = a.m[r * 4 + c];
    const float c00 = L[4] * L[8] - L[5] * L[7], c01 = L[5] * L[6] - L[3] * L[8], c02 = L[3] * L[7] - L[4] * L[6];
    const float det = sum3f(L[0] * c00, L[1] * c01, L[2] * c02), id = 1.f / det;
    float Li[9];
    Li[0] = c00 * id; Li[1] = (L[2] * L[7] - L[1] * L[8]) * id; Li[2] = (L[1] * L[5] - L[2] * L[4]) * id;
    Li[3] = c01 * id; Li[4] = (L[0] * L[8] - L[2] * L[6]) * id; Li[5] = (L[2] * L[3] - L[0] * L[5]) * id;
    Li[6] = c02 * id; Li[7] = (L[1] * L[6] - L[0] * L[7]) * id; Li[8] = (L[0] * L[4] - L[1] * L[3]) * id;
    const float t[3] = {a.m[3], a.m[7], a.m[11]}; float nt[3];
    mat3_vec(Li, t, nt);
    Aff r;
    for (int i = 0; i < 3; ++i) { for (int k = 0; k < 3; ++k) r.m[i * 4 + k] = Li[i * 3 + k]; r.m[i * 4 + 3] = -nt[i]; }
    return r;
}
// Affine3f::rotation(): orthogonal polar factor of the linear part (Eigen uses an
// SVD); Newton iteration X <- (X + X^-T)/2 in double reaches the same matrix.
void polar_rotation(const float* L, float* Rout) {
    double X[9]; for (int i = 0; i < 9; ++i) X[i] = L[i];
    for (int it = 0; it < 32; ++it) {
        const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
        const double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
        double inv[9];
        inv[0] = c00 / det; inv[1] = (X[2] * X[7] - X[1] * X[8]) / det; inv[2] = (X[1] * X[5] - X[2] * X[4]) / det;
        inv[3] = c01 / det; inv[4] = (X[0] * X[8] - X[2] * X[6]) / det; inv[5] = (X[2] * X[3] - X[0] * X[5]) / det;
        inv[6] = c02 / det; inv[7] = (X[1] * X[6] - X[0] * X[7]) / det; inv[8] = (X[0] * X[4] - X[1] * X[3]) / det;
        double delta = 0, Y[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            Y[r * 3 + c] = 0.5 * (X[r * 3 + c] + inv[c * 3 + r]);
            delta = std::max(delta, std::fabs(Y[r * 3 + c] - X[r * 3 + c]));
        }
        std::memcpy(X, Y, sizeof(X));
        if (delta < 1e-15) break;
    }
    for (int i = 0; i < 9; ++i) Rout[i] = (float)X[i];
}

void sym_eig6(const double* Hin, double* ev) {   // cyclic Jacobi
    double A[36]; std::memcpy(A, Hin, sizeof(A));
    // (the eigenvalues are rounded to f32 by the caller, cvo.cpp:728: off-diagonal mass below 1e-30 of the diagonal's moves them by far less than half an ulp of
    //  that; running on until it underflowed took eight to ten sweeps instead of four or five -- 0.5 ms of host time per 64-pair score block)
    double diag2 = 0; for (int i = 0; i < 6; ++i) diag2 += A[i * 6 + i] * A[i * 6 + i];
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0; for (int i = 0; i < 6; ++i) for (int j = i + 1; j < 6; ++j) off += A[i * 6 + j] * A[i * 6 + j];
        if (off < 1e-300 || off <= 1e-30 * (diag2 + off)) break;
        for (int p = 0; p < 6; ++p) for (int q = p + 1; q < 6; ++q) {
            if (A[p * 6 + q] == 0.0) continue;
            const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.0 * A[p * 6 + q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 6; ++k) { const double akp = A[k * 6 + p], akq = A[k * 6 + q]; A[k * 6 + p] = c * akp - s * akq; A[k * 6 + q] = s * akp + c * akq; }
            for (int k = 0; k < 6; ++k) { const double apk = A[p * 6 + k], aqk = A[q * 6 + k]; A[p * 6 + k] = c * apk - s * aqk; A[q * 6 + k] = s * apk + c * aqk; }
        }
    }
    for (int i = 0; i < 6; ++i) ev[i] = A[i * 6 + i];
}

// 21 upper-triangle terms -> 6x6 [[A, C^T],[C, D]] (cvo.cpp:700-704), then the
// scale / eigen-shift epilogue of se3_Hessian (cvo.cpp:726-758) in f32 as the reference.
void finish_hessian(const double* terms21, int inliers, double Hout[36]) {
    float H[36];
    if (inliers) {
        const double* t = terms21;
        float A3[9], C3[9], D3[9];
        A3[0] = (float)t[0]; A3[1] = A3[3] = (float)t[1]; A3[2] = A3[6] = (float)t[2]; A3[4] = (float)t[3]; A3[5] = A3[7] = (float)t[4]; A3[8] = (float)t[5];
        for (int i = 0; i < 9; ++i) C3[i] = (float)t[6 + i];
        D3[0] = (float)t[15]; D3[1] = D3[3] = (float)t[16]; D3[2] = D3[6] = (float)t[17]; D3[4] = (float)t[18]; D3[5] = D3[7] = (float)t[19]; D3[8] = (float)t[20];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            H[r * 6 + c] = A3[r * 3 + c]; H[r * 6 + 3 + c] = C3[c * 3 + r]; H[(3 + r) * 6 + c] = C3[r * 3 + c]; H[(3 + r) * 6 + 3 + c] = D3[r * 3 + c];
        }
        const float scale = (float)(-1.0 / 100000);                 // cvo.cpp:727
        for (int i = 0; i < 36; ++i) H[i] = H[i] * scale;
        double Hd[36], evd[6]; for (int i = 0; i < 36; ++i) Hd[i] = H[i];
        sym_eig6(Hd, evd);
        float ev[6]; for (int i = 0; i < 6; ++i) ev[i] = (float)evd[i];
        auto min_abs = [&]() { int mi = 0; for (int i = 1; i < 6; ++i) if (std::fabs(ev[i]) < std::fabs(ev[mi])) mi = i; return ev[mi]; };
        float sufficient_scale = 0.0f, min_eigen = min_abs();
        int guard = 0;
        while (std::fabs(min_eigen) < 1.0 && guard++ < 1000) {      // cvo.cpp:740-746
            sufficient_scale += (float)(1.0 - min_eigen);
            for (int i = 0; i < 6; ++i) ev[i] += (float)(1.0 - min_eigen) * 1.0f;
            min_eigen = min_abs();
        }
        for (int i = 0; i < 6; ++i) H[i * 6 + i] += sufficient_scale * 1.0f;   // cvo.cpp:747
    } else {
        for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1.f : 0.f;         // cvo.cpp:755
    }
    for (int i = 0; i < 36; ++i) Hout[i] = (double)H[i];                      // cvo.cpp:758
}

}  // namespace

// =============================================================================
struct cvo_handle_s {
    cvo_params prm;
    Engine eng;
    std::shared_ptr<Cloud> fixed, moving, previous;     // ptr_fixed_pcd / ptr_moving_pcd / ptr_previous_pcd, cvo.hpp:91-94 (shared with last_generated() only: never written again)
    bool pre_pc_init = false, init = false, first_frame = true;
    int num_fixed = 0, num_moving = 0;
    float R[9], T[3], ell;
    Aff transform, prev_transform, accum_transform;
    int iter = 0, A_nonzero = 0;
    int num_want = 3000;                                             // pcd_generator.cpp:22
    Cloud scratch_a, scratch_b;                                      // host clouds handed to function_inner_product / se3_Hessian directly
    // The tracker calls compute_innerproduct(tran = the transform it has just been given) right behind every match_* (local_tracker.cpp:356-375, 415-431): with
    // cvo_set_tail_scores(h, 1) the align launch answers that block in its tail (Engine::tail_scores) and the answers wait here for the call -- valid for
    // exactly these clouds, this ell and this transform; anything else goes to the score kernel as before.  Off by default: one pair alone runs on eight cooperating
    // workgroups, whose tail (a transform, a cull, two list walks, five exchanges) costs the launch 0.12 ms where the score launch behind it costs 0.07-0.10
    // (profiles/r04_tracker_path.txt); batches, one workgroup per pair with every CU busy, are where it pays (cvo_batch_set_tail_scores).
    // What "answers" means for a handle (CVO_HIP_HANDLE_TAIL=queue|kernel, default queue): the score kernel for {mv, fx, the result's transform and ell} is QUEUED right behind the
    // align kernel, before the host starts waiting for the alignment (transform and ell come from the device-resident state, as cvo_batch_enqueue_innerproduct has it);
    // cvo_align returns when the alignment is in, the score kernel runs while the caller looks at the transform, and compute_innerproduct(tran = that transform) only
    // collects.  "kernel" is the batches' way (the align launch's own tail), slower for one pair on eight workgroups (above).
    // cvo_set_tail_scores: 0 never, 1 always, 2 (default) when the handle's previous alignment was followed by exactly that question -- the tracker's two objects are asked
    // after every frame (local_tracker.cpp:375, 431) and queue from their second frame on; a loop-closure object (compute_innerproduct_lc) never does.
    int tail_mode = 2;
    bool asked_after_align = false;                                  // compute_innerproduct(tran = the last alignment's transform) came since that alignment
    bool tail_scores = false;                                        // this alignment starts the block (decided per alignment from the two above)
    bool tail_in_kernel = false;
    bool queued_valid = false;                                       // a score block for (queued_fixed, queued_moving, h->transform, h->ell) is queued or done on eng.score_stream
    const Cloud* queued_fixed = nullptr; const Cloud* queued_moving = nullptr; float queued_ell = 0.f, queued_tran[12];
    int queued_hits = 0;                                             // score blocks answered by what an alignment had queued
    int shared_hits = 0;                                             // clouds this handle took from last_generated() instead of generating them
    int staged_hits = 0;                                             // clouds this handle took from a generation started ahead of time (cvo_stage_next_frame)
    bool tail_valid = false;
    double tail_r[5][24];
    const Cloud* tail_fixed = nullptr; const Cloud* tail_moving = nullptr;
    float tail_ell = 0.f, tail_tran[12];
};

struct cvo_batch_s {
    cvo_params prm;
    Engine eng;
    int max_pairs = 0;
    std::vector<std::unique_ptr<Cloud>> fixed, moving;
    std::vector<PairState> init_states;     // what set_pair / set_state last gave
    bool states_dirty = true;               // device states differ from init_states
    int last_n = 0;
};

namespace {
Cloud* slot_cloud(cvo_handle_s* h, int slot) {
    switch (slot) { case CVO_SLOT_FIXED: return h->fixed.get(); case CVO_SLOT_MOVING: return h->moving.get(); case CVO_SLOT_PREVIOUS: return h->previous.get(); }
    return nullptr;
}
void slots_changed(cvo_handle_s* h) { h->tail_valid = false; h->queued_valid = false; }   // answers held for the clouds that were there are void (a new cloud may live at an old one's address)
std::shared_ptr<Cloud>& slot_to_fill(cvo_handle_s* h) {            // cvo.cpp:352-366: the first cloud is the FIXED one, every later one MOVING
    slots_changed(h);
    if (!h->init) { if (!h->fixed || h->fixed.use_count() > 1 || h->fixed->n > 0) h->fixed = std::make_shared<Cloud>(); return h->fixed; }
    h->moving = std::make_shared<Cloud>();
    return h->moving;
}
void fresh_state(PairState& s, float ell) {
    std::memset(&s, 0, sizeof(s));
    s.R[0] = s.R[4] = s.R[8] = 1.f; s.ell = ell;
    s.transform[0] = s.transform[5] = s.transform[10] = 1.f;
}
int do_align(cvo_handle_s* h, cvo_trace_row* trace, int trace_cap, int* trace_len) {
    if (trace_len) *trace_len = 0;
    if (!h->fixed || !h->moving || h->fixed->n <= 0 || h->moving->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "align: empty fixed or moving cloud");
    PairState st; fresh_state(st, h->ell);
    std::memcpy(st.R, h->R, sizeof(st.R)); std::memcpy(st.T, h->T, sizeof(st.T));
    st.iter = h->iter;
    std::memcpy(st.transform, h->transform.m, sizeof(st.transform));
    std::vector<Engine::PairIn> pairs{{h->fixed.get(), h->moving.get()}};
    const bool want_trace = trace && trace_cap > 0;
    h->tail_valid = false; h->queued_valid = false;
    h->tail_scores = h->tail_mode == 1 || (h->tail_mode == 2 && h->asked_after_align);
    h->asked_after_align = false;
    h->eng.tail_scores = h->tail_scores && h->tail_in_kernel;
    int rc = h->eng.launch(pairs, &st, true, nullptr, want_trace, trace_cap); if (rc) return rc;
    if (want_trace) {
        // results copy is already queued; queue the trace copies behind it on the same stream
        HIP_TRY(hipMemcpyAsync(trace, h->eng.d_trace.p, sizeof(TraceRow) * trace_cap, hipMemcpyDeviceToHost, h->eng.stream));
        HIP_TRY(hipMemcpyAsync(trace_len, h->eng.d_tracelen.p, sizeof(int), hipMemcpyDeviceToHost, h->eng.stream));
    }
    bool queued = false;
    if (h->tail_scores && !h->tail_in_kernel && !want_trace) {       // the tracker's score block (cvo.cpp:489-500) behind the align kernel, transform and ell from the pair's state
        const Cloud* fx = h->fixed.get(); const Cloud* mv = h->moving.get();
        const Engine::ScoreReq rq[5] = {{mv, nullptr, fx, false, 0.f, 0, false}, {mv, nullptr, fx, false, 0.f, 0, true}, {fx, nullptr, fx, false, 0.f, 0, false},
                                        {mv, nullptr, mv, false, 0.f, 0, false}, {mv, nullptr, fx, true, 0.f, 0, true}};
        queued = h->eng.score_enqueue(rq, 5, h->eng.last_stream) == CVO_OK;
    }
    rc = queued ? h->eng.wait_align_only() : h->eng.wait(); if (rc) { h->eng.score_pending = 0; return rc; }
    const PairState& r = h->eng.results()[0];
    if (r.status != CVO_OK) {
        // the block queued behind a FAILED alignment was computed from that alignment's device state: it answers nobody's question.  Let it leave the
        // device (the next score launch reuses its buffers) and drop it; the handle keeps the transform and ell of the alignment before.
        if (queued) { (void)hipStreamSynchronize(h->eng.last_stream); h->eng.score_pending = 0; }
        return fail(r.status, "align kernel reported an error (6 = inter-workgroup wait timed out)");
    }
    if (queued) { h->queued_valid = true; h->queued_fixed = h->fixed.get(); h->queued_moving = h->moving.get(); }   // (ell and transform: below, from the result)
    std::memcpy(h->R, r.R, sizeof(h->R)); std::memcpy(h->T, r.T, sizeof(h->T));
    h->ell = r.ell; h->iter = r.iter; h->A_nonzero = r.A_nonzero;
    Aff prev; std::memcpy(prev.m, r.prev_transform, sizeof(prev.m));
    h->prev_transform = prev;                                        // cvo.cpp:815
    h->accum_transform = aff_mul(h->accum_transform, prev);          // cvo.cpp:816
    std::memcpy(h->transform.m, r.transform, sizeof(float) * 12);    // update_tf, cvo.cpp:817
    if (h->queued_valid) { h->queued_ell = h->ell; std::memcpy(h->queued_tran, h->transform.m, sizeof(h->queued_tran)); }
    if (h->eng.last_tail) {                                           // the score block this launch answered in its tail, for the compute_innerproduct that follows
        std::memcpy(h->tail_r, h->eng.h_tail.p, sizeof(h->tail_r));
        h->tail_fixed = h->fixed.get(); h->tail_moving = h->moving.get(); h->tail_ell = h->ell;
        std::memcpy(h->tail_tran, h->transform.m, sizeof(h->tail_tran));
        h->tail_valid = true;
    }
    return CVO_OK;
}
}  // namespace

extern "C" {

const char* cvo_last_error(void) { return g_err.c_str(); }

int cvo_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < count; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int cvo_default_params(cvo_params* p) {
    if (!p) return fail(CVO_ERR_INVALID, "null params");
    p->ell = 0.15; p->sigma = 0.1; p->sp_thres = 8e-3; p->c = 7.0; p->d = 7.0; p->c_ell = 200; p->c_sigma = 1;
    p->max_iter = 2000; p->min_step = 2 * 1.0e-1; p->eps = 5 * 1.0e-5; p->eps_2 = 1.0e-5;
    return CVO_OK;
}

int cvo_create(const cvo_params* p, int device, cvo_handle* out) {
    if (!out) return fail(CVO_ERR_INVALID, "null out");
    *out = nullptr;
    std::unique_ptr<cvo_handle_s> h(new cvo_handle_s());
    if (p) h->prm = *p; else cvo_default_params(&h->prm);
    int rc = h->eng.init(device, h->prm); if (rc) { h->eng.destroy(); return rc; }   // whatever init had created before it failed
    h->fixed.reset(new Cloud());                                     // ptr_fixed_pcd(new point_cloud), cvo.cpp:30
    h->ell = h->prm.ell;
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(h->R, I, sizeof(I)); h->T[0] = h->T[1] = h->T[2] = 0;  // cvo.cpp:66-67
    h->transform = h->prev_transform = h->accum_transform = aff_identity();   // cvo.cpp:68-70
    *out = h.release();
    return CVO_OK;
}

int cvo_destroy(cvo_handle h) {
    if (!h) return CVO_OK;
    (void)hipSetDevice(h->eng.device);
    if (h->eng.stream) (void)hipStreamSynchronize(h->eng.stream);
    h->fixed.reset(); h->moving.reset(); h->previous.reset();
    h->eng.destroy();
    delete h;
    return CVO_OK;
}

int cvo_set_pcd(cvo_handle h, const float* xyz, const float* feat, int n) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    const bool first = !h->init;
    std::shared_ptr<Cloud>& c = slot_to_fill(h);
    int rc = h->eng.upload(*c, xyz, feat, n); if (rc) return rc;
    if (first) { h->init = true; return CVO_OK; }                    // cvo.cpp:352-360
    h->num_fixed = h->fixed ? h->fixed->n : 0; h->num_moving = h->moving->n;   // cvo.cpp:370-371
    h->A_nonzero = 0;                                                // cvo.cpp:385
    return CVO_OK;
}

int cvo_set_tail_scores(cvo_handle h, int on) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (on < 0 || on > 2) return fail(CVO_ERR_INVALID, "cvo_set_tail_scores: 0 (never), 1 (always) or 2 (when the previous alignment was asked)");
    h->tail_mode = on; h->asked_after_align = false;
    if (const char* e = std::getenv("CVO_HIP_HANDLE_TAIL")) h->tail_in_kernel = std::strcmp(e, "kernel") == 0;
    if (!on) slots_changed(h);
    return CVO_OK;
}
int cvo_set_num_want(cvo_handle h, int num_want) {
    if (!h || num_want <= 0) return fail(CVO_ERR_INVALID, "bad argument");
    h->num_want = num_want; return CVO_OK;
}

namespace {
// The frame this thread's last cloud was generated from, byte for byte?  Then `c` becomes a device copy of that cloud (positions + features, selected pixels:
// 100 KB at 3 k points, queued on the handle's stream) and nothing is generated: 0.22 ms -> the compare of 1.5 MB the staging copy would have read anyway.
// A different frame differs within the first bytes and costs nothing.
int take_from(const LastGenerated& g, cvo_handle_s* h, Cloud& c, const unsigned char* bgr8, const unsigned short* depth16, int w, int hh, const cvo_camera& cam, bool* taken) {
    *taken = false;
    if (!g.cloud || !g.stage || g.device != h->eng.device || g.w != w || g.h != hh || g.num_want != h->num_want ||
        std::memcmp(&g.cam, &cam, sizeof(cam)) != 0 || !bgr8 || !depth16) return CVO_OK;
    const size_t n = (size_t)w * hh;
    {
        std::lock_guard<std::mutex> lk(g.stage->mu);
        if (g.stage->stamp.load(std::memory_order_acquire) != g.stamp || g.stage->buf.bytes < 5 * n) return CVO_OK;
        const unsigned char* st = static_cast<const unsigned char*>(g.stage->buf.p);
        if (std::memcmp(st, bgr8, 3 * n) != 0 || std::memcmp(st + 3 * n, depth16, 2 * n) != 0) return CVO_OK;
    }
    const Cloud& src = *g.cloud;
    HIP_TRY(hipSetDevice(h->eng.device));
    c.n = src.n; c.n_px = src.n_px; c.cost_hint = src.cost_hint; c.boxes_valid = false; c.raw = nullptr;
    if (src.n > 0) {
        int rc = c.buf.ensure((size_t)src.n * REC * sizeof(float)); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c.buf.p, src.buf.p, (size_t)src.n * REC * sizeof(float), hipMemcpyDeviceToDevice, h->eng.stream));
    }
    if (src.n_px > 0) {
        int rc = c.px.ensure((size_t)src.n_px * 2 * sizeof(uint16_t)); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c.px.p, src.px.p, (size_t)src.n_px * 2 * sizeof(uint16_t), hipMemcpyDeviceToDevice, h->eng.stream));
    }
    HIP_TRY(hipStreamSynchronize(h->eng.stream));                    // complete when the call returns, as a generated cloud is (any stream may read it next)
    *taken = true;
    return CVO_OK;
}
int take_generated(cvo_handle_s* h, Cloud& c, const unsigned char* bgr8, const unsigned short* depth16, int w, int hh, const cvo_camera& cam, bool* taken) {
    *taken = false;
    if (!share_generated_clouds()) return CVO_OK;
    const int rc = take_from(last_generated(), h, c, bgr8, depth16, w, hh, cam, taken);
    if (rc == CVO_OK && *taken) ++h->shared_hits;
    return rc;
}

// ---- the NEXT frame's cloud, generated ahead of its set_pcd (cvo_stage_next_frame).  The reference's tracker handles a frame strictly in sequence
// (local_tracker.cpp:356 set_pcd + align, :375 scores, :415 set_pcd + align, :431 scores) and its generator is the first thing each frame waits for; frame t + 1's
// images do not depend on frame t's alignment, so a caller that has them early hands them over here and goes on with frame t: a worker thread with an engine of
// its own (own stream, own scratch) runs the generator beside the alignment -- eight workgroups of 256 CUs --, and cvo_set_pcd_images finds the finished cloud by
// comparing the images byte for byte, as it does for the second object of a frame.  One staged frame per host thread; the images must stay as they are until the
// cvo_set_pcd_images that takes them (or the next cvo_stage_next_frame) has returned.
struct FrameStager {
    Engine eng; bool ready = false;
    std::thread th; std::mutex mu; std::condition_variable cv;
    bool quit = false, busy = false, have = false;
    const unsigned char* bgr = nullptr; const unsigned short* depth = nullptr; int w = 0, h = 0, num_want = 0; cvo_camera cam{};
    LastGenerated out;              // valid when `have`: the staged frame and its cloud
    std::shared_ptr<ImageStage> stages[2]; int flip = 0;
    int rc = CVO_OK; std::string err;
    int start(int device, const cvo_params& prm) {
        const int r = eng.init(device, prm); if (r) return r;
        ready = true;
        th = std::thread([this] { run(); });
        return CVO_OK;
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return quit || busy; });
            if (quit) return;
            lk.unlock();
            auto cloud = std::make_shared<Cloud>();
            // two stages in turn: the frame staged before this one has just been taken (or is about to be), and its second object still compares against ITS images
            if (!stages[flip]) stages[flip] = std::make_shared<ImageStage>();
            eng.img_stage = stages[flip]; flip ^= 1;
            const int r = eng.generate_pcd(*cloud, bgr, depth, w, h, cam, num_want);
            std::string e = r ? g_err : std::string();
            lk.lock();
            rc = r; err = e; have = (r == CVO_OK);
            if (have) {
                out.stage = eng.img_stage; out.stamp = out.stage->stamp.load(std::memory_order_acquire);
                out.device = eng.device; out.w = w; out.h = h; out.num_want = num_want; out.cam = cam; out.cloud = cloud;
            } else out.cloud.reset();
            busy = false;
            cv.notify_all();
        }
    }
    void wait_idle() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return !busy; }); }
    ~FrameStager() {
        if (ready) {
            { std::lock_guard<std::mutex> lk(mu); quit = true; }
            cv.notify_all();
            if (th.joinable()) th.join();
            out.cloud.reset();
            eng.destroy();
        }
    }
};
struct StagerOwner { FrameStager* p = nullptr; ~StagerOwner() { delete p; } };
FrameStager* frame_stager(int device, const cvo_params& prm, int* rc_out) {
    static thread_local StagerOwner own;
    *rc_out = CVO_OK;
    if (own.p && own.p->eng.device != device) { delete own.p; own.p = nullptr; }
    if (!own.p) {
        own.p = new FrameStager();
        const int rc = own.p->start(device, prm);
        if (rc) { delete own.p; own.p = nullptr; *rc_out = rc; }
    }
    frame_stager_slot() = own.p;
    return own.p;
}
}  // namespace

int cvo_set_pcd_images(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera* cam) {
    if (!h || !cam) return fail(CVO_ERR_INVALID, "null argument");
    const bool first = !h->init;
    std::shared_ptr<Cloud>& c = slot_to_fill(h);
    bool taken = false;
    int rc = take_generated(h, *c, bgr8, depth16, width, height, *cam, &taken); if (rc) return rc;
    if (!taken) {
        if (FrameStager* fs = frame_stager_slot()) {                  // a frame staged ahead (cvo_stage_next_frame)?  wait for its generation, then compare
            fs->wait_idle();
            if (fs->have) {
                rc = take_from(fs->out, h, *c, bgr8, depth16, width, height, *cam, &taken); if (rc) return rc;
                if (taken) {
                    ++h->staged_hits;
                    if (share_generated_clouds()) last_generated() = fs->out;   // the frame's second object finds it where it looks (take_generated)
                    fs->have = false; fs->out.cloud.reset();
                }
            }
        }
    }
    if (!taken) {
        LastGenerated& g = last_generated();
        g.cloud.reset();                                             // (frees the previous frame's cloud unless a handle still holds it)
        rc = h->eng.generate_pcd(*c, bgr8, depth16, width, height, *cam, h->num_want); if (rc) return rc;
        if (share_generated_clouds()) {
            g.stage = h->eng.img_stage; g.stamp = g.stage->stamp.load(std::memory_order_acquire);
            g.device = h->eng.device; g.w = width; g.h = height; g.num_want = h->num_want; g.cam = *cam; g.cloud = c;
        }
    }
    if (first) { h->init = true; return CVO_OK; }                    // cvo.cpp:352-360
    h->num_fixed = h->fixed ? h->fixed->n : 0; h->num_moving = h->moving->n;   // cvo.cpp:370-371
    h->A_nonzero = 0;                                                // cvo.cpp:385
    return CVO_OK;
}
int cvo_stage_next_frame(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera* cam) {
    if (!h || !cam || !bgr8 || !depth16) return fail(CVO_ERR_INVALID, "null argument");
    if (width < 64 || height < 64 || (size_t)width * height > (size_t)1 << 26) return fail(CVO_ERR_INVALID, "image size out of range");
    int rc = CVO_OK;
    FrameStager* fs = frame_stager(h->eng.device, h->prm, &rc);
    if (!fs) return rc;
    fs->wait_idle();                                                  // (a frame staged before and never asked for: dropped)
    {
        std::lock_guard<std::mutex> lk(fs->mu);
        fs->have = false; fs->out.cloud.reset();
        fs->bgr = bgr8; fs->depth = depth16; fs->w = width; fs->h = height; fs->num_want = h->num_want; fs->cam = *cam;
        fs->busy = true;
    }
    fs->cv.notify_all();
    return CVO_OK;
}
int cvo_staged_frame_count(cvo_handle h, int* count) { if (!h || !count) return fail(CVO_ERR_INVALID, "null argument"); *count = h->staged_hits; return CVO_OK; }
int cvo_queued_score_count(cvo_handle h, int* count) { if (!h || !count) return fail(CVO_ERR_INVALID, "null argument"); *count = h->queued_hits; return CVO_OK; }
int cvo_shared_cloud_count(cvo_handle h, int* count) { if (!h || !count) return fail(CVO_ERR_INVALID, "null argument"); *count = h->shared_hits; return CVO_OK; }

int cvo_get_cloud(cvo_handle h, int slot, float* xyz, float* feat, int cap, int* n) {
    if (!h || !n) return fail(CVO_ERR_INVALID, "null argument");
    Cloud* c = slot_cloud(h, slot);
    *n = c ? c->n : 0;
    if (!c || c->n == 0 || cap < c->n) return CVO_OK;
    if (!xyz || !feat) return fail(CVO_ERR_INVALID, "null output array");
    HIP_TRY(hipSetDevice(h->eng.device));
    DevBuf tmp; int rc = tmp.ensure(sizeof(float) * 8 * (size_t)c->n); if (rc) return rc;
    float* dx = static_cast<float*>(tmp.p); float* df = dx + 3 * (size_t)c->n;
    hipError_t e = pcd_launch_unpack(c->rec(), c->n, dx, df, h->eng.stream);
    if (e != hipSuccess) { tmp.release(); return fail(CVO_ERR_HIP, std::string("unpack kernel: ") + hipGetErrorString(e)); }
    hipError_t e1 = hipMemcpyAsync(xyz, dx, sizeof(float) * 3 * (size_t)c->n, hipMemcpyDeviceToHost, h->eng.stream);
    hipError_t e2 = hipMemcpyAsync(feat, df, sizeof(float) * 5 * (size_t)c->n, hipMemcpyDeviceToHost, h->eng.stream);
    hipError_t e3 = hipStreamSynchronize(h->eng.stream);
    tmp.release();
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(CVO_ERR_HIP, "cloud download failed");
    return CVO_OK;
}

int cvo_get_selected_points(cvo_handle h, int slot, unsigned short* px, int cap, int* n) {
    if (!h || !n) return fail(CVO_ERR_INVALID, "null argument");
    Cloud* c = slot_cloud(h, slot);
    *n = c ? c->n_px : 0;
    if (!c || c->n_px == 0 || cap < c->n_px) return CVO_OK;
    if (!px) return fail(CVO_ERR_INVALID, "null output array");
    HIP_TRY(hipSetDevice(h->eng.device));
    HIP_TRY(hipMemcpy(px, c->px.p, sizeof(unsigned short) * 2 * (size_t)c->n_px, hipMemcpyDeviceToHost));
    return CVO_OK;
}

int cvo_align_traced(cvo_handle h, cvo_trace_row* trace, int trace_cap, int* trace_len) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (trace && !trace_len) return fail(CVO_ERR_INVALID, "trace_len required with trace");
    return do_align(h, trace, trace_cap, trace_len);
}
int cvo_align(cvo_handle h) { return cvo_align_traced(h, nullptr, 0, nullptr); }

int cvo_match_odometry(cvo_handle h, const float* xyz, const float* feat, int n, double transform_out[12]) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (!h->init) return fail(CVO_ERR_NOT_INITIALIZED, "cvo not initialized !");    // cvo.cpp:463-466
    int rc = cvo_set_pcd(h, xyz, feat, n); if (rc) return rc;
    rc = cvo_align(h); if (rc) return rc;
    if (transform_out) for (int i = 0; i < 12; ++i) transform_out[i] = (double)h->transform.m[i];   // cvo.cpp:472
    return CVO_OK;
}
int cvo_match_odometry_images(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera* cam,
                              double transform_out[12]) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (!h->init) return fail(CVO_ERR_NOT_INITIALIZED, "cvo not initialized !");    // cvo.cpp:463-466
    int rc = cvo_set_pcd_images(h, bgr8, depth16, width, height, cam); if (rc) return rc;
    rc = cvo_align(h); if (rc) return rc;
    if (transform_out) for (int i = 0; i < 12; ++i) transform_out[i] = (double)h->transform.m[i];   // cvo.cpp:472
    return CVO_OK;
}
int cvo_match_keyframe_images(cvo_handle h, const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera* cam,
                              double transform_out[12]) {
    return cvo_match_odometry_images(h, bgr8, depth16, width, height, cam, transform_out);   // cvo.cpp:563-576 is the same body
}
int cvo_match_keyframe(cvo_handle h, const float* xyz, const float* feat, int n, double transform_out[12]) {
    return cvo_match_odometry(h, xyz, feat, n, transform_out);       // cvo.cpp:563-576 is the same body
}

namespace {
void finish_inn_p(const double r[24], cvo_inn_p* out) {
    double sum = r[1]; if (sum == 0) sum = 1;                        // cvo.cpp:455-456
    out->value = (float)r[0]; out->num = (int)sum; out->num_e = 0;   // cvo.cpp:457
}
}  // namespace

int cvo_function_inner_product(cvo_handle h, int slot_a, const float* tran_a, int slot_b, cvo_inn_p* out) {
    if (!h || !out) return fail(CVO_ERR_INVALID, "null argument");
    Cloud* a = slot_cloud(h, slot_a); Cloud* b = slot_cloud(h, slot_b);
    if (!a || !b || a->n <= 0 || b->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "function_inner_product: empty cloud slot");
    double r[1][24];
    const Engine::ScoreReq rq[1] = {{a, tran_a, b, false, h->ell}};
    int rc = h->eng.score_many(rq, 1, r); if (rc) return rc;
    finish_inn_p(r[0], out);
    return CVO_OK;
}

int cvo_se3_hessian(cvo_handle h, int slot_a, const float* tran_a, int slot_b, double H[36], int* inliers) {
    if (!h || !H || !inliers) return fail(CVO_ERR_INVALID, "null argument");
    Cloud* a = slot_cloud(h, slot_a); Cloud* b = slot_cloud(h, slot_b);
    if (!a || !b || a->n <= 0 || b->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "se3_Hessian: empty cloud slot");
    double r[1][24];
    const Engine::ScoreReq rq[1] = {{a, tran_a, b, true, h->ell}};
    int rc = h->eng.score_many(rq, 1, r); if (rc) return rc;
    *inliers += (int)r[0][1];                                        // cvo.cpp:708 increments the caller's variable
    finish_hessian(r[0] + 2, *inliers, H);
    return CVO_OK;
}

namespace {
int stage_clouds(cvo_handle h, const float* xyz_a, const float* feat_a, int n_a, const float* xyz_b, const float* feat_b, int n_b) {
    if (n_a <= 0 || n_b <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "empty cloud");
    int rc = h->eng.upload(h->scratch_a, xyz_a, feat_a, n_a); if (rc) return rc;
    return h->eng.upload(h->scratch_b, xyz_b, feat_b, n_b);
}
}  // namespace
int cvo_function_inner_product_clouds(cvo_handle h, const float* xyz_a, const float* feat_a, int n_a, const float* xyz_b, const float* feat_b, int n_b,
                                      cvo_inn_p* out) {
    if (!h || !out) return fail(CVO_ERR_INVALID, "null argument");
    int rc = stage_clouds(h, xyz_a, feat_a, n_a, xyz_b, feat_b, n_b); if (rc) return rc;
    double r[1][24];
    const Engine::ScoreReq rq[1] = {{&h->scratch_a, nullptr, &h->scratch_b, false, h->ell}};
    rc = h->eng.score_many(rq, 1, r); if (rc) return rc;
    finish_inn_p(r[0], out);
    return CVO_OK;
}
int cvo_se3_hessian_clouds(cvo_handle h, const float* xyz_a, const float* feat_a, int n_a, const float* xyz_b, const float* feat_b, int n_b,
                           double H[36], int* inliers) {
    if (!h || !H || !inliers) return fail(CVO_ERR_INVALID, "null argument");
    int rc = stage_clouds(h, xyz_a, feat_a, n_a, xyz_b, feat_b, n_b); if (rc) return rc;
    double r[1][24];
    const Engine::ScoreReq rq[1] = {{&h->scratch_a, nullptr, &h->scratch_b, true, h->ell}};
    rc = h->eng.score_many(rq, 1, r); if (rc) return rc;
    *inliers += (int)r[0][1];                                        // cvo.cpp:708
    finish_hessian(r[0] + 2, *inliers, H);
    return CVO_OK;
}

// The whole score block of the tracker is one launch (the reference runs 4 KD-tree builds + 5 sweeps, cvo.cpp:489-500).
int cvo_compute_innerproduct(cvo_handle h, cvo_inn_p* inn_pre, cvo_inn_p* inn_post, double post_hessian[36], const float tran[12],
                             int* inliers, cvo_inn_p* inn_fixed_pcd, cvo_inn_p* inn_moving_pcd, float* cos_angle) {
    if (!h || !inn_pre || !inn_post || !post_hessian || !tran || !inliers || !inn_fixed_pcd || !inn_moving_pcd || !cos_angle)
        return fail(CVO_ERR_INVALID, "null argument");
    Cloud* fx = slot_cloud(h, CVO_SLOT_FIXED); Cloud* mv = slot_cloud(h, CVO_SLOT_MOVING);
    if (!fx || !mv || fx->n <= 0 || mv->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "function_inner_product: empty cloud slot");
    const Engine::ScoreReq rq[5] = {{mv, nullptr, fx, false, h->ell},        // cvo.cpp:489
                                    {mv, tran, fx, false, h->ell},           // cvo.cpp:491
                                    {fx, nullptr, fx, false, h->ell},        // cvo.cpp:496
                                    {mv, nullptr, mv, false, h->ell},        // cvo.cpp:497
                                    {mv, tran, fx, true, h->ell}};           // cvo.cpp:500
    double r[5][24];
    int rc;
    if (h->eng.launched && std::memcmp(h->transform.m, tran, sizeof(float) * 12) == 0) h->asked_after_align = true;   // the tracker's pattern: the next alignment starts this block itself (mode 2)
    if (h->queued_valid && h->eng.score_pending == 5) {
        // queued behind the align kernel for the alignment's own transform and ell (do_align): for exactly that question the answers are on their way or there
        const bool same = h->queued_fixed == fx && h->queued_moving == mv && h->queued_ell == h->ell && std::memcmp(h->queued_tran, tran, sizeof(h->queued_tran)) == 0;
        h->queued_valid = false;
        rc = h->eng.score_collect(5, r); if (rc) return rc;          // (collected either way: the pinned block is free again)
        if (!same) { rc = h->eng.score_many(rq, 5, r); if (rc) return rc; } else ++h->queued_hits;
    } else if (h->tail_valid && h->tail_fixed == fx && h->tail_moving == mv && h->tail_ell == h->ell && std::memcmp(h->tail_tran, tran, sizeof(h->tail_tran)) == 0) {
        // answered by the align launch itself; what its workgroups could not answer (PairDesc::score_out[23]) goes to the score kernel now
        std::memcpy(r, h->tail_r, sizeof(r));
        static const int bit_of[5] = {TAIL_PRE, TAIL_POST, TAIL_FIXED, TAIL_MOVING, TAIL_HESSIAN};
        const int mask = (int)r[0][23];
        Engine::ScoreReq miss[5]; int where[5], nmiss = 0;
        for (int q = 0; q < 5; ++q) if (!(mask & bit_of[q])) { miss[nmiss] = rq[q]; where[nmiss++] = q; }
        if (nmiss) {
            double extra[5][24];
            rc = h->eng.score_many(miss, nmiss, extra); if (rc) return rc;
            for (int k = 0; k < nmiss; ++k) std::memcpy(r[where[k]], extra[k], sizeof(double) * 24);
        }
    } else {
        rc = h->eng.score_many(rq, 5, r); if (rc) return rc;
    }
    finish_inn_p(r[0], inn_pre); finish_inn_p(r[1], inn_post); finish_inn_p(r[2], inn_fixed_pcd); finish_inn_p(r[3], inn_moving_pcd);
    *cos_angle = inn_post->value / (sqrtf(inn_fixed_pcd->value) * sqrtf(inn_moving_pcd->value));                  // cvo.cpp:498
    *inliers += (int)r[4][1];                                        // cvo.cpp:708
    finish_hessian(r[4] + 2, *inliers, post_hessian);
    return CVO_OK;
}

int cvo_compute_innerproduct_lc(cvo_handle h, cvo_inn_p* inn_prior, cvo_inn_p* inn_lc_prior, cvo_inn_p* inn_lc_pre, cvo_inn_p* inn_lc_post,
                                double post_hessian[36], const float prior_tran[12], const float lc_prior_tran[12],
                                const float lc_prior_tran_2[12], const float lc_tran[12], int* inliers_svd, int* inliers_pnpransac,
                                cvo_inn_p* inn_fixed_pcd, cvo_inn_p* inn_moving_pcd, float* cos_angle) {
    if (!h || !inn_prior || !inn_lc_prior || !inn_lc_pre || !inn_lc_post || !post_hessian || !prior_tran || !lc_prior_tran ||
        !lc_prior_tran_2 || !lc_tran || !inliers_svd || !inliers_pnpransac || !inn_fixed_pcd || !inn_moving_pcd || !cos_angle)
        return fail(CVO_ERR_INVALID, "null argument");
    Cloud* fx = slot_cloud(h, CVO_SLOT_FIXED); Cloud* mv = slot_cloud(h, CVO_SLOT_MOVING);
    if (!fx || !mv || fx->n <= 0 || mv->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "function_inner_product: empty cloud slot");
    const Engine::ScoreReq rq[8] = {{mv, prior_tran, fx, false, h->ell},     // cvo.cpp:539
                                    {mv, lc_prior_tran, fx, false, h->ell},  // cvo.cpp:541
                                    {mv, nullptr, fx, false, h->ell},        // cvo.cpp:543
                                    {mv, lc_tran, fx, false, h->ell},        // cvo.cpp:545
                                    {fx, nullptr, fx, false, h->ell},        // cvo.cpp:550
                                    {mv, nullptr, mv, false, h->ell},        // cvo.cpp:551
                                    {mv, lc_tran, fx, true, h->ell},         // cvo.cpp:555
                                    {mv, lc_prior_tran_2, fx, true, h->ell}};   // cvo.cpp:558
    double r[8][24];
    int rc = h->eng.score_many(rq, 8, r); if (rc) return rc;
    finish_inn_p(r[0], inn_prior); finish_inn_p(r[1], inn_lc_prior); finish_inn_p(r[2], inn_lc_pre); finish_inn_p(r[3], inn_lc_post);
    finish_inn_p(r[4], inn_fixed_pcd); finish_inn_p(r[5], inn_moving_pcd);
    *cos_angle = inn_lc_post->value / (sqrtf(inn_fixed_pcd->value) * sqrtf(inn_moving_pcd->value));                       // cvo.cpp:552
    *inliers_svd = (int)r[6][1];                                                                                          // cvo.cpp:554-555
    finish_hessian(r[6] + 2, *inliers_svd, post_hessian);
    *inliers_pnpransac = (int)r[7][1];                                                                                    // cvo.cpp:557-558 (only the inlier count is used)
    return CVO_OK;
}

int cvo_update_fixed_pcd(cvo_handle h) { if (!h) return fail(CVO_ERR_INVALID, "null handle"); slots_changed(h); h->fixed = std::move(h->moving); return CVO_OK; }
int cvo_update_previous_pcd(cvo_handle h) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    slots_changed(h); h->previous = std::move(h->moving); h->pre_pc_init = true; return CVO_OK;
}
int cvo_reset_transform(cvo_handle h, const float odometry[12]) {
    if (!h || !odometry) return fail(CVO_ERR_INVALID, "null argument");
    std::memcpy(h->transform.m, odometry, sizeof(float) * 12); return CVO_OK;
}
int cvo_reset_keyframe(cvo_handle h, const float odometry[12]) {
    if (!h || !odometry) return fail(CVO_ERR_INVALID, "null argument");
    slots_changed(h);
    if (!h->pre_pc_init) { h->fixed = std::move(h->moving); }
    else { h->fixed = std::move(h->previous); cvo_update_previous_pcd(h); }
    return cvo_reset_transform(h, odometry);
}
int cvo_reset_initial(cvo_handle h, const float odometry[12], float init_inverse_out[12]) {
    if (!h || !odometry) return fail(CVO_ERR_INVALID, "null argument");
    Aff od; std::memcpy(od.m, odometry, sizeof(od.m));
    const Aff init = aff_inverse(aff_mul(h->transform, od));          // cvo.cpp:613
    float L[9]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L[r * 3 + c] = init.m[r * 4 + c];
    polar_rotation(L, h->R);                                         // cvo.cpp:614
    h->T[0] = init.m[3]; h->T[1] = init.m[7]; h->T[2] = init.m[11];   // cvo.cpp:615
    if (init_inverse_out) { const Aff back = aff_inverse(init); std::memcpy(init_inverse_out, back.m, sizeof(back.m)); }   // cvo.cpp:617
    return CVO_OK;
}

int cvo_get_fixed_and_moving_number(cvo_handle h, int* fixed_num, int* moving_num) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (fixed_num) *fixed_num = h->num_fixed;
    if (moving_num) *moving_num = h->num_moving;
    return CVO_OK;
}
int cvo_get_iteration_number(cvo_handle h, int* iteration) { if (!h || !iteration) return fail(CVO_ERR_INVALID, "null argument"); *iteration = h->iter; return CVO_OK; }
int cvo_get_A_nonzero(cvo_handle h, int* nonzero) { if (!h || !nonzero) return fail(CVO_ERR_INVALID, "null argument"); *nonzero = h->A_nonzero; return CVO_OK; }
int cvo_get_transform(cvo_handle h, float transform[12]) { if (!h || !transform) return fail(CVO_ERR_INVALID, "null argument"); std::memcpy(transform, h->transform.m, sizeof(float) * 12); return CVO_OK; }
int cvo_get_prev_accum_transform(cvo_handle h, float prev_transform[12], float accum_transform[12]) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (prev_transform) std::memcpy(prev_transform, h->prev_transform.m, sizeof(float) * 12);
    if (accum_transform) std::memcpy(accum_transform, h->accum_transform.m, sizeof(float) * 12);
    return CVO_OK;
}
int cvo_get_init(cvo_handle h, int* init) { if (!h || !init) return fail(CVO_ERR_INVALID, "null argument"); *init = h->init ? 1 : 0; return CVO_OK; }
int cvo_get_first_frame(cvo_handle h, int* first_frame) { if (!h || !first_frame) return fail(CVO_ERR_INVALID, "null argument"); *first_frame = h->first_frame ? 1 : 0; return CVO_OK; }
int cvo_set_first_frame(cvo_handle h, int first_frame) { if (!h) return fail(CVO_ERR_INVALID, "null handle"); h->first_frame = first_frame != 0; return CVO_OK; }
int cvo_get_state(cvo_handle h, float R[9], float T[3], float* ell) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (R) std::memcpy(R, h->R, sizeof(h->R));
    if (T) std::memcpy(T, h->T, sizeof(h->T));
    if (ell) *ell = h->ell;
    return CVO_OK;
}
int cvo_set_state(cvo_handle h, const float R[9], const float T[3], float ell) {
    if (!h) return fail(CVO_ERR_INVALID, "null handle");
    if (R) std::memcpy(h->R, R, sizeof(h->R));
    if (T) std::memcpy(h->T, T, sizeof(h->T));
    h->ell = ell;
    return CVO_OK;
}
int cvo_set_workgroups(cvo_handle h, int workgroups_per_pair) {
    if (!h || workgroups_per_pair < 0) return fail(CVO_ERR_INVALID, "bad argument");
    h->eng.wg_request = workgroups_per_pair; return CVO_OK;
}

// ------------------------------------------------------------------ batches
namespace {
int selftest(int device, int kind, int n, const float* in, int in_w, float* out, int out_w) {
    int rc = check_device(device, nullptr); if (rc) return rc;
    if (n <= 0 || !in || !out) return fail(CVO_ERR_INVALID, "bad self-test arguments");
    HIP_TRY(hipSetDevice(device));
    DevBuf din, dout;
    if ((rc = din.ensure(sizeof(float) * (size_t)n * in_w)) || (rc = dout.ensure(sizeof(float) * (size_t)n * out_w))) { din.release(); dout.release(); return rc; }
    hipError_t e = hipMemcpy(din.p, in, sizeof(float) * (size_t)n * in_w, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_selftest(kind, static_cast<const float*>(din.p), static_cast<float*>(dout.p), n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, sizeof(float) * (size_t)n * out_w, hipMemcpyDeviceToHost);
    din.release(); dout.release();
    if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("self-test: ") + hipGetErrorString(e));
    return CVO_OK;
}
}  // namespace
int cvo_adaptive_default_params(cvo_adaptive_params* p) {
    if (!p) return fail(CVO_ERR_INVALID, "null params");
    p->ell_init = 0.1; p->ell_min = 0.0391; p->ell_max = 0.15; p->dl_step = 0.3;
    p->sigma = 0.1; p->sp_thres = 8.315e-3; p->c = 7.0; p->d = 7.0; p->c_ell = 0.5; p->c_sigma = 1;
    p->max_iter = 2000; p->min_step = 2 * 1.0e-1; p->eps = 5 * 1.0e-5; p->eps_2 = 1.0e-5;
    return CVO_OK;
}
int cvo_adaptive_align(int device, const cvo_adaptive_params* p_in, const float* fixed_xyz, const float* fixed_feat, int n_fixed,
                       const float* moving_xyz, const float* moving_feat, int n_moving, float R_inout[9], float T_inout[3], float* ell_out,
                       float transform_out[12], int* iter, cvo_adaptive_row* trace, int trace_cap, int* trace_len) {
    if (trace_len) *trace_len = 0;
    if (!R_inout || !T_inout) return fail(CVO_ERR_INVALID, "null pose");
    if (n_fixed <= 0 || n_moving <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "adaptive align: empty cloud");
    cvo_adaptive_params ap; if (p_in) ap = *p_in; else cvo_adaptive_default_params(&ap);
    cvo_params bp; cvo_default_params(&bp);
    Engine eng;
    int rc = eng.init(device, bp); if (rc) return rc;
    struct Guard { Engine& e; ~Guard() { e.destroy(); } } guard{eng};
    Cloud fx, mv;
    if ((rc = eng.upload(fx, fixed_xyz, fixed_feat, n_fixed))) return rc;
    if ((rc = eng.upload(mv, moving_xyz, moving_feat, n_moving))) return rc;
    DevBuf d_y, d_state, d_trace, d_len, d_part;
    struct Bufs { DevBuf *a, *b, *c, *d, *e; ~Bufs() { a->release(); b->release(); c->release(); d->release(); e->release(); } } bufs{&d_y, &d_state, &d_trace, &d_len, &d_part};
    const bool want_trace = trace && trace_cap > 0;
    if ((rc = d_y.ensure(sizeof(float4) * (size_t)n_moving)) || (rc = d_state.ensure(sizeof(AdaptiveState))) ||
        (rc = d_trace.ensure(sizeof(AdaptiveRow) * (size_t)std::max(1, trace_cap))) || (rc = d_len.ensure(sizeof(int) * 4)) ||
        (rc = d_part.ensure(sizeof(double) * 16 * (size_t)adaptive_partial_records(n_fixed, n_moving)))) return rc;
    AdaptiveState st; std::memset(&st, 0, sizeof(st));
    std::memcpy(st.R, R_inout, sizeof(st.R)); std::memcpy(st.T, T_inout, sizeof(st.T));
    st.ell = ap.ell_init; st.ell_max = ap.ell_max; st.iter = iter ? *iter : 0; st.status = -1;
    HIP_TRY(hipMemcpyAsync(d_state.p, &st, sizeof(st), hipMemcpyHostToDevice, eng.stream));
    HIP_TRY(hipMemsetAsync(d_len.p, 0, sizeof(int) * 4, eng.stream));
    AdaptiveArgs A; std::memset(&A, 0, sizeof(A));
    A.fixed = fx.rec(); A.moving = mv.rec(); A.nf = n_fixed; A.nm = n_moving;
    A.ybuf = static_cast<float4*>(d_y.p); A.state = static_cast<AdaptiveState*>(d_state.p);
    A.trace = want_trace ? static_cast<AdaptiveRow*>(d_trace.p) : nullptr; A.trace_cap = want_trace ? trace_cap : 0; A.trace_len = static_cast<int*>(d_len.p);
    A.ell_min = ap.ell_min; A.dl_step = ap.dl_step;
    A.P.sigma = ap.sigma; A.P.sp_thres = ap.sp_thres; A.P.c = ap.c; A.P.d = ap.d; A.P.c_ell = ap.c_ell; A.P.c_sigma = ap.c_sigma;
    A.P.min_step = ap.min_step; A.P.eps = ap.eps; A.P.eps_2 = ap.eps_2; A.P.max_iter = ap.max_iter; A.P.skin = 0.f; A.P.skin_alpha = 0.f; A.P.alpha_gamma = 0.f; A.P.first_scale = 1.f; A.P.fuse_refine = 0; A.P.nt_min = 0; A.P.overlap_stop_test = 0; A.P.predict = 0.f; A.P.predict_steps = 0.f; A.P.resort = 0; A.P.adopt_kmax = 0; A.P.adopt_on = 0; A.P.adopt_inject = 0; A.P.adopt_dwell = 0; A.P.colocate = 0;
    A.partials = static_cast<double*>(d_part.p);
    // a few iterations are queued at a time (five small kernels each, the rows of the sweeps spread over the device); kernels queued behind a
    // stop return at once, and the host looks at the stop flag between the chunks
    if (ap.max_iter <= 0) { st.status = 0; st.stop = 1; HIP_TRY(hipMemcpy(d_state.p, &st, sizeof(st), hipMemcpyHostToDevice)); }
    for (int done = 0; done < std::max(ap.max_iter, 0);) {
        const int chunk = std::min(8, ap.max_iter - done);
        hipError_t e = launch_adaptive(A, chunk, eng.stream);
        if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("adaptive kernel launch: ") + hipGetErrorString(e));
        int stop = 0;
        HIP_TRY(hipMemcpyAsync(&stop, &static_cast<AdaptiveState*>(d_state.p)->stop, sizeof(int), hipMemcpyDeviceToHost, eng.stream));
        HIP_TRY(hipStreamSynchronize(eng.stream));
        done += chunk;
        if (stop) break;
    }
    HIP_TRY(hipMemcpy(&st, d_state.p, sizeof(st), hipMemcpyDeviceToHost));
    if (st.status != 0) return fail(CVO_ERR_HIP, "adaptive kernel did not complete");
    std::memcpy(R_inout, st.R, sizeof(st.R)); std::memcpy(T_inout, st.T, sizeof(st.T));
    if (ell_out) *ell_out = st.ell;
    if (transform_out) std::memcpy(transform_out, st.transform, sizeof(float) * 12);
    if (iter) *iter = st.iter;
    if (want_trace) {
        int n = 0; HIP_TRY(hipMemcpy(&n, d_len.p, sizeof(int), hipMemcpyDeviceToHost));
        n = std::min(n, trace_cap);
        if (n > 0) HIP_TRY(hipMemcpy(trace, d_trace.p, sizeof(AdaptiveRow) * (size_t)n, hipMemcpyDeviceToHost));
        if (trace_len) *trace_len = n;
    }
    return CVO_OK;
}

int cvo_selftest_cubic_step(int device, int n, const float* coef_minstep, float* step_out) { return selftest(device, 0, n, coef_minstep, 5, step_out, 1); }
int cvo_selftest_exp_sek3(int device, int n, const float* omega_v_dt, float* dR_dT_out) { return selftest(device, 1, n, omega_v_dt, 7, dR_dT_out, 12); }
int cvo_selftest_dist_se3(int device, int n, const float* dR_dT, float* dist_out) { return selftest(device, 2, n, dR_dT, 12, dist_out, 1); }
int cvo_selftest_libm(int device, int n, const float* x, float* out6) { return selftest(device, 3, n, x, 1, out6, 6); }
int cvo_selftest_pair_values(int device, const cvo_params* params, float ell, int n, const float* y_g, float* a_out, float* d2_d2c_out) {
    int rc = check_device(device, nullptr); if (rc) return rc;
    if (n <= 0 || !y_g || !a_out || !(ell > 0.f)) return fail(CVO_ERR_INVALID, "bad self-test arguments");
    cvo_params p; if (params) p = *params; else cvo_default_params(&p);
    HIP_TRY(hipSetDevice(device));
    DevBuf din, dout, daux;
    if ((rc = din.ensure(sizeof(float) * (size_t)n * 8)) || (rc = dout.ensure(sizeof(float) * (size_t)n * 4)) || (rc = daux.ensure(sizeof(float) * (size_t)n * 2))) {
        din.release(); dout.release(); daux.release(); return rc;
    }
    hipError_t e = hipMemcpy(din.p, y_g, sizeof(float) * (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_selftest_pairs(static_cast<const float*>(din.p), static_cast<float*>(dout.p), static_cast<float*>(daux.p), n, ell, to_dev(p), nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(a_out, dout.p, sizeof(float) * (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && d2_d2c_out) e = hipMemcpy(d2_d2c_out, daux.p, sizeof(float) * (size_t)n * 2, hipMemcpyDeviceToHost);
    din.release(); dout.release(); daux.release();
    if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("self-test: ") + hipGetErrorString(e));
    return CVO_OK;
}

int cvo_batch_create(const cvo_params* p, int device, int max_pairs, cvo_batch* out) {
    if (!out || max_pairs <= 0) return fail(CVO_ERR_INVALID, "bad argument");
    *out = nullptr;
    std::unique_ptr<cvo_batch_s> b(new cvo_batch_s());
    if (p) b->prm = *p; else cvo_default_params(&b->prm);
    int rc = b->eng.init(device, b->prm); if (rc) { b->eng.destroy(); return rc; }
    b->max_pairs = max_pairs;
    b->eng.defer_pack = true;                                       // hand-overs are packed by the next align launch (Engine::upload_many)
    b->eng.rec_hint = max_pairs + 1;                                // room for the padding record of an uneven shard (cvo_shard_range)
    b->eng.coop_first_scale = 1.75f;                                // few pairs on many workgroups each = loop-closure candidates: wider first lists (launch_impl)
    b->fixed.resize(max_pairs); b->moving.resize(max_pairs);
    b->init_states.resize(max_pairs);
    for (auto& s : b->init_states) fresh_state(s, b->prm.ell);
    *out = b.release();
    return CVO_OK;
}
int cvo_batch_destroy(cvo_batch b) {
    if (!b) return CVO_OK;
    (void)hipSetDevice(b->eng.device);
    if (b->eng.stream) (void)hipStreamSynchronize(b->eng.stream);
    b->fixed.clear(); b->moving.clear();
    b->eng.destroy();
    delete b;
    return CVO_OK;
}
int cvo_batch_set_pair(cvo_batch b, int p, const float* fixed_xyz, const float* fixed_feat, int n_fixed,
                       const float* moving_xyz, const float* moving_feat, int n_moving) {
    if (!b || p < 0 || p >= b->max_pairs) return fail(CVO_ERR_INVALID, "bad pair index");
    if (!b->fixed[p]) b->fixed[p].reset(new Cloud());
    if (!b->moving[p]) b->moving[p].reset(new Cloud());
    int rc = b->eng.upload(*b->fixed[p], fixed_xyz, fixed_feat, n_fixed); if (rc) return rc;
    rc = b->eng.upload(*b->moving[p], moving_xyz, moving_feat, n_moving); if (rc) return rc;
    fresh_state(b->init_states[p], b->prm.ell);
    b->states_dirty = true;
    return CVO_OK;
}
int cvo_batch_set_pairs(cvo_batch b, int first, int count, const float* const* fixed_xyz, const float* const* fixed_feat, const int* n_fixed,
                        const float* const* moving_xyz, const float* const* moving_feat, const int* n_moving) {
    if (!b || first < 0 || count <= 0 || first + count > b->max_pairs) return fail(CVO_ERR_INVALID, "bad pair range");
    if (!fixed_xyz || !fixed_feat || !n_fixed || !moving_xyz || !moving_feat || !n_moving) return fail(CVO_ERR_INVALID, "null argument");
    std::vector<Engine::UploadItem> items; items.reserve(2 * (size_t)count);
    for (int k = 0; k < count; ++k) {
        const int p = first + k;
        if (!b->fixed[p]) b->fixed[p].reset(new Cloud());
        if (!b->moving[p]) b->moving[p].reset(new Cloud());
        items.push_back(Engine::UploadItem{b->fixed[p].get(), fixed_xyz[k], fixed_feat[k], n_fixed[k]});
        items.push_back(Engine::UploadItem{b->moving[p].get(), moving_xyz[k], moving_feat[k], n_moving[k]});
    }
    int rc = b->eng.upload_many(items.data(), (int)items.size()); if (rc) return rc;
    for (int k = 0; k < count; ++k) fresh_state(b->init_states[first + k], b->prm.ell);
    b->states_dirty = true;
    return CVO_OK;
}
int cvo_host_register(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(CVO_ERR_INVALID, "bad argument");
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(CVO_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, ptr, 0) != hipSuccess || d != ptr) {   // the kernels take the caller's own addresses: they must be the device's too
        (void)hipGetLastError(); (void)hipHostUnregister(ptr);
        return fail(CVO_ERR_HIP, "registered memory is not addressable by the device at the caller's address");
    }
    HostRegistry& r = host_registry();
    std::lock_guard<std::mutex> lk(r.mu);
    r.ranges.push_back(HostRange{static_cast<const unsigned char*>(ptr), static_cast<const unsigned char*>(ptr) + bytes});
    return CVO_OK;
}
int cvo_host_unregister(void* ptr) {
    if (!ptr) return fail(CVO_ERR_INVALID, "null pointer");
    HostRegistry& r = host_registry();
    {
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = std::find_if(r.ranges.begin(), r.ranges.end(), [&](const HostRange& g) { return g.lo == static_cast<const unsigned char*>(ptr); });
        if (it == r.ranges.end()) return fail(CVO_ERR_INVALID, "not a range cvo_host_register was given");
        r.ranges.erase(it);
    }
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(CVO_ERR_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e)); }
    return CVO_OK;
}
int cvo_batch_set_state(cvo_batch b, int p, const float R[9], const float T[3], float ell) {
    if (!b || p < 0 || p >= b->max_pairs || !R || !T) return fail(CVO_ERR_INVALID, "bad argument");
    std::memcpy(b->init_states[p].R, R, sizeof(float) * 9); std::memcpy(b->init_states[p].T, T, sizeof(float) * 3);
    b->init_states[p].ell = ell;
    b->states_dirty = true;
    return CVO_OK;
}
int cvo_batch_set_workgroups(cvo_batch b, int workgroups_per_pair) {
    if (!b || workgroups_per_pair < 0) return fail(CVO_ERR_INVALID, "bad argument");
    b->eng.wg_request = workgroups_per_pair; return CVO_OK;
}
int cvo_batch_set_max_workgroups(cvo_batch b, int max_workgroups) {
    if (!b || max_workgroups < 0) return fail(CVO_ERR_INVALID, "bad argument");
    b->eng.max_wgs = max_workgroups; return CVO_OK;
}
int cvo_batch_set_adoption(cvo_batch b, int on) { if (!b) return fail(CVO_ERR_INVALID, "null batch"); b->eng.adopt = on != 0; return CVO_OK; }
int cvo_batch_last_adoptions(cvo_batch b, int* pairs_helped) {
    if (!b || !pairs_helped) return fail(CVO_ERR_INVALID, "null argument");
    int rc = b->eng.wait(); if (rc) return rc;
    const PairState* r = b->eng.results(); int n = 0;
    for (int i = 0; i < b->last_n; ++i) n += r[i].joined_at > 0 ? 1 : 0;
    *pairs_helped = n; return CVO_OK;
}
int cvo_batch_last_adoption_retractions(cvo_batch b, int* retractions) {
    if (!b || !retractions) return fail(CVO_ERR_INVALID, "null argument");
    int rc = b->eng.wait(); if (rc) return rc;
    const PairState* r = b->eng.results(); int n = 0;
    for (int i = 0; i < b->last_n; ++i) n += r[i].adopt_retracted;
    *retractions = n; return CVO_OK;
}
int cvo_batch_reset_states(cvo_batch b) { if (!b) return fail(CVO_ERR_INVALID, "null batch"); b->states_dirty = true; return CVO_OK; }

int cvo_batch_align_async(cvo_batch b, int n_pairs, void* stream) {
    if (!b || n_pairs <= 0 || n_pairs > b->max_pairs) return fail(CVO_ERR_INVALID, "bad pair count");
    std::vector<Engine::PairIn> pairs(n_pairs);
    for (int i = 0; i < n_pairs; ++i) pairs[i] = Engine::PairIn{b->fixed[i].get(), b->moving[i].get()};
    int rc = b->eng.launch(pairs, b->init_states.data(), b->states_dirty, static_cast<hipStream_t>(stream), false, 0);
    if (rc) return rc;
    b->states_dirty = false;      // device states now evolve launch to launch (warm start) until reset
    b->last_n = n_pairs;
    return CVO_OK;
}
int cvo_batch_wait(cvo_batch b, cvo_pair_result* results, int n) {
    if (!b) return fail(CVO_ERR_INVALID, "null batch");
    int rc = b->eng.wait(); if (rc) return rc;
    if (results) {
        if (n > b->last_n) return fail(CVO_ERR_INVALID, "more results requested than pairs launched");
        const PairState* r = b->eng.results();
        for (int i = 0; i < n; ++i) {
            std::memcpy(results[i].transform, r[i].transform, sizeof(float) * 12);
            std::memcpy(results[i].R, r[i].R, sizeof(float) * 9); std::memcpy(results[i].T, r[i].T, sizeof(float) * 3);
            results[i].ell = r[i].ell; results[i].iter = r[i].iter; results[i].A_nonzero = r[i].A_nonzero;
            results[i].iterations_run = r[i].iterations_run; results[i].status = r[i].status;
            results[i].rebuilds = r[i].rebuilds; results[i].dense_fallbacks = r[i].dense_fallbacks;
        }
    }
    return CVO_OK;
}
int cvo_batch_done(cvo_batch b, int* done) {
    if (!b || !done) return fail(CVO_ERR_INVALID, "null argument");
    *done = 1;
    if (!b->eng.launched) return CVO_OK;
    HIP_TRY(hipSetDevice(b->eng.device));
    const hipError_t e = hipEventQuery(b->eng.ev1);                 // recorded right behind the launch on its stream
    if (e == hipErrorNotReady) { *done = 0; (void)hipGetLastError(); return CVO_OK; }
    if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    return CVO_OK;
}
int cvo_batch_last_launch(cvo_batch b, float* kernel_ms, long long* iterations_total, long long* candidates_total) {
    if (!b) return fail(CVO_ERR_INVALID, "null batch");
    if (kernel_ms) *kernel_ms = b->eng.last_ms;
    long long it = 0, ca = 0;
    const PairState* r = b->eng.results();
    for (int i = 0; i < b->last_n; ++i) { it += r[i].iterations_run; ca += r[i].candidates_total; }
    if (iterations_total) *iterations_total = it;
    if (candidates_total) *candidates_total = ca;
    return CVO_OK;
}
int cvo_batch_last_nonzeros(cvo_batch b, long long* nonzeros_total) {
    if (!b || !nonzeros_total) return fail(CVO_ERR_INVALID, "null argument");
    long long nz = 0;
    const PairState* r = b->eng.results();
    for (int i = 0; i < b->last_n; ++i) nz += r[i].nonzeros_total;
    *nonzeros_total = nz;
    return CVO_OK;
}
int cvo_batch_last_pair_seconds(cvo_batch b, int n, double* seconds) {
    if (!b || !seconds || n <= 0 || n > b->last_n) return fail(CVO_ERR_INVALID, "bad argument");
    const PairState* r = b->eng.results();
    for (int i = 0; i < n; ++i) seconds[i] = 1e-8 * (double)r[i].clk_ticks;      // 100 MHz ticks workgroup 0 spent on the pair
    return CVO_OK;
}
int cvo_batch_last_pair_spans(cvo_batch b, int n, double* start_s, double* end_s, int* joined_at) {
    if (!b || !start_s || !end_s || n <= 0 || n > b->last_n) return fail(CVO_ERR_INVALID, "bad argument");
    const PairState* r = b->eng.results();
    for (int i = 0; i < n; ++i) {
        start_s[i] = 1e-8 * (double)r[i].clk_t0; end_s[i] = 1e-8 * (double)(r[i].clk_t0 + r[i].clk_ticks);
        if (joined_at) joined_at[i] = r[i].joined_at;
    }
    return CVO_OK;
}
int cvo_batch_last_tail_seconds(cvo_batch b, double seconds[4]) {
    if (!b || !seconds) return fail(CVO_ERR_INVALID, "null argument");
    const PairState* r = b->eng.results();
    for (int q = 0; q < 4; ++q) { seconds[q] = 0; for (int i = 0; i < b->last_n; ++i) seconds[q] += 1e-8 * (double)r[i].tail_ticks[q]; }
    return CVO_OK;
}
int cvo_batch_last_cull_masks(cvo_batch b, int n, unsigned long long* masks, unsigned long long* predicted) {
    if (!b || !masks || n <= 0 || n > b->last_n) return fail(CVO_ERR_INVALID, "bad argument");
    const PairState* r = b->eng.results();
    for (int i = 0; i < n; ++i) { masks[i] = r[i].cull_mask; if (predicted) predicted[i] = r[i].predict_mask; }
    return CVO_OK;
}
int cvo_batch_last_phase_seconds(cvo_batch b, double seconds[10]) {
    if (!b || !seconds) return fail(CVO_ERR_INVALID, "null argument");
    const PairState* r = b->eng.results();
    for (int q = 0; q < 10; ++q) seconds[q] = 0;
    for (int i = 0; i < b->last_n; ++i) for (int q = 0; q < 10; ++q) seconds[q] += 1e-8 * (double)r[i].phase_ticks[q];
    {   // slot 9 doubles as the measured shader clock in GHz (cycles per 100 MHz tick), averaged over pairs
        double cyc = 0, tk = 0;
        for (int i = 0; i < b->last_n; ++i) { cyc += (double)r[i].clk_cycles; tk += (double)r[i].clk_ticks; }
        if (std::getenv("CVO_HIP_REPORT_CLOCK") && tk > 0) std::fprintf(stderr, "[cvo_hip] shader clock %.3f GHz\n", cyc / tk * 0.1);
    }
    return CVO_OK;
}
int cvo_batch_compute_innerproduct_lc(cvo_batch b, int n, const float* prior_tran, const float* lc_prior_tran, const float* lc_prior_tran_2,
                                      cvo_lc_scores* out) {
    if (!b || !prior_tran || !lc_prior_tran || !lc_prior_tran_2 || !out) return fail(CVO_ERR_INVALID, "null argument");
    if (n <= 0 || n > b->last_n) return fail(CVO_ERR_INVALID, "more pairs than the last launch aligned");
    int rc = b->eng.wait(); if (rc) return rc;
    const PairState* res = b->eng.results();
    std::vector<Engine::ScoreReq> rq((size_t)n * 8);
    for (int i = 0; i < n; ++i) {
        const Cloud* fx = b->fixed[i].get(); const Cloud* mv = b->moving[i].get();
        if (!fx || !mv || fx->n <= 0 || mv->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "compute_innerproduct_lc: empty cloud in the batch");
        const float* lc_tran = res[i].transform;                     // lc_post = result.transform, keyframe_graph.cpp:702
        const float ell = res[i].ell;                                // the ell align() left behind (Q1), cvo.cpp:395
        Engine::ScoreReq* q = &rq[(size_t)i * 8];
        q[0] = {mv, prior_tran + 12 * i, fx, false, ell};            // cvo.cpp:539
        q[1] = {mv, lc_prior_tran + 12 * i, fx, false, ell};         // cvo.cpp:541
        q[2] = {mv, nullptr, fx, false, ell};                        // cvo.cpp:543
        q[3] = {mv, lc_tran, fx, false, ell};                        // cvo.cpp:545
        q[4] = {fx, nullptr, fx, false, ell};                        // cvo.cpp:550
        q[5] = {mv, nullptr, mv, false, ell};                        // cvo.cpp:551
        q[6] = {mv, lc_tran, fx, true, ell};                         // cvo.cpp:555
        q[7] = {mv, lc_prior_tran_2 + 12 * i, fx, true, ell};        // cvo.cpp:558
    }
    std::vector<double> r((size_t)n * 8 * 24);
    rc = b->eng.score_many(rq.data(), n * 8, reinterpret_cast<double (*)[24]>(r.data())); if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const double (*ri)[24] = reinterpret_cast<const double (*)[24]>(r.data() + (size_t)i * 8 * 24);
        cvo_lc_scores& o = out[i];
        finish_inn_p(ri[0], &o.inn_prior); finish_inn_p(ri[1], &o.inn_lc_prior); finish_inn_p(ri[2], &o.inn_pre); finish_inn_p(ri[3], &o.inn_post);
        finish_inn_p(ri[4], &o.inn_fixed_pcd); finish_inn_p(ri[5], &o.inn_moving_pcd);
        o.cos_angle = o.inn_post.value / (sqrtf(o.inn_fixed_pcd.value) * sqrtf(o.inn_moving_pcd.value));   // cvo.cpp:552
        o.inliers_svd = (int)ri[6][1];
        finish_hessian(ri[6] + 2, o.inliers_svd, o.post_hessian);
        o.inliers_pnpransac = (int)ri[7][1];
        const bool reject = (o.inn_post.value <= o.inn_pre.value) || (o.inn_post.value <= o.inn_lc_prior.value) ||
                            (o.inn_post.value <= o.inn_prior.value) || o.cos_angle < 0.1f;          // keyframe_graph.cpp:711-712
        o.accept = reject ? 0 : 1;
    }
    return CVO_OK;
}
// The tracker's score block (cvo.cpp:475-503) for every pair of the last launch, each with its own align() result as `tran` and
// the ell that align() left behind (Q1).  enqueue: one launch behind the align launch on its stream, transforms read from the
// device-resident states, nothing waits; results: waits and finishes the sums on the host.
int cvo_batch_enqueue_innerproduct(cvo_batch b, int n) {
    if (!b) return fail(CVO_ERR_INVALID, "null argument");
    if (n <= 0 || n > b->last_n || !b->eng.launched) return fail(CVO_ERR_INVALID, "more pairs than the last launch aligned");
    std::vector<Engine::ScoreReq> rq((size_t)n * 5);
    for (int i = 0; i < n; ++i) {
        const Cloud* fx = b->fixed[i].get(); const Cloud* mv = b->moving[i].get();
        if (!fx || !mv || fx->n <= 0 || mv->n <= 0) return fail(CVO_ERR_EMPTY_CLOUD, "compute_innerproduct: empty cloud in the batch");
        Engine::ScoreReq* q = &rq[(size_t)i * 5];
        q[0] = {mv, nullptr, fx, false, 0.f, i, false};              // cvo.cpp:489
        q[1] = {mv, nullptr, fx, false, 0.f, i, true};               // cvo.cpp:491
        q[2] = {fx, nullptr, fx, false, 0.f, i, false};              // cvo.cpp:496
        q[3] = {mv, nullptr, mv, false, 0.f, i, false};              // cvo.cpp:497
        q[4] = {mv, nullptr, fx, true, 0.f, i, true};                // cvo.cpp:500
    }
    return b->eng.score_enqueue(rq.data(), n * 5, b->eng.last_stream);
}
int cvo_batch_set_tail_scores(cvo_batch b, int on) { if (!b) return fail(CVO_ERR_INVALID, "null batch"); b->eng.tail_scores = on != 0; return CVO_OK; }
namespace {
// the score blocks the last launch answered in its tail; whatever a pair's workgroup could not answer (PairDesc::score_out[23]) is
// computed by the score kernel now, all missing requests of the batch in one launch
int collect_tail_scores(cvo_batch b, int n, double* r /* n x 5 x 24 */) {
    if (n <= 0 || n > b->last_n) return fail(CVO_ERR_INVALID, "more pairs than the last launch aligned");
    int rc = b->eng.wait(); if (rc) return rc;
    std::memcpy(r, b->eng.h_tail.p, sizeof(double) * (size_t)n * 5 * 24);
    static const int bit_of[5] = {TAIL_PRE, TAIL_POST, TAIL_FIXED, TAIL_MOVING, TAIL_HESSIAN};
    std::vector<Engine::ScoreReq> rq; std::vector<int> where;
    for (int i = 0; i < n; ++i) {
        const int mask = (int)r[(size_t)i * 120 + 23];
        const Cloud* fx = b->fixed[i].get(); const Cloud* mv = b->moving[i].get();
        const Engine::ScoreReq all[5] = {{mv, nullptr, fx, false, 0.f, i, false}, {mv, nullptr, fx, false, 0.f, i, true}, {fx, nullptr, fx, false, 0.f, i, false},
                                         {mv, nullptr, mv, false, 0.f, i, false}, {mv, nullptr, fx, true, 0.f, i, true}};   // cvo.cpp:489, 491, 496, 497, 500
        for (int q = 0; q < 5; ++q) if (!(mask & bit_of[q])) { rq.push_back(all[q]); where.push_back(i * 5 + q); }
    }
    if (!rq.empty()) {
        std::vector<double> extra(rq.size() * 24);
        rc = b->eng.score_many(rq.data(), (int)rq.size(), reinterpret_cast<double (*)[24]>(extra.data())); if (rc) return rc;
        for (size_t k = 0; k < rq.size(); ++k) std::memcpy(r + (size_t)where[k] * 24, extra.data() + k * 24, sizeof(double) * 24);
    }
    for (int i = 0; i < n; ++i) r[(size_t)i * 120 + 23] = 0.0;
    return CVO_OK;
}
}  // namespace
int cvo_batch_last_tail_answers(cvo_batch b, int n, int* masks) {
    if (!b || !masks || n <= 0 || n > b->last_n) return fail(CVO_ERR_INVALID, "bad argument");
    int rc = b->eng.wait(); if (rc) return rc;
    for (int i = 0; i < n; ++i) masks[i] = b->eng.last_tail ? (int)static_cast<const double*>(b->eng.h_tail.p)[(size_t)i * 120 + 23] : 0;
    return CVO_OK;
}
int cvo_batch_innerproduct_results(cvo_batch b, int n, cvo_track_scores* out) {
    if (!b || !out) return fail(CVO_ERR_INVALID, "null argument");
    std::vector<double> r((size_t)std::max(n, 1) * 5 * 24);
    int rc = (b->eng.last_tail && b->eng.score_pending == 0) ? collect_tail_scores(b, n, r.data())
                                                             : b->eng.score_collect(n * 5, reinterpret_cast<double (*)[24]>(r.data()));
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const double (*ri)[24] = reinterpret_cast<const double (*)[24]>(r.data() + (size_t)i * 5 * 24);
        cvo_track_scores& o = out[i];
        finish_inn_p(ri[0], &o.inn_pre); finish_inn_p(ri[1], &o.inn_post); finish_inn_p(ri[2], &o.inn_fixed_pcd); finish_inn_p(ri[3], &o.inn_moving_pcd);
        o.cos_angle = o.inn_post.value / (sqrtf(o.inn_fixed_pcd.value) * sqrtf(o.inn_moving_pcd.value));           // cvo.cpp:498
        o.inliers = (int)ri[4][1];                                   // cvo.cpp:708 with the caller's counter starting at 0 (local_tracker.cpp:240)
        finish_hessian(ri[4] + 2, o.inliers, o.post_hessian);
    }
    return CVO_OK;
}
int cvo_batch_compute_innerproduct(cvo_batch b, int n, cvo_track_scores* out) {
    int rc = cvo_batch_enqueue_innerproduct(b, n); if (rc) return rc;
    return cvo_batch_innerproduct_results(b, n, out);
}
int cvo_batch_results_to_device(cvo_batch b, void* dst_device, int n, void* stream) {
    if (!b || !dst_device || n <= 0 || n > b->last_n) return fail(CVO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(b->eng.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->eng.last_stream;
    hipError_t e = launch_copy_records(static_cast<const float*>(b->eng.d_records.p), static_cast<float*>(dst_device), n, s);
    if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("record copy kernel launch: ") + hipGetErrorString(e));
    return CVO_OK;
}
int cvo_batch_result_records(cvo_batch b, const void** records_device) {
    if (!b || !records_device) return fail(CVO_ERR_INVALID, "null argument");
    if (!b->eng.launched) return fail(CVO_ERR_INVALID, "no launch yet");
    *records_device = b->eng.d_records.p;
    return CVO_OK;
}

// ---------------------------------------------------------------- multi-GPU: RCCL all-gather of the result records
}  // extern "C"

namespace {
// RCCL is bound at run time (dlopen) so that single-GPU users of the library do not need it at load time.
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value: 128 bytes */ struct Id128, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
};
struct Id128 { char b[CVO_COMM_ID_BYTES]; };
Rccl g_rccl;
int rccl_load() {
    if (g_rccl.lib) return CVO_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return fail(CVO_ERR_HIP, std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : ""));
#define CVO_RCCL_SYM(field, sym) do { *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, sym); if (!g_rccl.field) return fail(CVO_ERR_HIP, std::string("RCCL symbol missing: ") + sym); } while (0)
    CVO_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); CVO_RCCL_SYM(CommInitRank, "ncclCommInitRank"); CVO_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    CVO_RCCL_SYM(CommDestroy, "ncclCommDestroy"); CVO_RCCL_SYM(AllGather, "ncclAllGather"); CVO_RCCL_SYM(GroupStart, "ncclGroupStart");
    CVO_RCCL_SYM(GroupEnd, "ncclGroupEnd"); CVO_RCCL_SYM(GetErrorString, "ncclGetErrorString");
    CVO_RCCL_SYM(CommCount, "ncclCommCount"); CVO_RCCL_SYM(CommUserRank, "ncclCommUserRank");
#undef CVO_RCCL_SYM
    g_rccl.lib = h;
    return CVO_OK;
}
#define RCCL_TRY(expr) do { int r__ = (expr); if (r__ != 0) return fail(CVO_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r__)); } while (0)
constexpr int RCCL_FLOAT = 7;     // ncclFloat32 (rccl.h)
}  // namespace

// `send`: a block of records that all carry CVO_ERR_RANK_FAILED, made when the communicator is: what this rank contributes to a gather whose own block
// could not be prepared (bad arguments, a buffer that would not grow, a fill kernel that would not launch) -- so that the rank still enters the collective.
// own_stream (cvo_comm_set_gather_stream; CVO_HIP_GATHER_STREAM=1 when the communicator is made): every gather of this communicator runs on ONE stream of the
// communicator's own, behind an event of the align launch it follows, and the launch's stream continues behind the gather's event -- the fallback for a RCCL that
// does not take collectives of one communicator from several streams at once.
struct cvo_comm_s { void* comm = nullptr; int n_ranks = 1, rank = 0, device = 0; DevBuf send; int fallback_records = 0;
                    bool own_stream = false; hipStream_t gstream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr; };

struct cvo_multi_s {
    int n_devices = 0, max_pairs = 0, last_n = 0;
    std::vector<int> devices;
    std::vector<cvo_batch> batches;
    std::vector<cvo_comm> comms;
    std::vector<DevBuf> recv;
};

namespace {
constexpr int COMM_FALLBACK_RECORDS = 1024;        // 64 KB; a gather of larger blocks grows it when it is first needed
// (re)make the communicator's block of failure records; the device is current.  Not fatal when it cannot be made: the communicator works without it,
// a failed prepare then returns without entering the collective (and says so).
int comm_fallback_block(cvo_comm_s* c, int records) {
    if (records <= c->fallback_records) return CVO_OK;
    c->fallback_records = 0;
    int rc = c->send.ensure(sizeof(float) * CVO_RESULT_FLOATS * (size_t)records); if (rc) return rc;
    hipError_t e = cvohip::launch_fill_records(static_cast<float*>(c->send.p), 0, records, CVO_ERR_RANK_FAILED, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(CVO_ERR_HIP, std::string("failure-record block: ") + hipGetErrorString(e));
    c->fallback_records = records;
    return CVO_OK;
}
}  // namespace

extern "C" {

int cvo_shard_range(int n_pairs_total, int rank, int n_ranks, int* first, int* count) {
    if (n_pairs_total < 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !first || !count) return fail(CVO_ERR_INVALID, "bad shard arguments");
    const int base = n_pairs_total / n_ranks, rem = n_pairs_total % n_ranks;      // block sizes differ by at most one
    *first = rank * base + std::min(rank, rem); *count = base + (rank < rem ? 1 : 0);
    return CVO_OK;
}
int cvo_comm_unique_id(char id[CVO_COMM_ID_BYTES]) {
    if (!id) return fail(CVO_ERR_INVALID, "null id");
    int rc = rccl_load(); if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(id));
    return CVO_OK;
}
int cvo_comm_create(const char id[CVO_COMM_ID_BYTES], int n_ranks, int rank, int device, cvo_comm* out) {
    if (!id || !out || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(CVO_ERR_INVALID, "bad communicator arguments");
    int rc = check_device(device, nullptr); if (rc) return rc;
    rc = rccl_load(); if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    Id128 uid; std::memcpy(uid.b, id, CVO_COMM_ID_BYTES);
    void* comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, n_ranks, uid, rank));
    cvo_comm_s* c = new cvo_comm_s(); c->comm = comm; c->n_ranks = n_ranks; c->rank = rank; c->device = device;
    if (const char* e = std::getenv("CVO_HIP_GATHER_STREAM")) c->own_stream = std::atoi(e) != 0;
    (void)comm_fallback_block(c, COMM_FALLBACK_RECORDS);
    *out = c;
    return CVO_OK;
}
int cvo_comm_create_all(const int* devices, int n_devices, cvo_comm* out) {
    if (!devices || !out || n_devices <= 0) return fail(CVO_ERR_INVALID, "bad communicator arguments");
    for (int i = 0; i < n_devices; ++i) { int rc = check_device(devices[i], nullptr); if (rc) return rc; }
    int rc = rccl_load(); if (rc) return rc;
    std::vector<void*> comms(n_devices, nullptr);
    RCCL_TRY(g_rccl.CommInitAll(comms.data(), n_devices, devices));
    for (int i = 0; i < n_devices; ++i) {
        cvo_comm_s* c = new cvo_comm_s(); c->comm = comms[i]; c->n_ranks = n_devices; c->rank = i; c->device = devices[i];
        if (const char* e = std::getenv("CVO_HIP_GATHER_STREAM")) c->own_stream = std::atoi(e) != 0;
        if (hipSetDevice(devices[i]) == hipSuccess) (void)comm_fallback_block(c, COMM_FALLBACK_RECORDS);
        out[i] = c;
    }
    return CVO_OK;
}
// what the communicator itself says (ncclCommCount / ncclCommUserRank), not what it was asked for
int cvo_comm_info(cvo_comm c, int* n_ranks, int* rank) {
    if (!c || !c->comm) return fail(CVO_ERR_INVALID, "null communicator");
    int n = 0, r = 0;
    RCCL_TRY(g_rccl.CommCount(c->comm, &n)); RCCL_TRY(g_rccl.CommUserRank(c->comm, &r));
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    return CVO_OK;
}
int cvo_comm_set_gather_stream(cvo_comm c, int on) {
    if (!c) return fail(CVO_ERR_INVALID, "null communicator");
    c->own_stream = on != 0; return CVO_OK;
}
// where the RCCL this process bound lives on disk (dladdr of ncclAllGather): a torch process resolves librccl.so.1 to torch's bundled copy, a plain C++ caller to /opt/rocm's
int cvo_comm_library_path(char* out, int cap) {
    if (!out || cap <= 0) return fail(CVO_ERR_INVALID, "bad argument");
    out[0] = 0;
    int rc = rccl_load(); if (rc) return rc;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(g_rccl.AllGather), &info) && info.dli_fname) { std::strncpy(out, info.dli_fname, (size_t)cap - 1); out[cap - 1] = 0; }
    return CVO_OK;
}
int cvo_comm_destroy(cvo_comm c) {
    if (!c) return fail(CVO_ERR_INVALID, "null communicator");
    (void)hipSetDevice(c->device);
    if (c->gstream) { (void)hipStreamSynchronize(c->gstream); (void)hipStreamDestroy(c->gstream); (void)hipEventDestroy(c->ev_in); (void)hipEventDestroy(c->ev_out); }
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->send.release();
    delete c;
    return CVO_OK;
}
namespace {
// Everything of a gather that can fail happens here, BEFORE the collective is entered: argument checks, the record block (growth,
// padding / status records).  The all-gather itself is posted by gather_post and reads the block in stream order.  When the preparation
// fails, gather_fallback_plan points the plan at the communicator's block of CVO_ERR_RANK_FAILED records instead: the rank still enters the
// collective (its peers read the failure in the gathered table instead of waiting for the rank forever) and the call returns the error.
struct GatherPlan { hipStream_t s = nullptr; float* send = nullptr; };
int gather_prepare(cvo_batch b, cvo_comm c, int n_valid, int n_block, int launch_status, void* recv_device, GatherPlan& plan) {
    if (!b || !c || !recv_device) return fail(CVO_ERR_INVALID, "bad gather arguments");
    if (c->device != b->eng.device) return fail(CVO_ERR_INVALID, "communicator and batch live on different devices");
    return b->eng.padded_records(n_valid, n_block, launch_status, b->last_n, &plan.s, &plan.send);
}
bool gather_fallback_plan(cvo_batch b, cvo_comm c, int n_block, void* recv_device, GatherPlan& plan) {
    if (!c || !c->comm || !recv_device || n_block <= 0) return false;                 // nothing to enter the collective with
    if (hipSetDevice(c->device) != hipSuccess) return false;
    if (n_block > c->fallback_records && comm_fallback_block(c, n_block) != CVO_OK) return false;
    // behind the rank's launch when there is one on this device (the peers' gathers of this step are ordered behind theirs), else the null stream
    plan.s = (b && b->eng.device == c->device) ? (b->eng.launched ? b->eng.last_stream : b->eng.stream) : nullptr;
    plan.send = static_cast<float*>(c->send.p);
    return true;
}
int gather_post(cvo_comm c, int n_block, void* recv_device, const GatherPlan& plan) {
    if (c->own_stream) {
        // launch stream -> event -> the communicator's stream: all-gather -> event -> the launch stream waits for it: whoever waits for the launch's stream
        // (cvo_batch_wait) still finds every rank's records in place, and the collectives of this communicator are posted to one stream, in call order
        if (!c->gstream) { HIP_TRY(hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming)); }
        HIP_TRY(hipEventRecord(c->ev_in, plan.s));
        HIP_TRY(hipStreamWaitEvent(c->gstream, c->ev_in, 0));
        RCCL_TRY(g_rccl.AllGather(plan.send, recv_device, (size_t)n_block * CVO_RESULT_FLOATS, RCCL_FLOAT, c->comm, c->gstream));
        HIP_TRY(hipEventRecord(c->ev_out, c->gstream));
        HIP_TRY(hipStreamWaitEvent(plan.s, c->ev_out, 0));
        return CVO_OK;
    }
    RCCL_TRY(g_rccl.AllGather(plan.send, recv_device, (size_t)n_block * CVO_RESULT_FLOATS, RCCL_FLOAT, c->comm, plan.s));
    return CVO_OK;
}
}  // namespace
int cvo_shard_block(int n_pairs_total, int n_ranks) { return (n_pairs_total < 0 || n_ranks <= 0) ? 0 : (n_pairs_total + n_ranks - 1) / n_ranks; }
int cvo_batch_padded_records(cvo_batch b, int n_valid, int n_block, int launch_status, const void** send_device) {
    if (!b || !send_device) return fail(CVO_ERR_INVALID, "null argument");
    hipStream_t s; float* send;
    int rc = b->eng.padded_records(n_valid, n_block, launch_status, b->last_n, &s, &send); if (rc) return rc;
    *send_device = send;
    return CVO_OK;
}
int cvo_batch_gather_results_padded(cvo_batch b, cvo_comm c, int n_valid, int n_block, int launch_status, void* recv_device) {
    int rc = rccl_load(); if (rc) return rc;
    GatherPlan plan;
    rc = gather_prepare(b, c, n_valid, n_block, launch_status, recv_device, plan);
    if (rc) {                                                       // the rank enters the collective all the same, with failure records
        const std::string msg = g_err;
        if (gather_fallback_plan(b, c, n_block, recv_device, plan) && gather_post(c, n_block, recv_device, plan) == CVO_OK)
            return fail(rc, msg + " (collective entered with CVO_ERR_RANK_FAILED records)");
        return fail(rc, msg + " (collective NOT entered)");
    }
    return gather_post(c, n_block, recv_device, plan);
}
int cvo_batch_gather_results(cvo_batch b, cvo_comm c, int n, void* recv_device) {
    return cvo_batch_gather_results_padded(b, c, n, n, CVO_OK, recv_device);
}
int cvo_gather_results_padded(cvo_batch* batches, cvo_comm* comms, int n_devices, const int* n_valid, int n_block, const int* launch_status, void* const* recv_device) {
    if (!batches || !comms || !recv_device || !n_valid || n_devices <= 0) return fail(CVO_ERR_INVALID, "bad gather arguments");
    int rc = rccl_load(); if (rc) return rc;
    std::vector<GatherPlan> plans(n_devices);
    int first_err = CVO_OK; std::string first_msg;
    for (int i = 0; i < n_devices; ++i) {                            // nothing can fail inside the RCCL group: a rank enqueued without its peers would wait for ever
        rc = gather_prepare(batches[i], comms[i], n_valid[i], n_block, launch_status ? launch_status[i] : CVO_OK, recv_device[i], plans[i]);
        if (rc) {                                                   // this device sends failure records; the error is reported after the collective is posted
            if (!first_err) { first_err = rc; first_msg = g_err + " (collective entered with CVO_ERR_RANK_FAILED records)"; }
            if (!gather_fallback_plan(batches[i], comms[i], n_block, recv_device[i], plans[i])) return fail(rc, first_msg = g_err + " (collective NOT entered)");
        }
    }
    RCCL_TRY(g_rccl.GroupStart());                                  // one process drives several ranks: their calls must be grouped
    for (int i = 0; i < n_devices; ++i) {
        const int r = gather_post(comms[i], n_block, recv_device[i], plans[i]);
        if (r && !first_err) { first_err = r; first_msg = g_err; }
    }
    const int ge = g_rccl.GroupEnd();
    if (first_err) return fail(first_err, first_msg);
    if (ge != 0) return fail(CVO_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge));
    return CVO_OK;
}
int cvo_gather_results(cvo_batch* batches, cvo_comm* comms, int n_devices, int n, void* const* recv_device) {
    if (n_devices <= 0) return fail(CVO_ERR_INVALID, "bad gather arguments");
    std::vector<int> nv(n_devices, n);
    return cvo_gather_results_padded(batches, comms, n_devices, nv.data(), n, nullptr, recv_device);
}
// rank-major gathered blocks (n_ranks x cvo_shard_block records) -> the n_pairs_total records in global pair order
int cvo_compact_records(const float* gathered, int n_pairs_total, int n_ranks, float* out, int* first_error) {
    if (!gathered || !out || n_pairs_total < 0 || n_ranks <= 0) return fail(CVO_ERR_INVALID, "bad argument");
    const int blk = cvo_shard_block(n_pairs_total, n_ranks);
    int err = CVO_OK;
    for (int r = 0; r < n_ranks; ++r) {
        int first = 0, count = 0; cvo_shard_range(n_pairs_total, r, n_ranks, &first, &count);
        const float* src = gathered + (size_t)r * blk * CVO_RESULT_FLOATS;
        for (int k = 0; k < count; ++k) {
            std::memcpy(out + (size_t)(first + k) * CVO_RESULT_FLOATS, src + (size_t)k * CVO_RESULT_FLOATS, sizeof(float) * CVO_RESULT_FLOATS);
            const int st = (int)src[(size_t)k * CVO_RESULT_FLOATS + 15];
            if (st != CVO_OK && err == CVO_OK) err = st;
        }
    }
    if (first_error) *first_error = err;
    return CVO_OK;
}

int cvo_multi_create(const cvo_params* p, const int* devices, int n_devices, int max_pairs_per_device, cvo_multi* out) {
    if (!devices || !out || n_devices <= 0 || max_pairs_per_device <= 0) return fail(CVO_ERR_INVALID, "bad multi arguments");
    std::unique_ptr<cvo_multi_s> m(new cvo_multi_s());
    m->n_devices = n_devices; m->max_pairs = max_pairs_per_device; m->devices.assign(devices, devices + n_devices);
    m->batches.assign(n_devices, nullptr); m->comms.assign(n_devices, nullptr); m->recv.resize(n_devices);
    int rc = cvo_comm_create_all(devices, n_devices, m->comms.data());
    for (int i = 0; i < n_devices && !rc; ++i) rc = cvo_batch_create(p, devices[i], max_pairs_per_device, &m->batches[i]);
    for (int i = 0; i < n_devices && !rc; ++i) {
        rc = (hipSetDevice(devices[i]) == hipSuccess) ? m->recv[i].ensure(sizeof(float) * (size_t)n_devices * max_pairs_per_device * CVO_RESULT_FLOATS)
                                                      : fail(CVO_ERR_HIP, "hipSetDevice failed");
    }
    if (rc) { const std::string msg = g_err; cvo_multi_destroy(m.release()); g_err = msg; return rc; }
    *out = m.release();
    return CVO_OK;
}
int cvo_multi_destroy(cvo_multi m) {
    if (!m) return fail(CVO_ERR_INVALID, "null multi");
    for (int i = 0; i < m->n_devices; ++i) {
        if (m->batches[i]) (void)cvo_batch_destroy(m->batches[i]);
        if (m->comms[i]) (void)cvo_comm_destroy(m->comms[i]);
        (void)hipSetDevice(m->devices[i]); m->recv[i].release();
    }
    delete m;
    return CVO_OK;
}
int cvo_multi_batch(cvo_multi m, int i, cvo_batch* out) {
    if (!m || !out || i < 0 || i >= m->n_devices) return fail(CVO_ERR_INVALID, "bad device index");
    *out = m->batches[i];
    return CVO_OK;
}
// n_pairs[i] pairs on device i (0 = none: the device only contributes padding).  Every device enters the gather whatever its launch
// returned: a launch that could not be made turns into status records, and the first error is reported after the collective is posted.
int cvo_multi_align_async_v(cvo_multi m, const int* n_pairs) {
    if (!m || !n_pairs) return fail(CVO_ERR_INVALID, "null argument");
    int n_block = 0;
    for (int i = 0; i < m->n_devices; ++i) {
        if (n_pairs[i] < 0 || n_pairs[i] > m->max_pairs) return fail(CVO_ERR_INVALID, "bad pair count");
        n_block = std::max(n_block, n_pairs[i]);
    }
    if (n_block <= 0) return fail(CVO_ERR_INVALID, "no pairs on any device");
    std::vector<int> st(m->n_devices, CVO_OK), nv(m->n_devices, 0);
    int first_err = CVO_OK; std::string first_msg;
    for (int i = 0; i < m->n_devices; ++i) {
        if (n_pairs[i] > 0) st[i] = cvo_batch_align_async(m->batches[i], n_pairs[i], nullptr);
        if (st[i] && !first_err) { first_err = st[i]; first_msg = g_err; }
        nv[i] = st[i] ? 0 : n_pairs[i];
    }
    std::vector<void*> recv(m->n_devices);
    for (int i = 0; i < m->n_devices; ++i) recv[i] = m->recv[i].p;
    int rc = cvo_gather_results_padded(m->batches.data(), m->comms.data(), m->n_devices, nv.data(), n_block, st.data(), recv.data()); if (rc) return rc;
    m->last_n = n_block;
    if (first_err) return fail(first_err, first_msg);
    return CVO_OK;
}
int cvo_multi_align_async(cvo_multi m, int n) {
    if (!m || n <= 0 || n > m->max_pairs) return fail(CVO_ERR_INVALID, "bad pair count");
    std::vector<int> nv(m->n_devices, n);
    return cvo_multi_align_async_v(m, nv.data());
}
int cvo_multi_wait(cvo_multi m, int from_device, float* records_out) {
    if (!m || from_device < 0 || from_device >= m->n_devices) return fail(CVO_ERR_INVALID, "bad argument");
    if (m->last_n <= 0) return fail(CVO_ERR_INVALID, "no launch to wait for");
    for (int i = 0; i < m->n_devices; ++i) {
        Engine& e = m->batches[i]->eng;
        if (e.launched) { int rc = cvo_batch_wait(m->batches[i], nullptr, 0); if (rc) return rc; }
        else { HIP_TRY(hipSetDevice(e.device)); HIP_TRY(hipStreamSynchronize(e.stream)); }   // the device only sent padding: its gather ran on the engine's own stream
    }
    if (records_out) {
        HIP_TRY(hipSetDevice(m->devices[from_device]));
        HIP_TRY(hipMemcpy(records_out, m->recv[from_device].p, sizeof(float) * (size_t)m->n_devices * m->last_n * CVO_RESULT_FLOATS, hipMemcpyDeviceToHost));
    }
    return CVO_OK;
}

}  // extern "C"
