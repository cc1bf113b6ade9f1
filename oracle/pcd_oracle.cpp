/* ============================================================================
 * oracle/pcd_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's point-cloud generator, the stage in front
 * of the alignment hot path: what cvo::set_pcd
 * (thirdparty/cvo/src/cvo.cpp:345-386) runs on an RGB-D frame before align():
 *   load_image        pcd_generator.cpp:618-630   (gray image)
 *   make_pyramid      pcd_generator.cpp:50-143    (3 levels, central differences)
 *   PixelSelector     thirdparty/PixelSelector2.cpp:34-436 (DSO selector: per
 *                     32x32 block gradient-histogram thresholds, hierarchical
 *                     selection with potential `pot`, one re-selection, random
 *                     sub-sampling with the srand(3141592) byte pattern)
 *   get_points_from_pixels  pcd_generator.cpp:456-499  (back-projection)
 *   get_features (type 1)   pcd_generator.cpp:590-612  (B, G, R, dx, dy)
 *
 * Parity status: PARITY UNPINNED -- the reference has no tests or fixtures for
 * this stage and cannot be built here (OpenCV, Eigen absent).  Third-party
 * arithmetic restated from its published definition:
 *   - cv::cvtColor(COLOR_RGB2GRAY) on 8-bit data (OpenCV >= 3.0, README tested
 *     3.3.1; not vendored): dst = (c0*4899 + c1*9617 + c2*1868 + 8192) >> 14
 *     with c0 the FIRST channel in memory (the reference hands a BGR image to
 *     an "RGB" conversion, so blue gets the 0.299 weight);
 *   - glibc rand()/srand() (TYPE_3 additive feedback generator); pinned in
 *     tests/ against this container's libc.
 * ========================================================================== */
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

typedef struct orc_camera { float scaling_factor, fx, fy, cx, cy; } orc_camera;   /* data_type.h:33-39 */

/* glibc srand(seed) + n calls of rand() & 0xFF          PixelSelector2.cpp:36-38 */
void orc_glibc_rand_bytes(unsigned seed, unsigned char* out, long n) {
    std::vector<uint32_t> r(344 + (size_t)n);
    r[0] = seed ? seed : 1;
    for (int i = 1; i < 31; ++i) {
        const long long hi = (int32_t)r[i - 1] / 127773, lo = (int32_t)r[i - 1] % 127773;
        long long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (size_t i = 34; i < 344 + (size_t)n; ++i) r[i] = r[i - 31] + r[i - 3];
    for (long k = 0; k < n; ++k) out[k] = (unsigned char)((r[344 + k] >> 1) & 0xFF);
}

/* cv::cvtColor(image, intensity, COLOR_RGB2GRAY) on the BGR bytes  pcd_generator.cpp:624 */
void orc_gray(const unsigned char* bgr, int w, int h, unsigned char* gray) {
    for (long i = 0; i < (long)w * h; ++i)
        gray[i] = (unsigned char)((bgr[3 * i] * 4899 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 1868 + (1 << 13)) >> 14);
}

namespace {

struct Frame {                     // cvo::frame, data_type.h:44-69 (what the selector reads)
    int w, h;
    std::vector<float> I[3], dx0, dy0, abs2[3];
};

// make_pyramid, pcd_generator.cpp:50-143
void make_pyramid(const unsigned char* gray, int w, int h, Frame& f) {
    f.w = w; f.h = h;
    int wl = w, hl = h;
    for (int l = 0; l < 3; ++l) { f.I[l].assign((size_t)wl * hl, 0.f); f.abs2[l].assign((size_t)wl * hl, 0.f); wl /= 2; hl /= 2; }
    f.dx0.assign((size_t)w * h, 0.f); f.dy0.assign((size_t)w * h, 0.f);
    for (long i = 0; i < (long)w * h; ++i) f.I[0][i] = gray[i];                      // :83-87
    wl = w; hl = h;
    for (int lvl = 0; lvl < 3; ++lvl) {
        std::vector<float>& I = f.I[lvl];
        if (lvl > 0) {                                                               // :103-118
            const int pw = wl * 2;
            const std::vector<float>& P = f.I[lvl - 1];
            for (int y = 0; y < hl; ++y)
                for (int x = 0; x < wl; ++x)
                    I[x + y * wl] = 0.25f * (P[2 * x + 2 * y * pw] + P[2 * x + 1 + 2 * y * pw] + P[2 * x + 2 * y * pw + pw] + P[2 * x + 1 + 2 * y * pw + pw]);
        }
        for (int idx = wl; idx < wl * (hl - 1); ++idx) {                             // :122-136: flat index, rows are not special
            float dx = 0.5f * (I[idx + 1] - I[idx - 1]);
            float dy = 0.5f * (I[idx + wl] - I[idx - wl]);
            if (!std::isfinite(dx)) dx = 0;
            if (!std::isfinite(dy)) dy = 0;
            if (lvl == 0) { f.dx0[idx] = dx; f.dy0[idx] = dy; }
            f.abs2[lvl][idx] = dx * dx + dy * dy;
        }
        wl /= 2; hl /= 2;
    }
}

int hist_quantil(const int* hist, float below) {                                   // PixelSelector2.cpp:59-68
    int th = hist[0] * below + 0.5f;
    for (int i = 0; i < 90; ++i) { th -= hist[i + 1]; if (th < 0) return i; }
    return 90;
}

struct Selector {                  // dso::PixelSelector, PixelSelector2.cpp:34-57
    int w, h, w32, h32, current_potential = 3;
    std::vector<unsigned char> random_pattern;
    std::vector<float> ths, ths_smoothed;
    Selector(int w_, int h_) : w(w_), h(h_), w32(w_ / 32), h32(h_ / 32) {
        random_pattern.resize((size_t)w * h);
        orc_glibc_rand_bytes(3141592u, random_pattern.data(), (long)w * h);
        ths.assign((size_t)w32 * h32 + 100, 0.f); ths_smoothed.assign((size_t)w32 * h32 + 100, 0.f);   // zero-initialised, +100 slack: rows past h32 read 0
    }
    void make_hists(const Frame& f) {                                                // :71-134
        std::vector<int> hist(100, 0);
        for (int y = 0; y < h32; ++y)
            for (int x = 0; x < w32; ++x) {
                std::fill(hist.begin(), hist.begin() + 50, 0);
                for (int j = 0; j < 32; ++j)
                    for (int i = 0; i < 32; ++i) {
                        const int it = i + 32 * x, jt = j + 32 * y;
                        if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
                        int g = sqrtf(f.abs2[0][(size_t)it + (size_t)jt * w]);
                        if (g > 48) g = 48;
                        hist[g + 1]++; hist[0]++;
                    }
                ths[x + y * w32] = hist_quantil(hist.data(), 0.5f) + 7;               // setting_minGradHistCut, setting_minGradHistAdd
            }
        // 3x3 box mean of the block thresholds, squared (:103-132).  The thresholds are small integers (7 .. 97), so the float
        // sums are exact and the order in which the reference adds the neighbours does not matter.
        for (int y = 0; y < h32; ++y)
            for (int x = 0; x < w32; ++x) {
                float sum = 0, num = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int xx = x + dx, yy = y + dy;
                        if (xx < 0 || xx >= w32 || yy < 0 || yy >= h32) continue;
                        num++; sum += ths[xx + yy * w32];
                    }
                ths_smoothed[x + y * w32] = (sum / num) * (sum / num);
            }
    }
    // select, PixelSelector2.cpp:286-433, with setting_selectDirectionDistribution = false (the direction table is read but unused)
    void select(const Frame& f, float* map, int pot, float th_factor, int n[3]) {
        const int w1 = w / 2, w2 = w / 4;
        std::memset(map, 0, sizeof(float) * (size_t)w * h);
        const float dw1 = 0.75f, dw2 = dw1 * dw1;
        int n3 = 0, n2 = 0, n4 = 0;
        for (int y4 = 0; y4 < h; y4 += 4 * pot) for (int x4 = 0; x4 < w; x4 += 4 * pot) {
            const int my3 = std::min(4 * pot, h - y4), mx3 = std::min(4 * pot, w - x4);
            int best4 = -1; float val4 = 0;
            for (int y3 = 0; y3 < my3; y3 += 2 * pot) for (int x3 = 0; x3 < mx3; x3 += 2 * pot) {
                const int x34 = x3 + x4, y34 = y3 + y4;
                const int my2 = std::min(2 * pot, h - y34), mx2 = std::min(2 * pot, w - x34);
                int best3 = -1; float val3 = 0;
                for (int y2 = 0; y2 < my2; y2 += pot) for (int x2 = 0; x2 < mx2; x2 += pot) {
                    const int x234 = x2 + x34, y234 = y2 + y34;
                    const int my1 = std::min(pot, h - y234), mx1 = std::min(pot, w - x234);
                    int best2 = -1; float val2 = 0;
                    for (int y1 = 0; y1 < my1; ++y1) for (int x1 = 0; x1 < mx1; ++x1) {
                        const int xf = x1 + x234, yf = y1 + y234, idx = xf + w * yf;
                        if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
                        const float th0 = ths_smoothed[(xf >> 5) + (yf >> 5) * w32];
                        const float th1 = th0 * dw1, th2 = th1 * dw2;
                        const float ag0 = f.abs2[0][idx];
                        if (ag0 > th0 * th_factor) {
                            const float dn = ag0;
                            if (dn > val2) { val2 = dn; best2 = idx; best3 = -2; best4 = -2; }
                        }
                        if (best3 == -2) continue;
                        const float ag1 = f.abs2[1][(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1];
                        if (ag1 > th1 * th_factor) {
                            const float dn = ag1;
                            if (dn > val3) { val3 = dn; best3 = idx; best4 = -2; }
                        }
                        if (best4 == -2) continue;
                        const float ag2 = f.abs2[2][(int)(xf * 0.25f + 0.125) + (int)(yf * 0.25f + 0.125) * w2];
                        if (ag2 > th2 * th_factor) {
                            const float dn = ag2;
                            if (dn > val4) { val4 = dn; best4 = idx; }
                        }
                    }
                    if (best2 > 0) { map[best2] = 1; val3 = 1e10; n2++; }
                }
                if (best3 > 0) { map[best3] = 2; val4 = 1e10; n3++; }
            }
            if (best4 > 0) { map[best4] = 4; n4++; }
        }
        n[0] = n2; n[1] = n3; n[2] = n4;
    }
    // makeMaps, PixelSelector2.cpp:136-282 (the FAST branch is commented out in the reference)
    int make_maps(const Frame& f, float* map, float density, int recursions_left, int* pot_used) {
        float num_have = 0, num_want = density, quotia;
        int ideal = current_potential;
        int n[3];
        select(f, map, current_potential, 1.f, n);
        if (pot_used) *pot_used = current_potential;
        num_have = n[0] + n[1] + n[2];
        quotia = num_want / num_have;
        const float K = num_have * (current_potential + 1) * (current_potential + 1);
        ideal = sqrtf(K / num_want) - 1;
        if (ideal < 1) ideal = 1;
        if (recursions_left > 0 && quotia > 1.25 && current_potential > 1) {
            if (ideal >= current_potential) ideal = current_potential - 1;
            current_potential = ideal;
            return make_maps(f, map, density, recursions_left - 1, pot_used);
        } else if (recursions_left > 0 && quotia < 0.25) {
            if (ideal <= current_potential) ideal = current_potential + 1;
            current_potential = ideal;
            return make_maps(f, map, density, recursions_left - 1, pot_used);
        }
        int num_have_sub = num_have;
        if (quotia < 0.95) {
            const int wh = w * h;
            int rn = 0;
            const unsigned char char_th = 255 * quotia;
            for (int i = 0; i < wh; ++i)
                if (map[i] != 0) {
                    if (random_pattern[rn] > char_th) { map[i] = 0; num_have_sub--; }
                    rn++;
                }
        }
        current_potential = ideal;
        return num_have_sub;
    }
};

}  // namespace

/* pcd_generator::create_pointcloud(1, ...) after load_image: the cloud set_pcd hands to align().
 * xyz: n x 3 (AoS, data_type.h:30), feat: 5 channel-major arrays of `cap` floats (channel c of point i at feat[c*cap + i]),
 * px: n x 2 selected pixel (x, y) (frame::selected_points, pcd_generator.cpp:488-489).  Returns the number of points, or
 * -(needed) if cap is too small.  Optional debug outputs may be NULL: gray (w*h), map (w*h, after sub-sampling),
 * ths_smoothed ((w/32)*(h/32)), info[4] = {potential used by the last select, pixels marked in the final map, makeMaps' return value, 0}. */
int orc_pcd_generate(const unsigned char* bgr, const unsigned short* depth, int w, int h, const orc_camera* cam, int num_want,
                     float* xyz, float* feat, unsigned short* px, int cap,
                     unsigned char* gray_out, float* map_out, float* ths_out, int* info) {
    std::vector<unsigned char> gray((size_t)w * h);
    orc_gray(bgr, w, h, gray.data());                                                // load_image, pcd_generator.cpp:618-630
    Frame f;
    make_pyramid(gray.data(), w, h, f);                                              // select_point, :150
    std::vector<float> map((size_t)w * h);
    Selector sel(w, h);                                                              // :154 (a new selector per frame: potential 3, same pattern)
    sel.make_hists(f);                                                               // makeMaps :185 (gradHistFrame != frame)
    int pot_used = 0;
    const int after = sel.make_maps(f, map.data(), (float)num_want, 1, &pot_used);   // :155
    int marked = 0;
    if (gray_out) std::memcpy(gray_out, gray.data(), gray.size());
    if (map_out) std::memcpy(map_out, map.data(), sizeof(float) * map.size());
    if (ths_out) std::memcpy(ths_out, sel.ths_smoothed.data(), sizeof(float) * (size_t)sel.w32 * sel.h32);
    // get_points_from_pixels (:456-499) and get_features, feature_type 1 (:590-612)
    int idx = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint16_t dep = depth[(size_t)y * w + x];
            if (map[(size_t)y * w + x] != 0) marked++;
            if (map[(size_t)y * w + x] != 0 && dep != 0) {
                if (idx < cap) {
                    float p2 = dep / cam->scaling_factor;
                    float p0 = (x - cam->cx) * p2 / cam->fx;
                    float p1 = (y - cam->cy) * p2 / cam->fy;
                    xyz[3 * idx + 0] = p0; xyz[3 * idx + 1] = p1; xyz[3 * idx + 2] = p2;
                    const size_t i = (size_t)y * w + x;
                    feat[0 * (size_t)cap + idx] = bgr[3 * i + 0];
                    feat[1 * (size_t)cap + idx] = bgr[3 * i + 1];
                    feat[2 * (size_t)cap + idx] = bgr[3 * i + 2];
                    feat[3 * (size_t)cap + idx] = f.dx0[i];
                    feat[4 * (size_t)cap + idx] = f.dy0[i];
                    px[2 * idx + 0] = (unsigned short)x; px[2 * idx + 1] = (unsigned short)y;
                }
                ++idx;
            }
        }
    if (info) { info[0] = pot_used; info[1] = marked; info[2] = after; info[3] = 0; }
    return idx <= cap ? idx : -idx;
}

}  // extern "C"
