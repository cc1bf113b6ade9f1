// mini_tracker.cpp -- replays the call sequence of the reference's LocalTracker
// (src/local_tracker.cpp:228-251 initNewLocalMap, :356-431 update, :506 accept) against
// the dependency-free C++ mirror of cvo::cvo, exactly as the tracker's two objects
// (cvo_odometry, cvo_keyframe; local_tracker.cpp:48-49) are used.
// usage: mini_tracker <dir>    reads <dir>/frame_<k>.bin = int32 n, n*3 f32 xyz, 5*n f32 feat
// prints one line per result: label, 12 transform floats, inn_post value/num, inliers, cos
#include <cstdio>
#include <string>
#include <vector>
#include "../../cvo_slam_amd/csrc/cvo_hip.hpp"

struct Frame { int n; std::vector<float> xyz, feat; };
static bool load(const std::string& path, Frame& f) {
    FILE* fp = std::fopen(path.c_str(), "rb"); if (!fp) return false;
    if (std::fread(&f.n, 4, 1, fp) != 1) { std::fclose(fp); return false; }
    f.xyz.resize((size_t)f.n * 3); f.feat.resize((size_t)f.n * 5);
    bool ok = std::fread(f.xyz.data(), 4, f.xyz.size(), fp) == f.xyz.size() && std::fread(f.feat.data(), 4, f.feat.size(), fp) == f.feat.size();
    std::fclose(fp); return ok;
}
static void report(const char* label, const cvo_hip::Affine3d& t, const cvo_hip::inn_p& post, int inliers, float cosang) {
    std::printf("%s", label);
    for (int i = 0; i < 12; ++i) std::printf(" %.9g", t.m[i]);
    std::printf(" %.9g %d %d %.9g\n", post.value, post.num, inliers, cosang);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: mini_tracker <dir>\n"); return 2; }
    std::vector<Frame> frames;
    for (int k = 0;; ++k) { Frame f; if (!load(std::string(argv[1]) + "/frame_" + std::to_string(k) + ".bin", f)) break; frames.push_back(f); }
    if (frames.size() < 3) { std::fprintf(stderr, "need at least 3 frames\n"); return 2; }
    try {
        cvo_hip::cvo cvo_odometry, cvo_keyframe;                                  // local_tracker.cpp:48-49
        cvo_hip::Affine3d T_odo, T_kf;
        cvo_hip::inn_p pre, post, fx, mv; cvo_hip::Matrix6d H; int inliers = 0; float cosang = 0;
        cvo_odometry.match_odometry(frames[1].xyz.data(), frames[1].feat.data(), frames[1].n, T_odo);   // not initialised: prints, T untouched
        cvo_odometry.set_pcd(frames[0].xyz.data(), frames[0].feat.data(), frames[0].n);                  // :228
        cvo_keyframe.set_pcd(frames[0].xyz.data(), frames[0].feat.data(), frames[0].n);                  // :231
        cvo_odometry.match_odometry(frames[1].xyz.data(), frames[1].feat.data(), frames[1].n, T_odo);    // :233
        cvo_hip::Affine3f tran = T_odo.cast_float();
        cvo_odometry.compute_innerproduct(pre, post, H, tran, inliers, fx, mv, cosang);                  // :251
        report("init_odo", T_odo, post, inliers, cosang);
        cvo_odometry.update_fixed_pcd();                                                                  // :277
        if (cvo_keyframe.first_frame) { cvo_keyframe.first_frame = false; cvo_keyframe.reset_transform(tran); }   // :330-333
        for (size_t k = 2; k < frames.size(); ++k) {
            const Frame& f = frames[k];
            cvo_odometry.match_odometry(f.xyz.data(), f.feat.data(), f.n, T_odo);                         // :356
            tran = T_odo.cast_float(); inliers = 0;
            cvo_odometry.compute_innerproduct(pre, post, H, tran, inliers, fx, mv, cosang);               // :375
            report("odo", T_odo, post, inliers, cosang);
            cvo_odometry.update_fixed_pcd();                                                              // :403
            cvo_hip::Affine3f odo = T_odo.cast_float();
            cvo_keyframe.reset_initial(odo);                                                              // :407
            cvo_keyframe.match_keyframe(f.xyz.data(), f.feat.data(), f.n, T_kf);                          // :415
            tran = T_kf.cast_float(); inliers = 0;
            cvo_keyframe.compute_innerproduct(pre, post, H, tran, inliers, fx, mv, cosang);               // :431
            report("kf", T_kf, post, inliers, cosang);
            cvo_keyframe.update_previous_pcd();                                                           // :506
        }
        int it = 0, nz = 0, nf = 0, nm = 0;
        cvo_keyframe.get_iteration_number(it); cvo_keyframe.get_A_nonzero(nz); cvo_keyframe.get_fixed_and_moving_number(nf, nm);
        std::printf("getters %d %d %d %d\n", it, nz, nf, nm);
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
