// cvo_hip.hpp -- dependency-free C++ mirror of the reference's `cvo::cvo` class
// (thirdparty/cvo/include/cvo.hpp:82-282) over the C ABI of include/cvo_hip.h.
//
// Same member names, argument meaning and error behaviour as the reference; Eigen /
// OpenCV types are replaced by plain structs (3x4 row-major transforms, 6x6 row-major
// Hessian) and the RGB / depth images by the cloud pcd_generator would have produced,
// so this header compiles with nothing but a C++11 compiler and libcvo_hip.so.  The
// Eigen/OpenCV drop-in for the real SLAM tree is include/cvo_adaptor.hpp.
#pragma once
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include "../../include/cvo_hip.h"

namespace cvo_hip {

struct Affine3f { float m[12]; Affine3f() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = 1.f; } };
struct Affine3d { double m[12]; Affine3d() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = 1.0; }
                  Affine3f cast_float() const { Affine3f a; for (int i = 0; i < 12; ++i) a.m[i] = (float)m[i]; return a; } };
struct Matrix6d { double m[36]; };

class inn_p {   // cvo.hpp:52-80
public:
    float value; int num; int num_e;
    void copy(const inn_p& r) { value = r.value; num = r.num; num_e = r.num_e; }
    inn_p(float v, int n, int n_e) : value(v), num(n), num_e(n_e) {}
    inn_p() : value(0), num(0), num_e(0) {}
};

class cvo {
    cvo_handle h_ = nullptr;
    void sync() { cvo_get_transform(h_, transform.m); cvo_get_prev_accum_transform(h_, prev_transform.m, accum_transform.m);
                  int i = 0; cvo_get_init(h_, &i); init = i != 0; cvo_get_iteration_number(h_, &iter); }
    static void to(const cvo_inn_p& a, inn_p& b) { b.value = a.value; b.num = a.num; b.num_e = a.num_e; }
public:
    // public members of the reference (cvo.hpp:139-144)
    bool first_frame = true;
    bool init = false;
    int iter = 0;
    Affine3f transform, prev_transform, accum_transform;

    // cvo(const string& calib_file): the calibration only feeds pcd_generator, which is not behind this boundary
    explicit cvo(const std::string& /*calib_file*/ = std::string(), int device = 0) {
        if (cvo_create(nullptr, device, &h_) != CVO_OK) throw std::runtime_error(std::string("cvo_create: ") + cvo_last_error());
    }
    ~cvo() { cvo_destroy(h_); }
    cvo(const cvo&) = delete; cvo& operator=(const cvo&) = delete;

    void set_pcd(const float* xyz, const float* feat, int n) { if (cvo_set_pcd(h_, xyz, feat, n) == CVO_OK) sync(); }

    // "cvo not initialized !" -> print and return, output untouched (cvo.cpp:463-466, 565-568)
    void match_odometry(const float* xyz, const float* feat, int n, Affine3d& transformd) {
        const int rc = cvo_match_odometry(h_, xyz, feat, n, transformd.m);
        if (rc == CVO_ERR_NOT_INITIALIZED) { std::printf("cvo not initialized !\n"); return; }
        if (rc != CVO_OK) throw std::runtime_error(std::string("match_odometry: ") + cvo_last_error());
        sync();
    }
    void match_keyframe(const float* xyz, const float* feat, int n, Affine3d& transformd) {
        const int rc = cvo_match_keyframe(h_, xyz, feat, n, transformd.m);
        if (rc == CVO_ERR_NOT_INITIALIZED) { std::printf("cvo not initialized !\n"); return; }
        if (rc != CVO_OK) throw std::runtime_error(std::string("match_keyframe: ") + cvo_last_error());
        sync();
    }
    // the reference's own signatures take the images (cvo.cpp:345, 461, 563): point-cloud generation on the GPU
    void set_pcd(const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera& cam) {
        if (cvo_set_pcd_images(h_, bgr8, depth16, width, height, &cam) != CVO_OK) throw std::runtime_error(std::string("set_pcd: ") + cvo_last_error());
        sync();
    }
    // not in the reference: the NEXT frame's images, handed over early (cvo_stage_next_frame in include/cvo_hip.h)
    void stage_next_frame(const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera& cam) {
        if (cvo_stage_next_frame(h_, bgr8, depth16, width, height, &cam) != CVO_OK) throw std::runtime_error(std::string("stage_next_frame: ") + cvo_last_error());
    }
    void match_keyframe(const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera& cam, Affine3d& transformd) {
        const int rc = cvo_match_keyframe_images(h_, bgr8, depth16, width, height, &cam, transformd.m);
        if (rc == CVO_ERR_NOT_INITIALIZED) { std::printf("cvo not initialized !\n"); return; }
        if (rc != CVO_OK) throw std::runtime_error(std::string("match_keyframe: ") + cvo_last_error());
        sync();
    }
    void match_odometry(const unsigned char* bgr8, const unsigned short* depth16, int width, int height, const cvo_camera& cam, Affine3d& transformd) {
        const int rc = cvo_match_odometry_images(h_, bgr8, depth16, width, height, &cam, transformd.m);
        if (rc == CVO_ERR_NOT_INITIALIZED) { std::printf("cvo not initialized !\n"); return; }
        if (rc != CVO_OK) throw std::runtime_error(std::string("match_odometry: ") + cvo_last_error());
        sync();
    }
    void align() { if (cvo_align(h_) != CVO_OK) throw std::runtime_error(std::string("align: ") + cvo_last_error()); sync(); }

    void compute_innerproduct(inn_p& inn_pre, inn_p& inn_post, Matrix6d& post_hessian, Affine3f& tran, int& inliers,
                              inn_p& inn_fixed_pcd, inn_p& inn_moving_pcd, float& cos_angle) {
        cvo_inn_p a, b, c, d;
        if (cvo_compute_innerproduct(h_, &a, &b, post_hessian.m, tran.m, &inliers, &c, &d, &cos_angle) != CVO_OK)
            throw std::runtime_error(std::string("compute_innerproduct: ") + cvo_last_error());
        to(a, inn_pre); to(b, inn_post); to(c, inn_fixed_pcd); to(d, inn_moving_pcd);
    }
    void compute_innerproduct_lc(inn_p& inn_prior, inn_p& inn_lc_prior, inn_p& inn_lc_pre, inn_p& inn_lc_post, Matrix6d& post_hessian,
                                 Affine3f& prior_tran, Affine3f& lc_prior_tran, Affine3f& lc_prior_tran_2, Affine3f& lc_tran,
                                 int& inliers_svd, int& inliers_pnpransac, inn_p& inn_fixed_pcd, inn_p& inn_moving_pcd, float& cos_angle) {
        cvo_inn_p a, b, c, d, e, f;
        if (cvo_compute_innerproduct_lc(h_, &a, &b, &c, &d, post_hessian.m, prior_tran.m, lc_prior_tran.m, lc_prior_tran_2.m, lc_tran.m,
                                        &inliers_svd, &inliers_pnpransac, &e, &f, &cos_angle) != CVO_OK)
            throw std::runtime_error(std::string("compute_innerproduct_lc: ") + cvo_last_error());
        to(a, inn_prior); to(b, inn_lc_prior); to(c, inn_lc_pre); to(d, inn_lc_post); to(e, inn_fixed_pcd); to(f, inn_moving_pcd);
    }

    void update_fixed_pcd() { cvo_update_fixed_pcd(h_); }
    void update_previous_pcd() { cvo_update_previous_pcd(h_); }
    void reset_keyframe(Affine3f& odometry) { cvo_reset_keyframe(h_, odometry.m); sync(); }
    void reset_transform(Affine3f& odometry) { cvo_reset_transform(h_, odometry.m); sync(); }
    Affine3f reset_initial(Affine3f& odometry) { Affine3f out; cvo_reset_initial(h_, odometry.m, out.m); return out; }

    void get_fixed_and_moving_number(int& fixed_num, int& moving_num) { cvo_get_fixed_and_moving_number(h_, &fixed_num, &moving_num); }
    void get_iteration_number(int& iteration) { cvo_get_iteration_number(h_, &iteration); }
    void get_A_nonzero(int& nonzero) { cvo_get_A_nonzero(h_, &nonzero); }
};

}  // namespace cvo_hip
