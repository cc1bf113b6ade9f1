// cvo_adaptor.hpp -- drop-in replacement of the reference's `cvo::cvo` class
// (thirdparty/cvo/include/cvo.hpp:82-282) that forwards to libcvo_hip.so.
//
// Use: in the CVO-SLAM tree replace `#include "cvo.hpp"` by this header (or install it as
// thirdparty/cvo/include/cvo.hpp), drop thirdparty/cvo/src/cvo.cpp from the `cvo` target and link
// `cvo_hip`.  local_tracker.cpp / keyframe_graph.cpp compile unchanged: every
// public member and signature they use is here (cvo.hpp:139-144, 216-276).
//
// Needs the reference's own dependencies (Eigen3, OpenCV core).  Images -> selected pixels -> cloud
// (the reference's pcd_generator + DSO PixelSelector) runs on the GPU too (cvo_set_pcd_images); define
// CVO_ADAPTOR_CPU_PCD to keep the reference's own CPU pcd_generator in front of the boundary instead
// (it then needs thirdparty/cvo/include/pcd_generator.hpp and its sources in the `cvo` target).
// The build container has neither Eigen nor OpenCV: there this header is compile-checked (syntax, signatures, every call into the
// C ABI) against minimal stand-in declarations of the few Eigen / OpenCV / data_type.h names it touches (tests/stubs/,
// tests/test_adaptor_compiles.py), and its public members are diffed against cvo.hpp:216-276.  The dependency-free twin that is
// compiled AND run is cvo_slam_amd/csrc/cvo_hip.hpp.
//
// Device: the GPU a cvo object lives on is CVO_HIP_DEVICE (environment, default 0) or cvo::cvo::set_default_device(d) before the
// object is made -- the reference's constructor signature has no room for it.
#ifndef CVO_H
#define CVO_H

#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <opencv2/core.hpp>       // cv::Mat, cv::Point2f, cv::FileStorage

#include "data_type.h"        // reference: cvo::frame, cvo::point_cloud, cvo::camera_info
#ifdef CVO_ADAPTOR_CPU_PCD
#include "pcd_generator.hpp"  // reference: image -> point cloud on the CPU
#endif
#include "cvo_hip.h"

namespace cvo {

class inn_p {   // cvo.hpp:52-80
public:
    float value; int num; int num_e;
    void copy(const inn_p& r) { value = r.value; num = r.num; num_e = r.num_e; }
    inn_p(const inn_p& r) : value(r.value), num(r.num), num_e(r.num_e) {}
    inn_p(float v, int n, int n_e) : value(v), num(n), num_e(n_e) {}
    inn_p() {}
};

class cvo {
    cvo_handle h_ = nullptr;
    camera_info cam_info;
    std::unique_ptr<frame> ptr_fixed_fr, ptr_moving_fr, ptr_previous_fr;   // kept for get_*_selected_points
    bool pre_pc_init = false;

    static void to12(const Eigen::Affine3f& a, float m[12]) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m[r * 4 + c] = a.matrix()(r, c); }
    static void from12(const float m[12], Eigen::Affine3f& a) { a = Eigen::Affine3f::Identity(); for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) a.matrix()(r, c) = m[r * 4 + c]; }
    void sync() {
        float t[12], p[12], ac[12];
        cvo_get_transform(h_, t); cvo_get_prev_accum_transform(h_, p, ac);
        from12(t, transform); from12(p, prev_transform); from12(ac, accum_transform);
        int i = 0; cvo_get_init(h_, &i); init = i != 0; cvo_get_iteration_number(h_, &iter);
    }
    // set_pcd's cloud (cvo.cpp:348-366).  Default: the images cross the boundary and the generator runs on the GPU; the
    // selected pixels come back for get_*_selected_points.
    void generate_and_upload(const cv::Mat& RGB_img, const cv::Mat& dep_img, frame* fr, int slot) {
#ifdef CVO_ADAPTOR_CPU_PCD
        // Eigen's N x 5 feature matrix is column-major (data_type.h:75): features.data() already is 5 channel-major arrays of N
        pcd_generator pcd_gen; pcd_gen.set_calib(cam_info);
        point_cloud pc;
        pcd_gen.load_image(RGB_img, dep_img, fr);
        pcd_gen.create_pointcloud(1, fr, &pc);
        static_assert(sizeof(Eigen::Vector3f) == 12, "cloud_t must be 12-byte AoS");
        cvo_set_pcd(h_, pc.num_points ? pc.positions[0].data() : nullptr, pc.features.data(), pc.num_points);
        (void)slot;
#else
        cv::Mat rgb = RGB_img.isContinuous() ? RGB_img : RGB_img.clone(), dep = dep_img.isContinuous() ? dep_img : dep_img.clone();
        const cvo_camera cam = {cam_info.scaling_factor, cam_info.fx, cam_info.fy, cam_info.cx, cam_info.cy};
        if (cvo_set_pcd_images(h_, rgb.data, reinterpret_cast<const unsigned short*>(dep.data), rgb.cols, rgb.rows, &cam) != CVO_OK) {
            std::cerr << "cvo set_pcd: " << cvo_last_error() << "\n"; return;
        }
        fr->image = RGB_img; fr->depth = dep_img; fr->h = rgb.rows; fr->w = rgb.cols;        // pcd_generator.cpp:621-629
        int n = 0; cvo_get_selected_points(h_, slot, nullptr, 0, &n);
        std::vector<unsigned short> px(2 * (size_t)n);
        if (n) cvo_get_selected_points(h_, slot, px.data(), n, &n);
        fr->selected_points.clear();
        for (int i = 0; i < n; ++i) fr->selected_points.push_back(cv::Point2f(px[2 * i], px[2 * i + 1]));   // pcd_generator.cpp:488-489
#endif
    }
    static void to(const cvo_inn_p& a, inn_p& b) { b.value = a.value; b.num = a.num; b.num_e = a.num_e; }
    static int& default_device_ref() { static int d = -1; return d; }
    static int pick_device() {
        if (default_device_ref() >= 0) return default_device_ref();
        const char* e = std::getenv("CVO_HIP_DEVICE");
        return e ? std::atoi(e) : 0;
    }

public:
    bool first_frame = true;          // cvo.hpp:139
    bool init = false;                // cvo.hpp:140
    int iter = 0;                     // cvo.hpp:141
    Eigen::Affine3f transform = Eigen::Affine3f::Identity(), prev_transform = Eigen::Affine3f::Identity(),
                    accum_transform = Eigen::Affine3f::Identity();   // cvo.hpp:142-144
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

    cvo(const std::string& calib_file) : ptr_fixed_fr(new frame()) {   // value-initialised: frame::~frame() delete[]s its pyramid pointers (data_type.h:62-68)     // cvo.cpp:18-71
        cv::FileStorage fSettings(calib_file, cv::FileStorage::READ);
        cam_info.fx = fSettings["Camera.fx"]; cam_info.fy = fSettings["Camera.fy"];
        cam_info.cx = fSettings["Camera.cx"]; cam_info.cy = fSettings["Camera.cy"];
        cam_info.scaling_factor = fSettings["DepthMapFactor"];
        if (cvo_create(nullptr, pick_device(), &h_) != CVO_OK) std::cerr << "cvo_create: " << cvo_last_error() << "\n";
    }
    ~cvo() { cvo_destroy(h_); }
    static void set_default_device(int device) { default_device_ref() = device; }   // GPU of the objects made from now on (-1: CVO_HIP_DEVICE / 0)

    // function_inner_product(cloud_a, cloud_b), cvo.hpp:222 / cvo.cpp:388-459: on clouds the caller holds (the handle's current ell applies)
    const inn_p function_inner_product(point_cloud* cloud_a, point_cloud* cloud_b) {
        cvo_inn_p r = {0.f, 0, 0};
        static_assert(sizeof(Eigen::Vector3f) == 12, "cloud_t must be 12-byte AoS");
        if (cvo_function_inner_product_clouds(h_, cloud_a->num_points ? cloud_a->positions[0].data() : nullptr, cloud_a->features.data(), cloud_a->num_points,
                                              cloud_b->num_points ? cloud_b->positions[0].data() : nullptr, cloud_b->features.data(), cloud_b->num_points, &r) != CVO_OK)
            std::cerr << "cvo function_inner_product: " << cvo_last_error() << "\n";
        return inn_p(r.value, r.num, r.num_e);
    }
    // se3_Hessian(cloud_a, cloud_b, inliers), cvo.hpp:260 / cvo.cpp:620-759
    Eigen::Matrix<double, 6, 6> se3_Hessian(point_cloud* cloud_a, point_cloud* cloud_b, int& inliers) {
        double H[36]; for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1.0 : 0.0;
        if (cvo_se3_hessian_clouds(h_, cloud_a->num_points ? cloud_a->positions[0].data() : nullptr, cloud_a->features.data(), cloud_a->num_points,
                                   cloud_b->num_points ? cloud_b->positions[0].data() : nullptr, cloud_b->features.data(), cloud_b->num_points, H, &inliers) != CVO_OK)
            std::cerr << "cvo se3_Hessian: " << cvo_last_error() << "\n";
        Eigen::Matrix<double, 6, 6> out;
        for (int r = 0; r < 6; ++r) for (int q = 0; q < 6; ++q) out(r, q) = H[r * 6 + q];
        return out;
    }

    void set_pcd(const cv::Mat& RGB_img, const cv::Mat& dep_img) {      // cvo.cpp:345-386
        if (!init) { generate_and_upload(RGB_img, dep_img, ptr_fixed_fr.get(), CVO_SLOT_FIXED); sync(); return; }
        ptr_moving_fr.reset(new frame());
        generate_and_upload(RGB_img, dep_img, ptr_moving_fr.get(), CVO_SLOT_MOVING);
    }
    // NOT a member of the reference's class (optional, off the tracker's call sequence): the NEXT frame's images, handed over while this frame is still
    // being tracked -- its cloud is then generated beside this frame's keyframe alignment and the next frame's set_pcd finds it (cvo_stage_next_frame in
    // include/cvo_hip.h; INTEGRATION.md shows the one line in run_SLAM's loop).  The two cv::Mat must stay alive and unchanged until that set_pcd.
#ifndef CVO_ADAPTOR_CPU_PCD
    void stage_next_frame(const cv::Mat& RGB_img, const cv::Mat& dep_img) {
        if (!RGB_img.isContinuous() || !dep_img.isContinuous()) return;                       // (set_pcd would clone them: nothing to recognise later)
        const cvo_camera cam = {cam_info.scaling_factor, cam_info.fx, cam_info.fy, cam_info.cx, cam_info.cy};
        if (cvo_stage_next_frame(h_, RGB_img.data, reinterpret_cast<const unsigned short*>(dep_img.data), RGB_img.cols, RGB_img.rows, &cam) != CVO_OK)
            std::cerr << "cvo stage_next_frame: " << cvo_last_error() << "\n";
    }
#endif
    void align() { if (cvo_align(h_) != CVO_OK) std::cerr << "cvo align: " << cvo_last_error() << "\n"; sync(); }   // cvo.cpp:763-821

    void match_odometry(const cv::Mat& RGB_img, const cv::Mat& dep_img, Eigen::Affine3d& transformd) {   // cvo.cpp:461-473
        if (init == false) { std::cout << "cvo not initialized !" << "\n"; return; }
        set_pcd(RGB_img, dep_img);
        align();
        transformd = transform.cast<double>();
    }
    void match_keyframe(const cv::Mat& RGB_img, const cv::Mat& dep_img, Eigen::Affine3d& transformd) {   // cvo.cpp:563-576
        match_odometry(RGB_img, dep_img, transformd);
    }

    void compute_innerproduct(inn_p& inn_pre, inn_p& inn_post, Eigen::Matrix<double, 6, 6>& post_hessian, Eigen::Affine3f& tran,
                              int& inliers, inn_p& inn_fixed_pcd, inn_p& inn_moving_pcd, float& cos_angle) {   // cvo.cpp:475-503
        float t[12]; to12(tran, t); double H[36]; cvo_inn_p a, b, c, d;
        if (cvo_compute_innerproduct(h_, &a, &b, H, t, &inliers, &c, &d, &cos_angle) != CVO_OK) { std::cerr << cvo_last_error() << "\n"; return; }
        to(a, inn_pre); to(b, inn_post); to(c, inn_fixed_pcd); to(d, inn_moving_pcd);
        for (int r = 0; r < 6; ++r) for (int q = 0; q < 6; ++q) post_hessian(r, q) = H[r * 6 + q];
    }
    void compute_innerproduct_lc(inn_p& inn_prior, inn_p& inn_lc_prior, inn_p& inn_lc_pre, inn_p& inn_lc_post,
                                 Eigen::Matrix<double, 6, 6>& post_hessian, Eigen::Affine3f& prior_tran, Eigen::Affine3f& lc_prior_tran,
                                 Eigen::Affine3f& lc_prior_tran_2, Eigen::Affine3f& lc_tran, int& inliers_svd, int& inliers_pnpransac,
                                 inn_p& inn_fixed_pcd, inn_p& inn_moving_pcd, float& cos_angle) {               // cvo.cpp:505-561
        float p[12], l1[12], l2[12], lt[12]; to12(prior_tran, p); to12(lc_prior_tran, l1); to12(lc_prior_tran_2, l2); to12(lc_tran, lt);
        double H[36]; cvo_inn_p a, b, c, d, e, f;
        if (cvo_compute_innerproduct_lc(h_, &a, &b, &c, &d, H, p, l1, l2, lt, &inliers_svd, &inliers_pnpransac, &e, &f, &cos_angle) != CVO_OK) {
            std::cerr << cvo_last_error() << "\n"; return;
        }
        to(a, inn_prior); to(b, inn_lc_prior); to(c, inn_lc_pre); to(d, inn_lc_post); to(e, inn_fixed_pcd); to(f, inn_moving_pcd);
        for (int r = 0; r < 6; ++r) for (int q = 0; q < 6; ++q) post_hessian(r, q) = H[r * 6 + q];
    }

    void update_fixed_pcd() { ptr_fixed_fr = std::move(ptr_moving_fr); cvo_update_fixed_pcd(h_); }                 // cvo.cpp:578-582
    void update_previous_pcd() { ptr_previous_fr = std::move(ptr_moving_fr); pre_pc_init = true; cvo_update_previous_pcd(h_); }   // :584-589
    void reset_keyframe(Eigen::Affine3f& odometry) {                                                               // :591-604
        if (!pre_pc_init) ptr_fixed_fr = std::move(ptr_moving_fr);
        else { ptr_fixed_fr = std::move(ptr_previous_fr); ptr_previous_fr = std::move(ptr_moving_fr); }
        float t[12]; to12(odometry, t); cvo_reset_keyframe(h_, t); sync();
    }
    void reset_transform(Eigen::Affine3f& odometry) { float t[12]; to12(odometry, t); cvo_reset_transform(h_, t); sync(); }   // :606-609
    Eigen::Affine3f reset_initial(Eigen::Affine3f& odometry) {                                                     // :611-618
        float t[12], o[12]; to12(odometry, t); cvo_reset_initial(h_, t, o);
        Eigen::Affine3f out; from12(o, out); return out;
    }

    void get_fixed_and_moving_number(int& fixed_num, int& moving_num) { cvo_get_fixed_and_moving_number(h_, &fixed_num, &moving_num); }
    void get_iteration_number(int& iteration) { cvo_get_iteration_number(h_, &iteration); }
    void get_A_nonzero(int& nonzero) { cvo_get_A_nonzero(h_, &nonzero); }
    void get_fixed_frame_selected_points(std::vector<cv::Point2f>& pts) { pts = ptr_fixed_fr->selected_points; }
    void get_moving_frame_selected_points(std::vector<cv::Point2f>& pts) { pts = ptr_moving_fr->selected_points; }
};

}  // namespace cvo
#endif  // CVO_H
