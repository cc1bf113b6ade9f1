// Compile check of include/cvo_adaptor.hpp against the stand-in headers of tests/stubs/ (g++ -fsyntax-only, never linked):
// every public member of the reference's cvo::cvo (thirdparty/cvo/include/cvo.hpp:139-144, 216-276) is used the way
// src/local_tracker.cpp and src/keyframe_graph.cpp use it.
#include "cvo_adaptor.hpp"

void use(cvo::cvo& c, const cv::Mat& rgb, const cv::Mat& dep, cvo::point_cloud* a, cvo::point_cloud* b) {
    Eigen::Affine3d td; Eigen::Affine3f tf; Eigen::Matrix<double, 6, 6> H; cvo::inn_p p0, p1, p2, p3, p4, p5; int i0 = 0, i1 = 0; float cs = 0;
    c.set_pcd(rgb, dep); c.match_odometry(rgb, dep, td); c.match_keyframe(rgb, dep, td); c.align();
    c.compute_innerproduct(p0, p1, H, tf, i0, p2, p3, cs);
    c.compute_innerproduct_lc(p0, p1, p2, p3, H, tf, tf, tf, tf, i0, i1, p4, p5, cs);
    c.update_fixed_pcd(); c.update_previous_pcd(); c.reset_keyframe(tf); c.reset_transform(tf); tf = c.reset_initial(tf);
    const cvo::inn_p r = c.function_inner_product(a, b); (void)r; H = c.se3_Hessian(a, b, i0);
    c.get_fixed_and_moving_number(i0, i1); c.get_iteration_number(i0); c.get_A_nonzero(i0);
    std::vector<cv::Point2f> pts; c.get_fixed_frame_selected_points(pts); c.get_moving_frame_selected_points(pts);
    bool f = c.first_frame; c.first_frame = !f; (void)c.init; (void)c.iter; (void)c.transform; (void)c.prev_transform; (void)c.accum_transform;
    cvo::cvo::set_default_device(0);
}
cvo::cvo* make(const std::string& calib) { return new cvo::cvo(calib); }

// ---- the public interface, member by member, with the types of the reference's declarations (cvo.hpp:216-276, 139-144):
// a mismatch in a return type, a parameter type or constness is a compile error here.
#include <type_traits>
using C = cvo::cvo;
using H66 = Eigen::Matrix<double, 6, 6>;
#define SAME(member, ...) static_assert(std::is_same<decltype(&C::member), __VA_ARGS__>::value, "signature of cvo::cvo::" #member " differs from cvo.hpp")
static_assert(std::is_constructible<C, const std::string&>::value, "cvo(const string& calib_file), cvo.hpp:216");
SAME(function_inner_product, const cvo::inn_p (C::*)(cvo::point_cloud*, cvo::point_cloud*));                                        // cvo.hpp:222
SAME(compute_innerproduct, void (C::*)(cvo::inn_p&, cvo::inn_p&, H66&, Eigen::Affine3f&, int&, cvo::inn_p&, cvo::inn_p&, float&));  // :225-226
SAME(compute_innerproduct_lc, void (C::*)(cvo::inn_p&, cvo::inn_p&, cvo::inn_p&, cvo::inn_p&, H66&, Eigen::Affine3f&, Eigen::Affine3f&, Eigen::Affine3f&,
                                          Eigen::Affine3f&, int&, int&, cvo::inn_p&, cvo::inn_p&, float&));                          // :229-232
SAME(set_pcd, void (C::*)(const cv::Mat&, const cv::Mat&));                                                                           // :238
SAME(match_odometry, void (C::*)(const cv::Mat&, const cv::Mat&, Eigen::Affine3d&));                                                  // :241
SAME(match_keyframe, void (C::*)(const cv::Mat&, const cv::Mat&, Eigen::Affine3d&));                                                  // :244
SAME(update_fixed_pcd, void (C::*)());                                                                                                // :246
SAME(update_previous_pcd, void (C::*)());                                                                                             // :248
SAME(reset_keyframe, void (C::*)(Eigen::Affine3f&));                                                                                  // :250
SAME(reset_transform, void (C::*)(Eigen::Affine3f&));                                                                                 // :252
SAME(reset_initial, Eigen::Affine3f (C::*)(Eigen::Affine3f&));                                                                        // :255
SAME(se3_Hessian, H66 (C::*)(cvo::point_cloud*, cvo::point_cloud*, int&));                                                            // :260
SAME(align, void (C::*)());                                                                                                           // :266
SAME(get_fixed_and_moving_number, void (C::*)(int&, int&));                                                                           // :268
SAME(get_iteration_number, void (C::*)(int&));                                                                                        // :269
SAME(get_A_nonzero, void (C::*)(int&));                                                                                               // :270
SAME(get_fixed_frame_selected_points, void (C::*)(std::vector<cv::Point2f>&));                                                        // :275
SAME(get_moving_frame_selected_points, void (C::*)(std::vector<cv::Point2f>&));                                                       // :276
static_assert(std::is_same<decltype(C::first_frame), bool>::value && std::is_same<decltype(C::init), bool>::value && std::is_same<decltype(C::iter), int>::value,
              "public data members, cvo.hpp:139-141");
static_assert(std::is_same<decltype(C::transform), Eigen::Affine3f>::value && std::is_same<decltype(C::prev_transform), Eigen::Affine3f>::value &&
              std::is_same<decltype(C::accum_transform), Eigen::Affine3f>::value, "public data members, cvo.hpp:142-144");
