// stand-in for the compile check of include/cvo_adaptor.hpp (tests/stubs/README.md): the members of the reference's
// thirdparty/cvo/include/data_type.h structs that the adaptor touches (names and types as declared there, data_type.h:30-82)
#pragma once
#include <vector>
#include <Eigen/Core>
#include <opencv2/core.hpp>
#define NUM_FEATURES 5
namespace cvo {
typedef std::vector<Eigen::Vector3f> cloud_t;
struct camera_info { float scaling_factor, fx, fy, cx, cy; };
struct frame { int frame_id; int h, w; cv::Mat image, depth; std::vector<cv::Point2f> selected_points; };
struct point_cloud { int num_points; cloud_t positions; Eigen::Matrix<float, Eigen::Dynamic, NUM_FEATURES> features; float dist_avg; };
}
