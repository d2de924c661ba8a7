// SuperPoint.h -- C++ shim with the class name and call shape the reference takes from the un-vendored
// super_point_inference package (Core/MultiMotionFusion.h:46,366; MultiMotionFusion.cpp:78,233):
//     kp_predictor = std::make_shared<SuperPoint>(keypoint_predictor_path);
//     std::tie(coordinates[i], descriptors[i]) = kp_predictor->getFeatures(img);
// forwarding to the C ABI of include/mmf_hip.h.  Differences a maintainer has to bridge:
//   * the constructor takes the 12 {weight, bias} arrays of SuperPointNet's state dict, not the path of the
//     TorchScript archive (reading that file needs libtorch, which stays on the reference's side);
//   * the image is a device pointer to interleaved u8 (cv::Mat::data after an upload), Eigen::MatrixX2d /
//     MatrixXd become row-major std::vector<double> (n x 2, n x 256), as the reference's Eigen types are not
//     part of this repository's dependencies.
#pragma once
#include <tuple>
#include <vector>

#include "RGBDOdometry.h"

class SuperPoint {
   public:
    // weights[24]: {w, b} of conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa convPb convDa convDb
    SuperPoint(mmf::Context& ctx, const float* const* weights, int max_width, int max_height, int max_keypoints = 4096) {
        mmf::check(mmf_superpoint_create(ctx.get(), weights, max_width, max_height, max_keypoints, &sp_), "mmf_superpoint_create");
        max_kp_ = max_keypoints;
    }
    ~SuperPoint() { mmf_superpoint_destroy(sp_); }
    SuperPoint(const SuperPoint&) = delete;
    SuperPoint& operator=(const SuperPoint&) = delete;

    // (coordinates normalised by (width, height), descriptors), strongest keypoint first
    std::tuple<std::vector<double>, std::vector<double>> getFeatures(const unsigned char* image_dev, int width, int height,
                                                                     int channels, float conf_thresh = 0.015f, int nms_dist = 4,
                                                                     int border = 4) {
        std::vector<int> xy((size_t)max_kp_ * 2);
        std::vector<float> conf((size_t)max_kp_), desc((size_t)max_kp_ * 256);
        int n = 0;
        mmf::check(mmf_superpoint_get_features(sp_, image_dev, width, height, channels, conf_thresh, nms_dist, border, xy.data(),
                                               conf.data(), desc.data(), &n),
                   "mmf_superpoint_get_features");
        std::vector<double> coordinates((size_t)n * 2), descriptors((size_t)n * 256);
        for (int k = 0; k < n; ++k) {
            coordinates[2 * k] = (double)xy[2 * k] / (double)width;
            coordinates[2 * k + 1] = (double)xy[2 * k + 1] / (double)height;
        }
        for (size_t k = 0; k < descriptors.size(); ++k) descriptors[k] = (double)desc[k];
        return std::make_tuple(std::move(coordinates), std::move(descriptors));
    }

   private:
    mmf_superpoint* sp_ = nullptr;
    int max_kp_ = 0;
};
