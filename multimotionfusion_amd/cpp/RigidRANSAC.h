// RigidRANSAC.h -- C++ shim with the reference's class name (Core/Utils/RigidRANSAC.h:6-32) and the keypoint
// matcher call of PointTracker::addKeypoints (Core/Utils/PointTracker.cpp:100-114), forwarding to the C ABI
// of include/mmf_hip.h.  Eigen::MatrixX3f arguments become row-major n x 3 float arrays on the host.
#pragma once
#include <vector>

#include "RGBDOdometry.h"

class RigidRANSAC {
   public:
    struct Config {
        int iterations;
        float inlier_threshold;
        float inlier_fraction;
    };
    struct Result {
        float transformation[16];           // row-major 4x4 of T_01: p0 ~ T_01 p1
        float error;                        // mean inlier distance of the best model, +inf when none was accepted
        std::vector<unsigned char> inlier;  // over the hash-sorted correspondence order, empty when none was accepted
    };

    RigidRANSAC(int iterations, float inlier_threshold, float inlier_fraction) {
        mmf::check(mmf_ransac_create(iterations, inlier_threshold, inlier_fraction, &r_), "mmf_ransac_create");
    }
    explicit RigidRANSAC(const Config& c) : RigidRANSAC(c.iterations, c.inlier_threshold, c.inlier_fraction) {}
    virtual ~RigidRANSAC() { mmf_ransac_destroy(r_); }
    RigidRANSAC(const RigidRANSAC&) = delete;
    RigidRANSAC& operator=(const RigidRANSAC&) = delete;

    virtual Result estimate(const float* p0, const float* p1, int n, const unsigned char* mask = nullptr) {
        Result res;
        res.inlier.assign(n, 0);
        int has = 0;
        mmf::check(mmf_ransac_estimate(r_, p0, p1, n, mask, res.transformation, &res.error, res.inlier.data(), &has),
                   "mmf_ransac_estimate");
        if (!has) res.inlier.clear();
        return res;
    }

   private:
    mmf_ransac* r_ = nullptr;
};

namespace tracker {
// cv::BFMatcher(cv::NORM_L2, true).match(current, previous, matches) + the distance test of PointTracker.cpp:108,
// on device-resident descriptor rows; trainIdx[q] = row of `previous` or -1
inline void matchDescriptors(mmf::Context& ctx, const float* current_dev, int n_current, const float* previous_dev,
                             int n_previous, int dim, float min_feature_distance, int* trainIdx_dev, float* distance_dev) {
    mmf::check(mmf_match_descriptors(ctx.get(), current_dev, n_current, previous_dev, n_previous, dim, min_feature_distance,
                                     trainIdx_dev, distance_dev),
               "mmf_match_descriptors");
}
}  // namespace tracker
