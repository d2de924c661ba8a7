// RGBDOdometry.h -- C++ shim with the reference's class name and method names
// (Core/Utils/RGBDOdometry.h:31-137) forwarding to the C ABI of include/mmf_hip.h.
//
// Drop-in notes
//   * GPUTexture* arguments of the reference become `const float*` / `const uint8_t*` device
//     images (RGBA32F predictions, interleaved u8 colour) -- the replacement has no GL interop.
//   * Eigen overloads with the reference's exact signatures are provided when <Eigen/Core> is
//     available; the array overloads are always there (this build environment has no Eigen).
//   * Errors: the reference prints and exit(-1)s (Core/Cuda/convenience.cuh:74-83); mmf::check
//     reproduces that.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/mmf_hip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define MMF_HAVE_EIGEN 1
#endif
#endif

namespace mmf {
inline void check(int status, const char* what) {
    if (status != MMF_OK) {
        std::fprintf(stderr, "%s failed: %s\n", what, mmf_last_error());
        std::exit(-1);
    }
}

// one context per device/stream, shared by every odometry / model object (the reference has the
// GPUConfig / GPUSetup singletons, Core/Utils/GPUConfig.h:27, Core/Model/Model.h:96-108)
class Context {
   public:
    explicit Context(int device = 0, void* stream = nullptr, bool private_stream = false) {
        check(mmf_ctx_create(device, stream, private_stream ? 1 : 0, &ctx_), "mmf_ctx_create");
    }
    ~Context() { mmf_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    mmf_ctx* get() const { return ctx_; }
    void synchronize() { check(mmf_ctx_synchronize(ctx_), "mmf_ctx_synchronize"); }

   private:
    mmf_ctx* ctx_ = nullptr;
};
}  // namespace mmf

class RGBDOdometry {
   public:
    static const int NUM_PYRS = MMF_NUM_PYRS;  // RGBDOdometry.h:72
    typedef mmf_odom_stats Stats;  // the public result members (RGBDOdometry.h:62-67) as a value

    RGBDOdometry(mmf::Context& ctx, int width, int height, float cx, float cy, float fx, float fy,
                 unsigned char maskID = 0, float distThresh = 0.10f,
                 float angleThresh = std::sin(20.f * 3.14159254f / 180.f))
        : width_(width), height_(height) {
        (void)maskID;  // not read by the kernels (MASK_RGB_RESIDUAL is undefined in the reference)
        mmf::check(mmf_odom_create(ctx.get(), width, height, cx, cy, fx, fy, distThresh, angleThresh, &o_),
                   "mmf_odom_create");
        refresh();
    }
    virtual ~RGBDOdometry() { mmf_odom_destroy(o_); }
    RGBDOdometry(const RGBDOdometry&) = delete;
    RGBDOdometry& operator=(const RGBDOdometry&) = delete;

    // Model::generateCUDATextures (Core/Model/Model.cpp:359-388)
    void buildDepthPyramid(const float* depth_l0_dev, size_t step = 0) {
        mmf::check(mmf_odom_build_depth_pyramid(o_, depth_l0_dev, step), "mmf_odom_build_depth_pyramid");
    }
    // initICP(depthPyramid, maskPyramid, depthCutoff); nullptr = the pyramid of buildDepthPyramid
    void initICP(const float* const depthPyramid[NUM_PYRS], const size_t steps[NUM_PYRS], const float depthCutoff) {
        mmf::check(mmf_odom_init_icp(o_, depthPyramid, steps, depthCutoff), "mmf_odom_init_icp");
    }
    void initICP(const float* predictedVertices, const float* predictedNormals, const float depthCutoff) {
        mmf::check(mmf_odom_init_icp_from_prediction(o_, predictedVertices, predictedNormals, depthCutoff),
                   "mmf_odom_init_icp_from_prediction");
    }
    void initICPModel(const float* predictedVertices, const float* predictedNormals, const float depthCutoff,
                      const float modelPose[16]) {
        mmf::check(mmf_odom_init_icp_model(o_, predictedVertices, predictedNormals, depthCutoff, modelPose),
                   "mmf_odom_init_icp_model");
    }
    void initRGB(const uint8_t* rgb, size_t step = 0, int channels = 3) {
        mmf::check(mmf_odom_init_rgb(o_, rgb, step, channels), "mmf_odom_init_rgb");
    }
    void initRGBModel(const uint8_t* rgb, size_t step = 0, int channels = 3) {
        mmf::check(mmf_odom_init_rgb_model(o_, rgb, step, channels), "mmf_odom_init_rgb_model");
    }
    void initFirstRGB(const uint8_t* rgb, size_t step = 0, int channels = 3) {
        mmf::check(mmf_odom_init_first_rgb(o_, rgb, step, channels), "mmf_odom_init_first_rgb");
    }

    // trans[3], rot[9] row major, in/out.  icpError/rgbError: device float images or nullptr.
    void getIncrementalTransformation(float trans[3], float rot[9], const bool& rgbOnly, const float& icpWeight,
                                      const bool& pyramid, const bool& fastOdom, const bool& so3,
                                      float* icpErrorSurface = nullptr, float* rgbErrorSurface = nullptr) {
        mmf::check(mmf_odom_get_incremental_transformation(o_, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3,
                                                           icpErrorSurface, rgbErrorSurface),
                   "mmf_odom_get_incremental_transformation");
        refresh();
    }

#ifdef MMF_HAVE_EIGEN
    // the reference's exact Eigen signatures (RGBDOdometry.h:47,56-60)
    void initICPModel(const float* predictedVertices, const float* predictedNormals, const float depthCutoff,
                      const Eigen::Matrix4f& modelPose) {
        Eigen::Matrix<float, 4, 4, Eigen::RowMajor> p = modelPose;
        initICPModel(predictedVertices, predictedNormals, depthCutoff, p.data());
    }
    void getIncrementalTransformation(Eigen::Vector3f& trans, Eigen::Matrix<float, 3, 3, Eigen::RowMajor>& rot,
                                      const bool& rgbOnly, const float& icpWeight, const bool& pyramid,
                                      const bool& fastOdom, const bool& so3, float* icpErrorSurface,
                                      float* rgbErrorSurface) {
        getIncrementalTransformation(trans.data(), rot.data(), rgbOnly, icpWeight, pyramid, fastOdom, so3,
                                     icpErrorSurface, rgbErrorSurface);
        lastA = Eigen::Map<const Eigen::Matrix<double, 6, 6, Eigen::RowMajor>>(stats_.lastA);
        lastb = Eigen::Map<const Eigen::Matrix<double, 6, 1>>(stats_.lastb);
    }
    Eigen::MatrixXd getCovariance() {
        Eigen::Matrix<double, 6, 6, Eigen::RowMajor> c;
        mmf::check(mmf_odom_get_covariance(o_, c.data()), "mmf_odom_get_covariance");
        return c;
    }
    Eigen::Matrix<double, 6, 6, Eigen::RowMajor> lastA;
    Eigen::Matrix<double, 6, 1> lastb;
#else
    void getCovariance(double cov[36]) { mmf::check(mmf_odom_get_covariance(o_, cov), "mmf_odom_get_covariance"); }
    double lastA[36];
    double lastb[6];
#endif

    // public result members of the reference (RGBDOdometry.h:62-67)
    float lastICPError = 0, lastICPCount = 0, lastRGBError = 0, lastRGBCount = 0, lastSO3Error = 0, lastSO3Count = 0;

    mmf_odom* handle() const { return o_; }

   private:
    void refresh() {
        mmf::check(mmf_odom_get_stats(o_, &stats_), "mmf_odom_get_stats");
        lastICPError = stats_.lastICPError;
        lastICPCount = stats_.lastICPCount;
        lastRGBError = stats_.lastRGBError;
        lastRGBCount = stats_.lastRGBCount;
        lastSO3Error = stats_.lastSO3Error;
        lastSO3Count = stats_.lastSO3Count;
#ifndef MMF_HAVE_EIGEN
        for (int i = 0; i < 36; ++i) lastA[i] = stats_.lastA[i];
        for (int i = 0; i < 6; ++i) lastb[i] = stats_.lastb[i];
#endif
    }
    mmf_odom* o_ = nullptr;
    mmf_odom_stats stats_;
    int width_, height_;
};
