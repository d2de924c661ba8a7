// MultiMotionFusion.h -- C++ shims with the reference's class and method names for the surfel
// model (Core/Model/Model.h:120-300 + Core/Model/ModelProjection.h:37-77) and the orchestrator
// (Core/MultiMotionFusion.h:78-160), forwarding to the C ABI of include/mmf_hip.h.
//
// GPUTexture* arguments / getters of the reference become device pointers to dense images
// (see mmf_model_texture for names and formats); there is no OpenGL object behind them.
#pragma once
#include <cstdint>
#include <vector>

#include "RGBDOdometry.h"

// FrameData (Core/FrameData.h:25-43) without OpenCV: device-resident colour / depth of one frame.
struct FrameDataDevice {
    long long timestamp = 0;
    const uint8_t* rgb = nullptr;  // u8 x 3, interleaved, width x height
    const float* depth = nullptr;  // float32 metres, 0 = invalid
};

class Model {
   public:
    static const int MAX_VERTICES = 1024 * 1024;  // Model.cpp:119-126

    struct surfel_t {  // Model.h:247-264, Vertex::SIZE = 48
        float position[3], confidence;
        float colour24, unused, initTime, timestamp;
        float normal[3], radius;
    };

    Model(mmf::Context& ctx, int width, int height, float cx, float cy, float fx, float fy, unsigned char id,
          float confidenceThresh, int maxSurfels = MAX_VERTICES) {
        mmf::check(mmf_model_create(ctx.get(), width, height, cx, cy, fx, fy, id, confidenceThresh, maxSurfels, &m_),
                   "mmf_model_create");
        owned_ = true;
    }
    explicit Model(mmf_model* borrowed) : m_(borrowed), owned_(false) {}
    ~Model() {
        if (owned_) mmf_model_destroy(m_);
    }
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    void initialise(const uint8_t* rgb, const float* depthRaw, const float* depthFiltered, int time, float maxDepth) {
        mmf::check(mmf_model_initialise(m_, rgb, depthRaw, depthFiltered, time, maxDepth), "mmf_model_initialise");
    }
    void predictIndices(int time, float depthCutoff, int timeDelta) {
        mmf::check(mmf_model_predict_indices(m_, time, depthCutoff, timeDelta), "mmf_model_predict_indices");
    }
    void combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta) {
        mmf::check(mmf_model_combined_predict(m_, depthCutoff, time, maxTime, timeDelta), "mmf_model_combined_predict");
    }
    // ModelProjection::synthesizeDepth (ModelProjection.h:49-50); result: texture("depth")
    void synthesizeDepth(float depthCutoff, float confThreshold, int time, int maxTime, int timeDelta) {
        mmf::check(mmf_model_synthesize_depth(m_, depthCutoff, confThreshold, time, maxTime, timeDelta),
                   "mmf_model_synthesize_depth");
    }
    void fuse(const int& time, const uint8_t* rgb, const uint8_t* mask, const float* depthRaw,
              const float* depthFiltered, const float depthCutoff, const float weighting) {
        mmf::check(mmf_model_fuse(m_, time, rgb, mask, depthRaw, depthFiltered, depthCutoff, weighting), "mmf_model_fuse");
    }
    void clean(const int& time, const int timeDelta, const float depthCutoff, const float* depthFiltered,
               const uint8_t* mask, float outlierCoefficient = 3.0f) {
        mmf::check(mmf_model_clean(m_, time, timeDelta, depthCutoff, depthFiltered, mask, outlierCoefficient),
                   "mmf_model_clean");
    }
    void performFillIn(const uint8_t* rawRGB, const float* rawDepth, bool frameToFrameRGB, bool lost) {
        mmf::check(mmf_model_perform_fill_in(m_, rawRGB, rawDepth, frameToFrameRGB, lost), "mmf_model_perform_fill_in");
    }
    bool requiresFillIn(float ratio = 0.75f) {
        int r = 0;
        mmf::check(mmf_model_requires_fill_in(m_, ratio, &r), "mmf_model_requires_fill_in");
        return r != 0;
    }
    unsigned lastCount() const {
        unsigned n = 0;
        mmf::check(mmf_model_count(m_, &n), "mmf_model_count");
        return n;
    }
    void overridePose(const float pose[16]) { mmf::check(mmf_model_set_pose(m_, pose), "mmf_model_set_pose"); }
    void getPose(float pose[16]) const { mmf::check(mmf_model_get_pose(m_, pose), "mmf_model_get_pose"); }
    std::vector<surfel_t> downloadMap() const {  // Model.cpp:1353-1384
        std::vector<surfel_t> out(lastCount());
        unsigned got = 0;
        if (!out.empty())
            mmf::check(mmf_model_download_map(m_, reinterpret_cast<float*>(out.data()), (unsigned)out.size(), &got),
                       "mmf_model_download_map");
        out.resize(got);
        return out;
    }
    // getVertexConfProjection() etc.: device image behind the reference's GPUTexture getter
    const void* texture(const char* name, size_t* bytes = nullptr) const {
        void* p = nullptr;
        size_t b = 0;
        mmf::check(mmf_model_texture(m_, name, &p, &b), "mmf_model_texture");
        if (bytes) *bytes = b;
        return p;
    }
    mmf_model* handle() const { return m_; }

   private:
    mmf_model* m_ = nullptr;
    bool owned_ = false;
};

class MultiMotionFusion {
   public:
    // MultiMotionFusion::MultiMotionFusion (MultiMotionFusion.cpp:21-97); unset options keep the
    // GUI defaults of GUI/MainController.cpp:333-345, 514-517
    MultiMotionFusion(mmf::Context& ctx, int width, int height, float cx, float cy, float fx, float fy,
                      const mmf_fusion_config* cfg = nullptr) {
        mmf::check(mmf_fusion_create(ctx.get(), width, height, cx, cy, fx, fy, cfg, &f_), "mmf_fusion_create");
    }
    ~MultiMotionFusion() { mmf_fusion_destroy(f_); }
    MultiMotionFusion(const MultiMotionFusion&) = delete;
    MultiMotionFusion& operator=(const MultiMotionFusion&) = delete;

    // bool processFrame(const FrameData&, const Eigen::Matrix4f* inPose = 0, float weightMultiplier = 1,
    //                   GroundTruthOdometryInterface* = nullptr, bool bootstrap = false)  (MultiMotionFusion.h:78-80).
    // Like the reference it prints "invalid image data" and returns false for a bad frame and
    // returns false after a regular frame (MultiMotionFusion.cpp:209-212, 853).
    bool processFrame(const FrameDataDevice& frame, const float* inPose = nullptr, const float weightMultiplier = 1.f,
                      const bool bootstrap = false) {
        const int rc = mmf_fusion_process_frame(f_, frame.rgb, frame.depth, frame.timestamp, inPose, weightMultiplier,
                                                bootstrap);
        if (rc == MMF_ERR_INVALID) {
            std::fprintf(stderr, "%s\n", mmf_last_error());
            return false;
        }
        mmf::check(rc, "mmf_fusion_process_frame");
        return false;
    }
    // the same frame step with odom_cfg.init == "kp" (MultiMotionFusion.cpp:312-384): trackTransform is
    // RigidRANSAC::Result::transformation of Model::getLastTrackTransform (row-major 4x4), icpRefine is
    // odom_cfg.icp_refine.  In the reference this is selected by the OdometryConfig passed to the constructor.
    bool processFrame(const FrameDataDevice& frame, const float trackTransform[16], const bool icpRefine,
                      const float weightMultiplier = 1.f) {
        const int rc = mmf_fusion_process_frame_init(f_, frame.rgb, frame.depth, frame.timestamp, trackTransform, icpRefine,
                                                     weightMultiplier);
        if (rc == MMF_ERR_INVALID) {
            std::fprintf(stderr, "%s\n", mmf_last_error());
            return false;
        }
        mmf::check(rc, "mmf_fusion_process_frame_init");
        return false;
    }
    // not in the reference: start the next frame's input-side work (filter, pyramids, gradients) on a second
    // stream while the current frame is still being fused; pass the same frame to processFrame afterwards
    void prefetchFrame(const FrameDataDevice& next) {
        mmf::check(mmf_fusion_prefetch_frame(f_, next.rgb, next.depth), "mmf_fusion_prefetch_frame");
    }
    void getCurrPose(float pose[16]) const { mmf::check(mmf_fusion_get_pose(f_, pose), "mmf_fusion_get_pose"); }
    int getTick() const { return mmf_fusion_tick(f_); }
    Model getBackgroundModel() { return Model(mmf_fusion_model(f_)); }
    mmf_odom* getFrameOdometryHandle() { return mmf_fusion_odometry(f_); }
    mmf_fusion* handle() const { return f_; }

   private:
    mmf_fusion* f_ = nullptr;
};
