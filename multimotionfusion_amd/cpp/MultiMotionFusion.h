// MultiMotionFusion.h -- C++ shims with the reference's class and method names for the surfel model
// (Core/Model/Model.h:120-360 + Core/Model/ModelProjection.h:37-77) and the orchestrator
// (Core/MultiMotionFusion.h:50-300), forwarding to the C ABI of include/mmf_hip.h.
//
// GPUTexture* arguments / getters keep their place in the signatures (GPUTexture.h: a view of a dense device image
// instead of an OpenGL texture).  Eigen / OpenCV types of the reference become plain arrays: poses are row-major
// float[16], FrameData carries raw pointers.  What the front-end pushes per GUI tick (GUI/MainController.cpp:641-670)
// and per frame (:588) compiles against this header; setters of subsystems that stay in the reference (CRF, model
// spawning policy, redetection) are kept as recorded no-ops so those call sites need no #ifdef.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "GPUTexture.h"
#include "RGBDOdometry.h"

// FrameData (Core/FrameData.h:25-43) without OpenCV: one frame in HOST memory, as the log readers deliver it
struct FrameData {
    int64_t timestamp = 0;
    const uint8_t* rgb = nullptr;   // CV_8UC3, width x height
    const float* depth = nullptr;   // CV_32FC1 metres, 0 = invalid
    const uint8_t* mask = nullptr;  // optional CV_8UC1 id image, already mapped to model ids (Segmentation.cpp:89-150)
    bool hasNewLabel = false;       // SegmentationResult::hasNewLabel of that id image
};
// the same frame already resident in HBM (no upload inside processFrame)
struct FrameDataDevice {
    long long timestamp = 0;
    const uint8_t* rgb = nullptr;  // u8 x 3, interleaved, width x height
    const float* depth = nullptr;  // float32 metres, 0 = invalid
    const uint8_t* mask = nullptr;
    bool hasNewLabel = false;
    // not in the reference (a LogReader knows its next frame): the buffers the NEXT call will be given; their
    // sensor-side preparation is then enqueued while this call waits for its pose (mmf_frame::next_rgb / next_depth)
    const uint8_t* nextRgb = nullptr;
    const float* nextDepth = nullptr;
};

// The frame geometry the reference keeps in two process-wide singletons (Core/Utils/Resolution.h:26-67,
// Core/Utils/Intrinsics.h:24-66): set once by the front-end (GUI/MainController.cpp:147-148), read by the constructors.
class Resolution {
   public:
    static const Resolution& getInstance() { return instance(); }
    static void setResolution(int width, int height) { instance().w_ = width, instance().h_ = height; }
    const int& width() const { return check(), w_; }
    const int& height() const { return check(), h_; }
    const int& cols() const { return check(), w_; }
    const int& rows() const { return check(), h_; }
    int numPixels() const { return check(), w_ * h_; }

   private:
    static Resolution& instance() {
        static Resolution r;
        return r;
    }
    void check() const {
        if (w_ <= 0 || h_ <= 0) {
            std::fprintf(stderr, "You haven't initialised the Resolution class!\n");
            std::exit(-1);
        }
    }
    int w_ = 0, h_ = 0;
};
class Intrinsics {
   public:
    static const Intrinsics& getInstance() { return instance(); }
    static void setIntrinics(float fx = 0, float fy = 0, float cx = 0, float cy = 0) {  // (sic: the reference's spelling)
        Intrinsics& i = instance();
        i.fx_ = fx, i.fy_ = fy, i.cx_ = cx, i.cy_ = cy;
        i.check();
    }
    const float& fx() const { return check(), fx_; }
    const float& fy() const { return check(), fy_; }
    const float& cx() const { return check(), cx_; }
    const float& cy() const { return check(), cy_; }

   private:
    static Intrinsics& instance() {
        static Intrinsics i;
        return i;
    }
    void check() const {
        if (fx_ == 0 || fy_ == 0) {
            std::fprintf(stderr, "You haven't initialised the Intrinsics class!\n");
            std::exit(-1);
        }
    }
    float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
};

struct OdometryConfig {  // Core/Model/Model.h:45-61
    std::string init;
    std::string init_frame;
    int init_lvl = 0;
    bool icp_refine = false;
    int segm_lvl = 0;
};
struct SegmentationConfiguration {  // Core/Segmentation/Segmentation.h:72-80 (the segmentation itself stays in the reference)
    std::string mode;
    int sp_size = 16;
};

// Model::getModel()'s return type (Core/Model/Buffers.h:3-6).  The reference hands out the GL names of the vertex buffer that
// holds the surfels (48-byte Vertex records); here the store is three float4 arrays in HBM, handed out in place.
struct OutputBuffer {
    const float* positionConfidence = nullptr;  // device, [count][4]: x y z confidence
    const float* colourTime = nullptr;          // device, [count][4]: colour24-as-float, unused, initTime, timestamp
    const float* normalRadius = nullptr;        // device, [count][4]: nx ny nz radius
    unsigned count = 0;
};

class ModelProjection;

class Model {
   public:
    static const int MAX_VERTICES = 1024 * 1024;  // Model.cpp:119-126
    enum class MatchingType { Drost };            // Model.h:112 (model matching stays in the reference)

    struct surfel_t {  // Model.h:247-264, Vertex::SIZE = 48
        float position[3], confidence;
        float colour24, unused, initTime, timestamp;
        float normal[3], radius;
    };

    // Model(id, confidenceThresh, odom_cfg, enableFillIn, ...) (Model.cpp:147-262): a stand-alone model with its own
    // frame-to-model odometry
    Model(mmf::Context& ctx, int width, int height, float cx, float cy, float fx, float fy, unsigned char id,
          float confidenceThresh, bool enableFillIn = true, int maxSurfels = MAX_VERTICES)
        : width_(width), height_(height), fill_in_(enableFillIn) {
        mmf::check(mmf_model_create(ctx.get(), width, height, cx, cy, fx, fy, id, confidenceThresh, maxSurfels, &m_),
                   "mmf_model_create");
        mmf::check(mmf_odom_create(ctx.get(), width, height, cx, cy, fx, fy, 0.10f, std::sin(20.f * 3.14159254f / 180.f), &o_),
                   "mmf_odom_create");
        owned_ = true;
        identity(lastPose_);
    }
    // a model (and its odometry) owned by a MultiMotionFusion object
    Model(mmf_model* borrowed, mmf_odom* odom, int width, int height, bool fillIn)
        : m_(borrowed), o_(odom), owned_(false), width_(width), height_(height), fill_in_(fillIn) {
        identity(lastPose_);
    }
    ~Model() {
        if (owned_) {
            mmf_model_destroy(m_);
            mmf_odom_destroy(o_);
        }
    }
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    // ----- first frame
    void initialise(GPUTexture* rgb, GPUTexture* depthRaw, GPUTexture* depthFiltered, int time, float maxDepth) {
        mmf::check(mmf_model_initialise(m_, rgb->ptr<uint8_t>(), depthRaw->ptr<float>(), depthFiltered->ptr<float>(), time, maxDepth),
                   "mmf_model_initialise");
    }

    // ----- tracking (Model.h:150-157, Model.cpp:359-433)
    // generateCUDATextures(depth, mask): the filtered depth every model's initICP builds its pyramid from
    // (Model::GPUSetup::depth_tmp; the mask pyramid is never read by the tracker)
    static void generateCUDATextures(GPUTexture* depth, GPUTexture* /*mask*/) { shared_depth() = depth->ptr<float>(); }

    void initICP(bool doFillIn, bool frameToFrameRGB, float depthCutoff, GPUTexture* rgb) {
        float pose[16];
        getPose(pose);
        const bool fill = doFillIn;
        mmf::check(mmf_odom_init_icp_model(o_, (const float*)texture(fill ? "fillVertex" : "vertexConf"),
                                           (const float*)texture(fill ? "fillNormal" : "normalRadius"), depthCutoff, pose),
                   "mmf_odom_init_icp_model");
        const bool fillImage = fill || (frameToFrameRGB && allowsFillIn());
        mmf::check(mmf_odom_init_rgb_model(o_, (const uint8_t*)texture(fillImage ? "fillImage" : "image"), 0, 4),
                   "mmf_odom_init_rgb_model");
        mmf::check(mmf_odom_build_depth_pyramid(o_, shared_depth(), 0), "mmf_odom_build_depth_pyramid");
        mmf::check(mmf_odom_init_icp(o_, nullptr, nullptr, depthCutoff), "mmf_odom_init_icp");
        mmf::check(mmf_odom_init_rgb(o_, rgb->ptr<uint8_t>(), 0, 3), "mmf_odom_init_rgb");
    }

    void performTracking(bool frameToFrameRGB, bool rgbOnly, float icpWeight, bool pyramid, bool fastOdom, bool so3,
                         float maxDepthProcessed, GPUTexture* rgb, int64_t logTimestamp, bool tryFillIn = false) {
        float pose[16];
        getPose(pose);
        std::memcpy(lastPose_, pose, sizeof(pose));  // lastPose = pose (Model.cpp:412)
        initICP(tryFillIn, frameToFrameRGB, maxDepthProcessed, rgb);
        float trans[3] = {pose[3], pose[7], pose[11]};
        float rot[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
        // enableErrorRecording (every Model of the reference is created with it): icpError / rgbError are written
        mmf::check(mmf_odom_get_incremental_transformation(o_, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3,
                                                           const_cast<float*>(getICPErrorTexture()->ptr<float>()),
                                                           const_cast<float*>(getRGBErrorTexture()->ptr<float>())),
                   "mmf_odom_get_incremental_transformation");
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) pose[4 * r + c] = rot[3 * r + c];
            pose[4 * r + 3] = trans[r];
        }
        mmf::check(mmf_model_set_pose(m_, pose), "mmf_model_set_pose");
        timestamp_ns_.push_back(logTimestamp);
    }

    // ----- fusion (Model.h:196-207)
    float computeFusionWeight(float weightMultiplier) const {
        float pose[16], w = 0.f;
        getPose(pose);
        mmf::check(mmf_compute_fusion_weight(pose, lastPose_, weightMultiplier, &w), "mmf_compute_fusion_weight");
        return w;
    }
    void fuse(const int& time, GPUTexture* rgb, GPUTexture* mask, GPUTexture* depthRaw, GPUTexture* depthFiltered,
              const float depthCutoff, const float weightMultiplier) {
        mmf::check(mmf_model_fuse(m_, time, rgb->ptr<uint8_t>(), mask->ptr<uint8_t>(), depthRaw->ptr<float>(),
                                  depthFiltered->ptr<float>(), depthCutoff, computeFusionWeight(weightMultiplier)),
                   "mmf_model_fuse");
    }
    void clean(const int& time, std::vector<float>& graph, const int timeDelta, const float depthCutoff, const bool /*isFern*/,
               GPUTexture* depthFiltered, GPUTexture* mask, float outlierCoefficient = 3.0f) {
        if (!graph.empty()) {  // deformation graph: loop closure stays in the reference
            std::fprintf(stderr, "Model::clean: deformation graphs are not supported on this path\n");
            std::exit(-1);
        }
        mmf::check(mmf_model_clean(m_, time, timeDelta, depthCutoff, depthFiltered->ptr<float>(), mask->ptr<uint8_t>(), outlierCoefficient),
                   "mmf_model_clean");
    }

    // ----- prediction and fill-in (Model.h:208-220)
    bool allowsFillIn() const { return fill_in_; }
    void performFillIn(GPUTexture* rawRGB, GPUTexture* rawDepth, bool frameToFrameRGB, bool lost) {
        if (!fill_in_) return;
        mmf::check(mmf_model_perform_fill_in(m_, rawRGB->ptr<uint8_t>(), rawDepth->ptr<float>(), frameToFrameRGB, lost),
                   "mmf_model_perform_fill_in");
    }
    void combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta, int /*ModelProjection::Prediction*/ = 0) {
        mmf::check(mmf_model_combined_predict(m_, depthCutoff, time, maxTime, timeDelta), "mmf_model_combined_predict");
    }
    void predictIndices(int time, float depthCutoff, int timeDelta) {
        mmf::check(mmf_model_predict_indices(m_, time, depthCutoff, timeDelta), "mmf_model_predict_indices");
    }
    // ModelProjection::synthesizeDepth (ModelProjection.h:49-50); result: texture("depth")
    void synthesizeDepth(float depthCutoff, float confThreshold, int time, int maxTime, int timeDelta) {
        mmf::check(mmf_model_synthesize_depth(m_, depthCutoff, confThreshold, time, maxTime, timeDelta), "mmf_model_synthesize_depth");
    }
    bool requiresFillIn(float ratio = 0.75f) {  // MultiMotionFusion::requiresFillIn(model, ratio)
        if (!fill_in_) return false;
        int r = 0;
        mmf::check(mmf_model_requires_fill_in(m_, ratio, &r), "mmf_model_requires_fill_in");
        return r != 0;
    }

    // ----- getters (Model.h:222-312)
    float getConfidenceThreshold() const { return mmf_model_confidence_threshold(m_); }
    void setConfidenceThreshold(float confThresh) { mmf::check(mmf_model_set_confidence_threshold(m_, confThresh), "setConfidenceThreshold"); }
    void setMaxDepth(float d) { mmf::check(mmf_model_set_max_depth(m_, d), "setMaxDepth"); }
    unsigned int getID() const { return (unsigned)mmf_model_id(m_); }
    unsigned lastCount() const {
        unsigned n = 0;
        mmf::check(mmf_model_count(m_, &n), "mmf_model_count");
        return n;
    }
    void overridePose(const float pose[16]) {  // pose = lastPose = p (Model.h:301-304)
        mmf::check(mmf_model_set_pose(m_, pose), "mmf_model_set_pose");
        std::memcpy(lastPose_, pose, sizeof(lastPose_));
    }
    void getPose(float pose[16]) const { mmf::check(mmf_model_get_pose(m_, pose), "mmf_model_get_pose"); }
    const float* getLastPose() const { return lastPose_; }
#ifdef MMF_HAVE_EIGEN
    Eigen::Matrix4f getPose() const {  // Model.h:292
        Eigen::Matrix<float, 4, 4, Eigen::RowMajor> p;
        getPose(p.data());
        return p;
    }
    void overridePose(const Eigen::Matrix4f& pose) {  // Model.h:301-304
        const Eigen::Matrix<float, 4, 4, Eigen::RowMajor> p = pose;
        overridePose(p.data());
    }
#endif
    std::vector<surfel_t> downloadMap() const {  // Model.cpp:1353-1384
        std::vector<surfel_t> out(lastCount());
        unsigned got = 0;
        if (!out.empty())
            mmf::check(mmf_model_download_map(m_, reinterpret_cast<float*>(out.data()), (unsigned)out.size(), &got),
                       "mmf_model_download_map");
        out.resize(got);
        return out;
    }
    // Model.h:297: the surfel store the last fuse / clean left, in place (valid until this model's next fuse / clean)
    const OutputBuffer& getModel() {
        mmf::check(mmf_model_surfel_arrays(m_, &vbo_.positionConfidence, &vbo_.colourTime, &vbo_.normalRadius, &vbo_.count),
                   "mmf_model_surfel_arrays");
        return vbo_;
    }
    // Model.h:278-284: the error images of the last tracking (R32F, written on the last level-0 iteration) and the
    // confidence-carrying vertex image of the splat -- what Segmentation.cpp:218-219 reads of every model
    GPUTexture* getICPErrorTexture() { return error_texture(icp_tex_, "icp_error"); }
    GPUTexture* getRGBErrorTexture() { return error_texture(rgb_tex_, "rgb_error"); }
    std::vector<float> downloadICPErrorTexture() { return download_f32(getICPErrorTexture()->ptr<float>(), (size_t)width_ * height_); }
    std::vector<float> downloadRGBErrorTexture() { return download_f32(getRGBErrorTexture()->ptr<float>(), (size_t)width_ * height_); }
    std::vector<float> downloadVertexConfTexture() {  // RGBA32F: x y z confidence
        return download_f32(static_cast<const float*>(texture("vertexConf")), (size_t)width_ * height_ * 4);
    }
    RGBDOdometry::Stats getFrameOdometryStats() const {
        RGBDOdometry::Stats s;
        mmf::check(mmf_odom_get_stats(o_, &s), "mmf_odom_get_stats");
        return s;
    }
    // the GPUTexture getters of Model / ModelProjection (Model.h:232-244, ModelProjection.h:52-77) by name:
    // index vertConf colorTime normRad | image vertexConf normalRadius time | depth | fillVertex fillNormal fillImage
    const void* texture(const char* name, size_t* bytes = nullptr) const {
        void* p = nullptr;
        size_t b = 0;
        mmf::check(mmf_model_texture(m_, name, &p, &b), "mmf_model_texture");
        if (bytes) *bytes = b;
        return p;
    }
    size_t getSplatVertexConfTexBytes() const {
        size_t b = 0;
        (void)texture("vertexConf", &b);
        return b;
    }
    GPUTexture getRGBProjection() const { return GPUTexture(texture("image"), width_, height_, GPUTexture::RGBA8, "image"); }
    GPUTexture getVertexConfProjection() const { return GPUTexture(texture("vertexConf"), width_, height_, GPUTexture::RGBA32F, "vertexConf"); }
    GPUTexture getNormalProjection() const { return GPUTexture(texture("normalRadius"), width_, height_, GPUTexture::RGBA32F, "normalRadius"); }
    GPUTexture getTimeProjection() const { return GPUTexture(texture("time"), width_, height_, GPUTexture::R16UI, "time"); }
    GPUTexture getFillInImageTexture() const { return GPUTexture(texture("fillImage"), width_, height_, GPUTexture::RGBA8, "fillImage"); }
    GPUTexture getFillInVertexTexture() const { return GPUTexture(texture("fillVertex"), width_, height_, GPUTexture::RGBA32F, "fillVertex"); }
    GPUTexture getFillInNormalTexture() const { return GPUTexture(texture("fillNormal"), width_, height_, GPUTexture::RGBA32F, "fillNormal"); }
    GPUTexture getSparseIndexTex() const { return GPUTexture(texture("index"), width_, height_, GPUTexture::R32UI, "index"); }
    GPUTexture getSparseVertConfTex() const { return GPUTexture(texture("vertConf"), width_, height_, GPUTexture::RGBA32F, "vertConf"); }
    GPUTexture getSparseColorTimeTex() const { return GPUTexture(texture("colorTime"), width_, height_, GPUTexture::RGBA32F, "colorTime"); }
    GPUTexture getSparseNormalRadTex() const { return GPUTexture(texture("normRad"), width_, height_, GPUTexture::RGBA32F, "normRad"); }
    mmf_model* handle() const { return m_; }
    mmf_odom* odometryHandle() const { return o_; }

   private:
    static const float*& shared_depth() {
        static const float* d = nullptr;
        return d;
    }
    static void identity(float* m) {
        for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f;
    }
    GPUTexture* error_texture(std::unique_ptr<GPUTexture>& t, const char* name) {
        void* p = nullptr;
        size_t b = 0;
        mmf::check(mmf_odom_buffer(o_, name, 0, &p, &b), "mmf_odom_buffer");
        if (!t)
            t.reset(new GPUTexture(p, width_, height_, GPUTexture::R32F, name));
        else
            t->rebind(p);
        return t.get();
    }
    static std::vector<float> download_f32(const float* dev, size_t n) {  // GPUTexture::downloadTexture (cv::Mat in the reference)
        std::vector<float> host(n);
        if (hipMemcpy(host.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) host.clear();
        return host;
    }
    mmf_model* m_ = nullptr;
    mmf_odom* o_ = nullptr;
    bool owned_ = false;
    int width_ = 0, height_ = 0;
    bool fill_in_ = false;
    float lastPose_[16];
    std::vector<int64_t> timestamp_ns_;
    OutputBuffer vbo_;
    std::unique_ptr<GPUTexture> icp_tex_, rgb_tex_;
};

typedef std::shared_ptr<Model> ModelPointer;
typedef std::list<ModelPointer> ModelList;

class MultiMotionFusion {
   public:
    // MultiMotionFusion::MultiMotionFusion (MultiMotionFusion.cpp:21-97); unset options keep the
    // GUI defaults of GUI/MainController.cpp:333-345, 514-517
    MultiMotionFusion(mmf::Context& ctx, int width, int height, float cx, float cy, float fx, float fy,
                      const mmf_fusion_config* cfg = nullptr)
        : width_(width), height_(height) {
        mmf::check(mmf_fusion_create(ctx.get(), width, height, cx, cy, fx, fy, cfg, &f_), "mmf_fusion_create");
        mmf::check(mmf_fusion_get_config(f_, &cfg_), "mmf_fusion_get_config");
    }
    // The reference's own argument list (Core/MultiMotionFusion.h:54-61, .cpp:21-97): the frame geometry comes from the
    // Resolution / Intrinsics singletons, the device context is the process-wide default one (device 0, like the reference's
    // cudaGetDeviceProperties(&prop, 0)).  Arguments of subsystems that stay in the reference (loop closure, relocalisation,
    // ferns, model matching, keypoint predictor path, segmentation mode) are recorded; closeLoops must be false, as the
    // front-end passes it (GUI/MainController.cpp:361, 514-515: openLoop is hard-wired).
    MultiMotionFusion(const int timeDelta = 200, const int countThresh = 35000, const float errThresh = 5e-05f, const float covThresh = 1e-05f,
                      const bool closeLoops = false, const bool iclnuim = false, const bool reloc = false, const float photoThresh = 115,
                      const float initConfidenceGlobal = 4, const float initConfidenceObject = 2, const float depthCut = 3,
                      const float icpThresh = 10, const bool fastOdom = false, const float fernThresh = 0.3095f, const bool so3 = true,
                      const bool frameToFrameRGB = false, const unsigned modelSpawnOffset = 20,
                      const Model::MatchingType /*matchingType*/ = Model::MatchingType::Drost, const std::string& exportDirectory = "",
                      const bool exportSegmentationResults = false, const std::string keypoint_predictor_path = {},
                      const OdometryConfig& odom_cfg = {}, const SegmentationConfiguration& segm_cfg = {})
        : width_(Resolution::getInstance().width()), height_(Resolution::getInstance().height()), odom_cfg_(odom_cfg), segm_cfg_(segm_cfg),
          exportDirectory_(exportDirectory) {
        if (closeLoops) {
            std::fprintf(stderr, "MultiMotionFusion: loop closure is not part of this path (the front-end runs with openLoop)\n");
            std::exit(-1);
        }
        mmf_fusion_config cfg;
        mmf::check(mmf_fusion_default_config(&cfg), "mmf_fusion_default_config");
        cfg.time_delta = timeDelta, cfg.conf_global_init = initConfidenceGlobal, cfg.conf_object_init = initConfidenceObject;
        cfg.depth_cutoff = depthCut, cfg.icp_weight = icpThresh, cfg.fast_odom = fastOdom, cfg.so3 = so3;
        cfg.frame_to_frame_rgb = frameToFrameRGB;
        other_["countThresh"] = (float)countThresh, other_["errThresh"] = errThresh, other_["covThresh"] = covThresh;
        other_["iclnuim"] = iclnuim, other_["reloc"] = reloc, other_["photoThresh"] = photoThresh, other_["fernThresh"] = fernThresh;
        other_["modelSpawnOffset"] = (float)modelSpawnOffset, other_["exportSegmentationResults"] = exportSegmentationResults;
        other_["hasKeypointPredictor"] = !keypoint_predictor_path.empty();
        const Intrinsics& K = Intrinsics::getInstance();
        mmf::check(mmf_fusion_create(defaultContext().get(), width_, height_, K.cx(), K.cy(), K.fx(), K.fy(), &cfg, &f_), "mmf_fusion_create");
        mmf::check(mmf_fusion_get_config(f_, &cfg_), "mmf_fusion_get_config");
    }
    static mmf::Context& defaultContext() {
        static mmf::Context ctx(0);
        return ctx;
    }
    const OdometryConfig& getOdometryConfig() const { return odom_cfg_; }
    const SegmentationConfiguration& getSegmentationConfiguration() const { return segm_cfg_; }

    virtual ~MultiMotionFusion() {
        for (auto& kv : textures_) delete kv.second;
        models_.clear();
        mmf_fusion_destroy(f_);
    }
    MultiMotionFusion(const MultiMotionFusion&) = delete;
    MultiMotionFusion& operator=(const MultiMotionFusion&) = delete;

    void preallocateModels(unsigned count) { mmf::check(mmf_fusion_preallocate_models(f_, count), "mmf_fusion_preallocate_models"); }

    // bool processFrame(const FrameData&, const Eigen::Matrix4f* inPose = 0, float weightMultiplier = 1,
    //                   GroundTruthOdometryInterface* = nullptr, bool bootstrap = false)  (MultiMotionFusion.h:78-80).
    // Like the reference it prints "invalid image data" and returns false for a bad frame and
    // returns false after a regular frame (MultiMotionFusion.cpp:209-212, 853).  The frame is uploaded inside (:221, :261).
    bool processFrame(const FrameData& frame, const float* inPose = nullptr, const float weightMultiplier = 1.f,
                      void* /*GroundTruthOdometryInterface*/ = nullptr, const bool bootstrap = false) {
        const FrameData next = next_;  // announced by announceNextFrame (one call only)
        next_ = FrameData();
        return finish(mmf_fusion_process_frame_host_next(f_, frame.rgb, frame.depth, frame.mask, frame.hasNewLabel, frame.timestamp, inPose,
                                                         weightMultiplier, bootstrap, next.rgb, next.depth),
                      "mmf_fusion_process_frame_host");
    }
    // not in the reference: a front-end whose reader is one frame ahead (GUI/MainController.cpp:547-590) announces the frame
    // the NEXT processFrame call will bring (same buffers, unchanged until then): it is uploaded while this one is tracked
    // and prepared while this one is fused.  Optional; a frame that was announced and never comes costs nothing but the copy.
    void announceNextFrame(const FrameData& next) { next_ = next; }
#ifdef MMF_HAVE_EIGEN
    // the reference's exact signature (MultiMotionFusion.h:78-80) and pose getter (:196)
    bool processFrame(const FrameData& frame, const Eigen::Matrix4f* inPose, const float weightMultiplier = 1.f,
                      void* gt_pose = nullptr, const bool bootstrap = false) {
        if (!inPose) return processFrame(frame, static_cast<const float*>(nullptr), weightMultiplier, gt_pose, bootstrap);
        const Eigen::Matrix<float, 4, 4, Eigen::RowMajor> p = *inPose;
        return processFrame(frame, p.data(), weightMultiplier, gt_pose, bootstrap);
    }
    Eigen::Matrix4f getCurrPose() const {
        Eigen::Matrix<float, 4, 4, Eigen::RowMajor> p;
        mmf::check(mmf_fusion_get_pose(f_, p.data()), "mmf_fusion_get_pose");
        return p;
    }
#endif
    bool processFrame(const FrameDataDevice& frame, const float* inPose = nullptr, const float weightMultiplier = 1.f,
                      const bool bootstrap = false) {
        mmf_segmentation seg;
        std::memset(&seg, 0, sizeof(seg));
        seg.mask = frame.mask, seg.has_new_label = frame.hasNewLabel;
        mmf_frame fr;
        std::memset(&fr, 0, sizeof(fr));
        fr.rgb = frame.rgb, fr.depth = frame.depth, fr.timestamp = frame.timestamp;
        fr.in_pose = inPose, fr.weight_multiplier = weightMultiplier, fr.bootstrap = bootstrap, fr.icp_refine = 1;
        fr.segmentation = frame.mask ? &seg : nullptr;
        fr.next_rgb = frame.nextRgb, fr.next_depth = frame.nextDepth;
        return finish(mmf_fusion_process_frame_ex(f_, &fr), "mmf_fusion_process_frame_ex");
    }
    // the same frame step with odom_cfg.init == "kp" (MultiMotionFusion.cpp:312-384): trackTransform is
    // RigidRANSAC::Result::transformation of Model::getLastTrackTransform (row-major 4x4), icpRefine is
    // odom_cfg.icp_refine.  In the reference this is selected by the OdometryConfig passed to the constructor.
    bool processFrame(const FrameDataDevice& frame, const float trackTransform[16], const bool icpRefine,
                      const float weightMultiplier = 1.f) {
        return finish(mmf_fusion_process_frame_init(f_, frame.rgb, frame.depth, frame.timestamp, trackTransform, icpRefine, weightMultiplier),
                      "mmf_fusion_process_frame_init");
    }
    // not in the reference: start the next frame's input-side work (filter, pyramids, gradients) on a second
    // stream while the current frame is still being fused; pass the same frame to processFrame afterwards
    void prefetchFrame(const FrameDataDevice& next) {
        mmf::check(mmf_fusion_prefetch_frame(f_, next.rgb, next.depth), "mmf_fusion_prefetch_frame");
    }

    void predict() { mmf::check(mmf_fusion_predict(f_), "mmf_fusion_predict"); }  // MultiMotionFusion.h:86

    // getIndexMap() (:92): the global model's projections (ModelProjection's getters live on Model here)
    Model& getIndexMap() { return *getBackgroundModel(); }
    ModelPointer getBackgroundModel() { return getModels().front(); }
    ModelList& getModels() {  // :107 -- rebuilt from the native list when it changed
        const int n = mmf_fusion_num_models(f_);
        bool same = (int)models_.size() == n;
        int i = 0;
        for (auto it = models_.begin(); same && it != models_.end(); ++it, ++i) same = (*it)->handle() == mmf_fusion_model_at(f_, i);
        if (!same) {
            models_.clear();
            for (int k = 0; k < n; ++k)
                models_.push_back(std::make_shared<Model>(mmf_fusion_model_at(f_, k), mmf_fusion_odometry_at(f_, k), width_, height_,
                                                          k == 0 && cfg_.fill_in));
        }
        return models_;
    }
    // getTextures() (:124): the raw input images of the current frame
    std::map<std::string, GPUTexture*>& getTextures() {
        struct Spec { const std::string* name; GPUTexture::Format fmt; };
        const Spec specs[] = {{&GPUTexture::RGB, GPUTexture::RGB8}, {&GPUTexture::DEPTH_METRIC, GPUTexture::R32F},
                              {&GPUTexture::DEPTH_METRIC_FILTERED, GPUTexture::R32F}, {&GPUTexture::MASK, GPUTexture::R8UI}};
        for (const Spec& s : specs) {
            const void* p = nullptr;
            size_t b = 0;
            mmf::check(mmf_fusion_texture(f_, s.name->c_str(), &p, &b), "mmf_fusion_texture");
            auto it = textures_.find(*s.name);
            if (it == textures_.end())
                textures_[*s.name] = new GPUTexture(p, width_, height_, s.fmt, *s.name);
            else
                it->second->rebind(p);
        }
        return textures_;
    }
    // getModelToModel() (:136): the front-end only reads lastICPError / lastICPCount of it for its plots
    // (GUI/MainController.cpp:627-640); here they are the global model's frame-to-model statistics
    RGBDOdometry::Stats getModelToModel() {
        RGBDOdometry::Stats s;
        mmf::check(mmf_odom_get_stats(mmf_fusion_odometry(f_), &s), "mmf_odom_get_stats");
        return s;
    }
    float getConfidenceThreshold() { return mmf_model_confidence_threshold(mmf_fusion_model(f_)); }

    // ----- the setters of GUI/MainController.cpp:641-670
    void setRgbOnly(const bool& v) { mmf::check(mmf_fusion_set_rgb_only(f_, v), "setRgbOnly"); }
    void setIcpWeight(const float& v) { mmf::check(mmf_fusion_set_icp_weight(f_, v), "setIcpWeight"); }
    void setOutlierCoefficient(const float& v) { mmf::check(mmf_fusion_set_outlier_coefficient(f_, v), "setOutlierCoefficient"); }
    void setPyramid(const bool& v) { mmf::check(mmf_fusion_set_pyramid(f_, v), "setPyramid"); }
    void setFastOdom(const bool& v) { mmf::check(mmf_fusion_set_fast_odom(f_, v), "setFastOdom"); }
    void setSo3(const bool& v) { mmf::check(mmf_fusion_set_so3(f_, v), "setSo3"); }
    void setFrameToFrameRGB(const bool& v) { mmf::check(mmf_fusion_set_frame_to_frame_rgb(f_, v), "setFrameToFrameRGB"); }
    void setConfidenceThreshold(const float& v) { mmf::check(mmf_fusion_set_confidence_threshold(f_, v), "setConfidenceThreshold"); }
    void setDepthCutoff(const float& v) { mmf::check(mmf_fusion_set_depth_cutoff(f_, v), "setDepthCutoff"); }
    void setEnableMultipleModels(bool v) { mmf::check(mmf_fusion_set_enable_multiple_models(f_, v), "setEnableMultipleModels"); }
    void setTick(const int& v) { mmf::check(mmf_fusion_set_tick(f_, v), "setTick"); }
    void scheduleDeactivation(const ModelPointer& m) { mmf::check(mmf_fusion_schedule_deactivation(f_, (int)m->getID()), "scheduleDeactivation"); }
    // settings of subsystems that stay in the reference's front-end (segmentation / CRF, spawning policy, redetection,
    // ferns): recorded, not interpreted here
    void setFernThresh(const float& v) { other_["fernThresh"] = v; }
    void setModelSpawnOffset(const unsigned& v) { other_["modelSpawnOffset"] = (float)v; }
    void setModelDeactivateCount(const unsigned& v) { other_["modelDeactivateCount"] = (float)v; }
    void setCrfPairwiseSigmaRGB(const float& v) { other_["crfPairwiseSigmaRGB"] = v; }
    void setCrfPairwiseSigmaPosition(const float& v) { other_["crfPairwiseSigmaPosition"] = v; }
    void setCrfPairwiseSigmaDepth(const float& v) { other_["crfPairwiseSigmaDepth"] = v; }
    void setCrfPairwiseWeightAppearance(const float& v) { other_["crfPairwiseWeightAppearance"] = v; }
    void setCrfPairwiseWeightSmoothness(const float& v) { other_["crfPairwiseWeightSmoothness"] = v; }
    void setCrfThresholdNew(const float& v) { other_["crfThresholdNew"] = v; }
    void setCrfUnaryWeightError(const float& v) { other_["crfUnaryWeightError"] = v; }
    void setCrfIteration(const unsigned& v) { other_["crfIteration"] = (float)v; }
    void setCrfUnaryKError(const float& v) { other_["crfUnaryKError"] = v; }
    void setNewModelMinRelativeSize(const float& v) { other_["newModelMinRelativeSize"] = v; }
    void setNewModelMaxRelativeSize(const float& v) { other_["newModelMaxRelativeSize"] = v; }
    void setEnableRedetection(bool v) { other_["enableRedetection"] = v; }
    void setSetInhibit(bool v) { other_["inhibitModels"] = v; }
    void setEnableSmartModelDelete(bool v) { other_["enableSmartModelDelete"] = v; }
    const std::map<std::string, float>& frontEndSettings() const { return other_; }

    // ----- getters (:186-216)
    const bool& getLost() { return lost_; }  // relocalisation stays in the reference: never lost here
    int getTick() const { return mmf_fusion_tick(f_); }
    int getTimeDelta() { return refresh().time_delta; }
    float getMaxDepthProcessed() { return refresh().max_depth_processed; }
    void getCurrPose(float pose[16]) const { mmf::check(mmf_fusion_get_pose(f_, pose), "mmf_fusion_get_pose"); }
    void exportPoses(const std::string& exportDir = "") {  // (the reference writes into the constructor's exportDirectory)
        mmf::check(mmf_fusion_export_poses(f_, (exportDir.empty() ? exportDirectory_ : exportDir).c_str()), "mmf_fusion_export_poses");
    }
    mmf_odom* getFrameOdometryHandle() { return mmf_fusion_odometry(f_); }
    mmf_fusion* handle() const { return f_; }

   private:
    bool finish(int rc, const char* what) {
        if (rc == MMF_ERR_INVALID) {
            std::fprintf(stderr, "%s\n", mmf_last_error());
            return false;
        }
        mmf::check(rc, what);
        return false;
    }
    const mmf_fusion_config& refresh() {
        mmf::check(mmf_fusion_get_config(f_, &cfg_), "mmf_fusion_get_config");
        return cfg_;
    }
    mmf_fusion* f_ = nullptr;
    mmf_fusion_config cfg_;
    int width_, height_;
    OdometryConfig odom_cfg_;
    SegmentationConfiguration segm_cfg_;
    std::string exportDirectory_;
    ModelList models_;
    std::map<std::string, GPUTexture*> textures_;
    std::map<std::string, float> other_;
    bool lost_ = false;
    FrameData next_;
};
