// The call sequence of the reference's front-end loop (GUI/MainController.cpp:547-715) against the C++ shims:
// per frame  mmf->processFrame(logReader->getFrameData(), currentPose, weightMultiplier, gt_init)        (:588)
//            mmf->getModelToModel().lastICPCount / lastICPError                                            (:627-640)
//            the block of setters pushed every GUI tick                                                    (:641-670)
//            mmf->getTextures()[GPUTexture::MASK], mmf->getModels(), mmf->getIndexMap()                  (:700-706, GUI draw)
//            mmf->setTick(mmf->getTick() + 1) on skip (:621), mmf->exportPoses() at the end (:713)
// on frames rendered here (a textured wall with a box in front of it, the camera sliding sideways), with a
// ground-truth id image so that one object model is spawned.  Also drives a stand-alone Model through
// Model::performTracking / fuse / clean with the reference's argument lists (Model.h:157, 200-207).
// Build: see tests/test_gpu_boundary.py.  Exit code 0 = every check passed.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../multimotionfusion_amd/cpp/MultiMotionFusion.h"

static const int W = 320, H = 240;
static const float FX = 264.f, FY = 264.f, CX = 160.f, CY = 120.f;

struct Frame {
    std::vector<uint8_t> rgb, mask;
    std::vector<float> depth;
};

// wall at z = 2.5 m, a 0.5 m box face at z = 1.6 m; camera at (shift, 0, 0) looking down +z
static Frame render(float shift, float box_shift) {
    Frame f;
    f.rgb.resize((size_t)W * H * 3), f.mask.resize((size_t)W * H), f.depth.resize((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float dx = (x - CX) / FX, dy = (y - CY) / FY;
            float z = 2.5f + 0.15f * std::sin(3.f * (shift + dx * 2.5f)) * std::cos(2.f * dy * 2.5f);
            float px = shift + dx * z, py = dy * z;
            uint8_t id = 0;
            const float bx = shift + dx * 1.6f - box_shift, by = dy * 1.6f;
            if (std::fabs(bx) < 0.25f && std::fabs(by) < 0.2f) {
                z = 1.6f + 0.1f * bx + 0.05f * by, px = bx, py = by, id = 1;
            }
            const float v = 0.5f + 0.2f * std::sin(9.f * px + (id ? 1.f : 0.f)) * std::sin(7.f * py) + 0.2f * std::sin(4.f * px + 3.f * py);
            const size_t i = (size_t)y * W + x;
            f.depth[i] = z, f.mask[i] = id;
            f.rgb[3 * i] = (uint8_t)(40 + 180 * v), f.rgb[3 * i + 1] = (uint8_t)(30 + 170 * v), f.rgb[3 * i + 2] = (uint8_t)(50 + 150 * (1 - v));
        }
    return f;
}

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) {                                                    \
            std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main(int argc, char** argv) {
    const std::string exportDir = argc > 1 ? argv[1] : "/tmp/";
    mmf::Context ctx(0);
    mmf_fusion_config cfg;
    mmf_fusion_default_config(&cfg);
    cfg.pose_logging = 1;
    MultiMotionFusion* mmf = new MultiMotionFusion(ctx, W, H, CX, CY, FX, FY, &cfg);
    mmf->preallocateModels(1);  // MainController.cpp:523

    const int n_frames = 8;
    for (int i = 0; i < n_frames; ++i) {
        const Frame fr = render(0.004f * i, 0.006f * i);
        FrameData frame;  // logReader->getFrameData()
        frame.timestamp = 1000 + 33 * i, frame.rgb = fr.rgb.data(), frame.depth = fr.depth.data();
        if (i >= 2) frame.mask = fr.mask.data(), frame.hasNewLabel = (i == 2);
        if (i == 2) mmf->setEnableMultipleModels(true);
        float* currentPose = nullptr;
        const float weightMultiplier = 1.f;
        if (mmf->processFrame(frame, currentPose, weightMultiplier, nullptr)) return 2;  // :588

        const RGBDOdometry::Stats m2m = mmf->getModelToModel();  // :627-640
        if (i > 0) CHECK(m2m.lastICPCount > 0.5f * W * H && !std::isnan(m2m.lastICPError));

        // SET PARAMETERS / SETTINGS (:641-670)
        mmf->setEnableMultipleModels(i >= 2);
        mmf->setEnableRedetection(false);
        mmf->setSetInhibit(false);
        mmf->setEnableSmartModelDelete(false);
        mmf->setRgbOnly(false);
        mmf->setPyramid(true);
        mmf->setFastOdom(i % 2 == 1);  // toggled at run time like a GUI checkbox
        mmf->setDepthCutoff(15.f);
        mmf->setIcpWeight(10.f);
        mmf->setOutlierCoefficient(3.f);
        mmf->setSo3(true);
        mmf->setFrameToFrameRGB(false);
        mmf->setModelSpawnOffset(20);
        mmf->setModelDeactivateCount(10);
        mmf->setNewModelMinRelativeSize(0.01f);
        mmf->setNewModelMaxRelativeSize(0.5f);
        mmf->setCrfPairwiseWeightAppearance(1.f);
        mmf->setCrfPairwiseWeightSmoothness(1.f);
        mmf->setCrfPairwiseSigmaDepth(1.f);
        mmf->setCrfPairwiseSigmaPosition(1.f);
        mmf->setCrfPairwiseSigmaRGB(1.f);
        mmf->setCrfThresholdNew(1.f);
        mmf->setCrfUnaryKError(1.f);
        mmf->setCrfUnaryWeightError(1.f);
        mmf->setCrfIteration(10);

        // what the GUI draws from (:700-706 and drawScene)
        std::map<std::string, GPUTexture*>& tex = mmf->getTextures();
        CHECK(tex.count(GPUTexture::RGB) && tex.count(GPUTexture::DEPTH_METRIC) && tex.count(GPUTexture::DEPTH_METRIC_FILTERED) && tex.count(GPUTexture::MASK));
        const std::vector<uint8_t> mask_now = tex[GPUTexture::MASK]->downloadTexture();
        CHECK(mask_now.size() == (size_t)W * H);
        size_t labelled = 0;
        for (uint8_t v : mask_now) labelled += v == 1;
        CHECK((i >= 2) == (labelled > 1000));
        ModelList& models = mmf->getModels();
        CHECK((int)models.size() == (i >= 2 ? 2 : 1) && models.front()->getID() == 0);
        Model& indexMap = mmf->getIndexMap();
        CHECK(indexMap.getSplatVertexConfTexBytes() == (size_t)W * H * 16);
        CHECK(mmf->getTick() == i + 2);
        std::printf("frame %d tick %d models %zu surfels %u icpCount %.0f icpError %.2e\n", i, mmf->getTick(), models.size(),
                    mmf->getBackgroundModel()->lastCount(), m2m.lastICPCount, m2m.lastICPError);
    }
    float pose[16];
    mmf->getCurrPose(pose);
    CHECK(std::fabs(pose[3] - 0.004f * (n_frames - 1)) < 0.01f);  // the camera slid sideways
    ModelList& models = mmf->getModels();
    CHECK(models.size() == 2 && models.back()->getID() == 1 && models.back()->lastCount() > 300);
    CHECK(!models.back()->allowsFillIn() && std::fabs(models.back()->getConfidenceThreshold() - 0.01f) < 1e-6f);
    mmf->predict();                       // MultiMotionFusion.h:86
    mmf->setTick(mmf->getTick() + 1);     // skip (:621)
    CHECK(mmf->getTick() == n_frames + 2);
    mmf->exportPoses(exportDir);          // :713
    for (int id = 0; id < 2; ++id) {
        std::ifstream in(exportDir + "poses-" + std::to_string(id) + ".txt");
        std::string line;
        int lines = 0;
        while (std::getline(in, line)) ++lines;
        CHECK(lines == (id == 0 ? n_frames : n_frames - 2));
    }
    // bad frame: "invalid image data", returns false, nothing changes (:209-212)
    FrameData bad;
    CHECK(mmf->processFrame(bad) == false && mmf->getTick() == n_frames + 2);
    delete mmf;

    // ---- a stand-alone Model driven with the reference's own argument lists
    {
        const Frame f0 = render(0.f, 0.f), f1 = render(0.005f, 0.f);
        uint8_t *d_rgb[2], *d_mask;
        float *d_depth[2], *d_filtered;
        for (int k = 0; k < 2; ++k) {
            const Frame& f = k ? f1 : f0;
            CHECK(hipMalloc((void**)&d_rgb[k], f.rgb.size()) == hipSuccess && hipMalloc((void**)&d_depth[k], f.depth.size() * 4) == hipSuccess);
            CHECK(hipMemcpy(d_rgb[k], f.rgb.data(), f.rgb.size(), hipMemcpyHostToDevice) == hipSuccess);
            CHECK(hipMemcpy(d_depth[k], f.depth.data(), f.depth.size() * 4, hipMemcpyHostToDevice) == hipSuccess);
        }
        CHECK(hipMalloc((void**)&d_mask, (size_t)W * H) == hipSuccess && hipMemset(d_mask, 0, (size_t)W * H) == hipSuccess);
        CHECK(hipMalloc((void**)&d_filtered, (size_t)W * H * 4) == hipSuccess);
        Model model(ctx, W, H, CX, CY, FX, FY, 0, 10.f, true);
        GPUTexture mask(d_mask, W, H, GPUTexture::R8UI, GPUTexture::MASK);
        GPUTexture filtered(d_filtered, W, H, GPUTexture::R32F, GPUTexture::DEPTH_METRIC_FILTERED);
        std::vector<float> rawGraph;
        for (int k = 0; k < 2; ++k) {
            const int tick = k + 1;
            GPUTexture rgb(d_rgb[k], W, H, GPUTexture::RGB8, GPUTexture::RGB), depth(d_depth[k], W, H, GPUTexture::R32F, GPUTexture::DEPTH_METRIC);
            mmf::check(mmf_filter_depth(ctx.get(), d_depth[k], W, H, 15.f, d_filtered), "mmf_filter_depth");
            if (k == 0) {
                model.initialise(&rgb, &depth, &filtered, tick, 20.f);
                mmf::check(mmf_odom_init_first_rgb(model.odometryHandle(), d_rgb[k], 0, 3), "initFirstRGB");
            } else {
                Model::generateCUDATextures(&filtered, &mask);
                model.performTracking(false, false, 10.f, true, false, true, 20.f, &rgb, 33, model.requiresFillIn());
                model.combinedPredict(20.f, tick, tick, 200);
                model.performFillIn(&rgb, &filtered, false, false);
                model.predictIndices(tick, 20.f, 200);
                model.fuse(tick, &rgb, &mask, &depth, &filtered, 20.f, 1.f);
                model.predictIndices(tick, 20.f, 200);
                model.clean(tick, rawGraph, 200, 20.f, false, &filtered, &mask);
            }
            model.combinedPredict(20.f, tick, tick, 200);
            model.performFillIn(&rgb, &filtered, false, false);
        }
        float p[16];
        model.getPose(p);
        const RGBDOdometry::Stats st = model.getFrameOdometryStats();
        std::printf("stand-alone Model: surfels %u pose.x %.4f icpCount %.0f weight %.3f\n", model.lastCount(), p[3], st.lastICPCount,
                    model.computeFusionWeight(1.f));
        CHECK(std::fabs(p[3] - 0.005f) < 0.002f && st.iterations_run == 19 && model.lastCount() > 0.8 * W * H);
        CHECK(model.computeFusionWeight(1.f) >= 0.5f && model.computeFusionWeight(1.f) < 1.f);  // the camera moved 5 mm
        // what Segmentation.cpp:218-219 reads of a model, and Model::getModel() (Model.h:278-297)
        const std::vector<float> icpErr = model.downloadICPErrorTexture(), vertConf = model.downloadVertexConfTexture();
        CHECK(icpErr.size() == (size_t)W * H && vertConf.size() == (size_t)W * H * 4 && model.getRGBErrorTexture()->width == W);
        size_t withError = 0, badConf = 0;
        for (float e : icpErr) withError += e > 0.f;
        for (size_t i = 0; i < (size_t)W * H; ++i) badConf += !(vertConf[4 * i + 3] >= 0.f);
        // (a two-frame map is all unstable surfels: the splat shows none of them yet, the tracker ran on the fill-in images)
        CHECK(withError > 1000 && badConf == 0);
        const OutputBuffer& vbo = model.getModel();
        const std::vector<Model::surfel_t> host = model.downloadMap();
        CHECK(vbo.count == model.lastCount() && vbo.count == host.size() && vbo.positionConfidence && vbo.colourTime && vbo.normalRadius);
        float first[4], lastn[4];
        CHECK(hipMemcpy(first, vbo.positionConfidence, sizeof(first), hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(hipMemcpy(lastn, vbo.normalRadius + 4 * (size_t)(vbo.count - 1), sizeof(lastn), hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(first[0] == host.front().position[0] && first[3] == host.front().confidence && lastn[3] == host.back().radius);
        ctx.synchronize();
        for (int k = 0; k < 2; ++k) (void)hipFree(d_rgb[k]), (void)hipFree(d_depth[k]);
        (void)hipFree(d_mask), (void)hipFree(d_filtered);
    }
    // ---- the reference's own constructor (Core/MultiMotionFusion.h:54-61) over the Resolution / Intrinsics singletons,
    //      as GUI/MainController.cpp:147-148, 514-517 use them; the reader announces the next frame with each call
    {
        Resolution::setResolution(W, H);
        Intrinsics::setIntrinics(FX, FY, CX, CY);
        CHECK(Resolution::getInstance().numPixels() == W * H && Intrinsics::getInstance().cx() == CX);
        OdometryConfig odom_cfg;
        SegmentationConfiguration segm_cfg;
        MultiMotionFusion ref(200, 35000, 5e-05f, 1e-05f, /*closeLoops*/ false, false, false, 115, /*confGlobal*/ 10.f, /*confObject*/ 0.01f,
                              /*depthCut*/ 15.f, /*icpWeight*/ 10.f, false, 0.3095f, true, false, 20, Model::MatchingType::Drost, exportDir, false,
                              std::string(), odom_cfg, segm_cfg);
        std::vector<Frame> frames;
        for (int i = 0; i < 4; ++i) frames.push_back(render(0.004f * i, 0.f));
        for (int i = 0; i < 4; ++i) {
            FrameData frame, next;
            frame.timestamp = 33 * i, frame.rgb = frames[i].rgb.data(), frame.depth = frames[i].depth.data();
            if (i + 1 < 4) {
                next.timestamp = 33 * (i + 1), next.rgb = frames[i + 1].rgb.data(), next.depth = frames[i + 1].depth.data();
                ref.announceNextFrame(next);
            }
            CHECK(ref.processFrame(frame) == false);
        }
        float p[16];
        ref.getCurrPose(p);
        CHECK(std::fabs(p[3] - 0.012f) < 0.004f && ref.getTick() == 5 && ref.getTimeDelta() == 200);
        ModelPointer bg = ref.getBackgroundModel();
        CHECK(bg->getModel().count == bg->lastCount() && bg->lastCount() > (unsigned)(0.8 * W * H));
        CHECK(bg->getICPErrorTexture()->bytes() == (size_t)W * H * 4);
        size_t withError = 0;
        for (float e : bg->downloadICPErrorTexture()) withError += e > 0.f;
        CHECK(withError > 1000);
    }
    std::printf("main controller sequence: ok\n");
    return 0;
}
