// Compile/link check of the C++ shims (multimotionfusion_amd/cpp/*.h) against libmmf_hip.so.
// Run with an argument on a GPU box to push two synthetic-free frames through processFrame.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <tuple>
#include <vector>

#include "../../multimotionfusion_amd/cpp/MultiMotionFusion.h"
#include "../../multimotionfusion_amd/cpp/RigidRANSAC.h"
#include "../../multimotionfusion_amd/cpp/SuperPoint.h"
#include "../../multimotionfusion_amd/cpp/cudafuncs.h"

int main(int argc, char** argv) {
    if (argc < 2) {  // CPU containers: the check is that everything above compiles and links
        // RigidRANSAC is host code: a translation by (1, 2, 3) must come back
        const float p1[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
        float p0[12];
        for (int i = 0; i < 12; ++i) p0[i] = p1[i] + (float)(i % 3 + 1);
        RigidRANSAC ransac(5, 0.1f, 0.5f);
        const RigidRANSAC::Result res = ransac.estimate(p0, p1, 4);
        const bool ok = std::fabs(res.transformation[3] - 1.f) < 1e-5f && std::fabs(res.transformation[11] - 3.f) < 1e-5f;
        std::printf("abi %d\n", mmf_abi_version());
        return (mmf_abi_version() == MMF_ABI_VERSION && ok) ? 0 : 1;
    }
    mmf::Context ctx(0);
    MultiMotionFusion mmf(ctx, 640, 480, 320.f, 240.f, 528.f, 528.f);
    RGBDOdometry odom(ctx, 640, 480, 320.f, 240.f, 528.f, 528.f);
    (void)argv;
    FrameDataDevice bad;  // null images: processFrame must print "invalid image data" and return false
    const bool r = mmf.processFrame(bad);
    float pose[16];
    mmf.getCurrPose(pose);
    ModelPointer bgp = mmf.getBackgroundModel();
    Model& bg = *bgp;
    std::printf("processFrame(bad)=%d tick=%d surfels=%u pose00=%g icpCount=%g\n", (int)r, mmf.getTick(), bg.lastCount(),
                pose[0], odom.lastICPCount);
    // SuperPoint with all-zero weights: every heat value is 1/65 >= 0.015, so the greedy suppression leaves a
    // regular grid of keypoints whose normalised coordinates lie in [0, 1)
    static const size_t wsize[12] = {64 * 9, 64 * 64 * 9, 64 * 64 * 9, 64 * 64 * 9, 128 * 64 * 9, 128 * 128 * 9,
                                     128 * 128 * 9, 128 * 128 * 9, 256 * 128 * 9, 65 * 256, 256 * 128 * 9, 256 * 256};
    std::vector<std::vector<float>> store;
    std::vector<const float*> weights;
    for (int l = 0; l < 12; ++l) {
        store.emplace_back(wsize[l], 0.f), store.emplace_back(256, 0.f);
    }
    for (auto& v : store) weights.push_back(v.data());
    SuperPoint kp(ctx, weights.data(), 64, 48, 256);
    unsigned char* img = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&img), 64 * 48) != hipSuccess || hipMemset(img, 0, 64 * 48) != hipSuccess) return 3;
    std::vector<double> coordinates, descriptors;
    std::tie(coordinates, descriptors) = kp.getFeatures(img, 64, 48, 1);
    (void)hipFree(img);
    bool kp_ok = !coordinates.empty() && descriptors.size() == coordinates.size() / 2 * 256;
    for (double c : coordinates) kp_ok = kp_ok && c >= 0.0 && c < 1.0;
    std::printf("keypoints=%zu\n", coordinates.size() / 2);
    return (r == false && mmf.getTick() == 1 && bg.lastCount() == 0 && kp_ok) ? 0 : 2;
}
