// Compile/link check of the C++ shims (multimotionfusion_amd/cpp/*.h) against libmmf_hip.so.
// Run with an argument on a GPU box to push two synthetic-free frames through processFrame.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../multimotionfusion_amd/cpp/MultiMotionFusion.h"
#include "../../multimotionfusion_amd/cpp/RigidRANSAC.h"

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
    Model bg = mmf.getBackgroundModel();
    std::printf("processFrame(bad)=%d tick=%d surfels=%u pose00=%g icpCount=%g\n", (int)r, mmf.getTick(), bg.lastCount(),
                pose[0], odom.lastICPCount);
    return (r == false && mmf.getTick() == 1 && bg.lastCount() == 0) ? 0 : 2;
}
