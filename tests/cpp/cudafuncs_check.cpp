// GPU run of the function-level shim (multimotionfusion_amd/cpp/cudafuncs.h): the reference's device entry points
// by their own names on DeviceArray2D handles, with known answers on a synthetic plane.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../multimotionfusion_amd/cpp/cudafuncs.h"

static int fails = 0;
#define EXPECT(cond)                                               \
    do {                                                           \
        if (!(cond)) {                                             \
            std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); \
            ++fails;                                               \
        }                                                          \
    } while (0)

int main() {
    const int W = 160, H = 120;
    const CameraModel intr(132.f, 132.f, 80.f, 60.f);
    std::vector<float> depth(W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) depth[y * W + x] = 1.5f + 0.002f * x + 0.001f * y;  // a tilted plane
    depth[10 * W + 10] = 0.f;                                                          // one invalid pixel
    DeviceArray2D<float> d_depth;
    d_depth.upload(depth.data(), W * sizeof(float), H, W);
    DeviceArray2D<unsigned char> no_mask;

    DeviceArray2D<float> vmap, nmap;
    createVMap(intr, d_depth, no_mask, vmap, 10.f);
    createNMap(vmap, nmap);
    EXPECT(vmap.rows() == 3 * H && vmap.cols() == W && nmap.rows() == 3 * H);
    std::vector<float> v(3 * W * H), n(3 * W * H);
    vmap.download(v.data(), W * sizeof(float));
    nmap.download(n.data(), W * sizeof(float));
    EXPECT(v[(2 * H + 30) * W + 40] == depth[30 * W + 40]);                             // z plane = depth
    EXPECT(std::fabs(v[30 * W + 40] - depth[30 * W + 40] * (40 - 80.f) / 132.f) < 1e-6f);  // x plane
    EXPECT(v[10 * W + 10] != v[10 * W + 10]);                                            // invalid = NaN in the x plane
    const float nx = n[30 * W + 40], ny = n[(H + 30) * W + 40], nz = n[(2 * H + 30) * W + 40];
    EXPECT(std::fabs(nx * nx + ny * ny + nz * nz - 1.f) < 1e-5f);

    // a second handle shares the allocation; create() with the same size keeps it
    DeviceArray2D<float> alias = vmap;
    const float* before = vmap.ptr();
    vmap.create(3 * H, W);
    EXPECT(alias.ptr() == before && vmap.ptr() == before);

    // identity transform returns the same bits
    const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const mat33 R(I9);
    const float3 t0 = make_float3(0.f, 0.f, 0.f);
    DeviceArray2D<float> vg, ng;
    tranformMaps(vmap, nmap, R, t0, vg, ng);
    std::vector<float> v2(3 * W * H);
    vg.download(v2.data(), W * sizeof(float));
    EXPECT(std::memcmp(v.data(), v2.data(), v.size() * sizeof(float)) == 0);

    // ICP of the frame against itself: every pixel with a normal is an inlier, the residual vanishes
    DeviceArray<JtJJtrSE3> sum, out;
    float A[36], b[6], residual[2];
    icpStep(R, t0, vmap, nmap, R, t0, intr, vg, ng, 0.10f, std::sin(20.f * 3.14159254f / 180.f), sum, out, A, b, residual, 0, 0);
    int with_normal = 0;
    for (int i = 0; i < W * H; ++i) with_normal += !(n[i] != n[i]);
    EXPECT((int)residual[1] == with_normal && residual[1] > 0.9f * W * H);
    EXPECT(residual[0] < 1e-6f);
    bool sym = true;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) sym = sym && A[r * 6 + c] == A[c * 6 + r];
    EXPECT(sym && A[0] > 0.f);

    // pyramids and gradients keep their shapes and known values
    DeviceArray2D<float> d1;
    pyrDownGaussF(d_depth, d1);
    EXPECT(d1.rows() == H / 2 && d1.cols() == W / 2);
    DeviceArray2D<float> v1;
    resizeVMap(vmap, v1);
    EXPECT(v1.rows() == 3 * (H / 2) && v1.cols() == W / 2);
    std::vector<unsigned char> img(W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) img[y * W + x] = (unsigned char)(x + 20);  // a horizontal ramp
    DeviceArray2D<unsigned char> d_img, d_img1;
    d_img.upload(img.data(), W, H, W);
    DeviceArray2D<short> dx, dy;
    computeDerivativeImages(d_img, dx, dy);
    std::vector<short> hx(W * H), hy(W * H);
    dx.download(hx.data(), W * sizeof(short));
    dy.download(hy.data(), W * sizeof(short));
    EXPECT(hx[50 * W + 50] == 3 && hy[50 * W + 50] == 0);  // unit ramp: 2 * (0.52201 + 0.79451 + 0.52201) = 3.68, truncated
    pyrDownUcharGauss(d_img, d_img1);
    EXPECT(d_img1.rows() == H / 2 && d_img1.cols() == W / 2);

    CameraModel cm = intr;
    DeviceArray2D<float3> cloud(H, W);
    projectToPointCloud(d_depth, cloud, cm, 0);
    std::vector<float3> hc(W * H);
    cloud.download(hc.data(), W * sizeof(float3));
    EXPECT(hc[30 * W + 40].z == depth[30 * W + 40]);

    std::printf("cudafuncs shim: %s\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
