// cudafuncs.h -- function-level swap: the device entry points of the reference with their own names and argument
// order (Core/Cuda/cudafuncs.cuh:64-193), plus minimal stand-ins for the resource handles that appear in those
// signatures (Core/Cuda/containers/device_array.hpp: DeviceArray / DeviceArray2D; Core/Cuda/types.cuh: mat33,
// CameraModel, DataTerm, JtJJtrSE3 / JtJJtrSO3), over the C ABI of include/mmf_hip.h.  With this header the
// reference's own RGBDOdometry.cpp can keep calling icpStep(...), pyrDownGaussF(...) etc..
//
// Differences a maintainer has to bridge:
//   * cudaSurfaceObject_t (the ICP / RGB error surfaces) becomes a DeviceArray2D<float>* (or nullptr);
//     cudaArray* (imageBGRToIntensity) becomes a DeviceArray2D<unsigned char> holding interleaved pixels + the
//     channel count; copyMaps / verticesToDepth take the RGBA32F prediction as DeviceArray<float> like the reference;
//   * `sum`, `out`, `threads`, `blocks` are accepted and ignored: the kernels own their scratch and launch shapes;
//   * the calls run on one process-wide context (device 0, the null stream) and are synchronous where the
//     reference is (the *Step functions and computeRgbResidual return host results);
//   * errors print and exit(-1) like the reference's cudaSafeCall (Core/Cuda/convenience.cuh:74-83).
// Needs the HIP runtime API for allocation: compile with -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include, link amdhip64.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <memory>
#include <vector>

#include "RGBDOdometry.h"

// ---- types.cuh -------------------------------------------------------------------------------------------------
struct mat33 {
    mat33() {}
    explicit mat33(const float* row_major9) { std::memcpy(data, row_major9, sizeof(data)); }
#ifdef MMF_HAVE_EIGEN
    mat33(Eigen::Matrix<float, 3, 3, Eigen::RowMajor>& e) { std::memcpy(data, e.data(), sizeof(data)); }
#endif
    float3 data[3];
};

struct DataTerm {  // 16 bytes, as mmf_dataterm
    short2 zero;
    short2 one;
    float diff;
    bool valid;
};
static_assert(sizeof(DataTerm) == sizeof(mmf_dataterm), "DataTerm layout");

struct CameraModel {
    float fx, fy, cx, cy;
    CameraModel() : fx(0), fy(0), cx(0), cy(0) {}
    CameraModel(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
    CameraModel operator()(int level) const {
        const int div = 1 << level;
        return CameraModel(fx / div, fy / div, cx / div, cy / div);
    }
};

struct JtJJtrSE3 {  // scratch element types of the reference's signatures (unused here)
    float v[29];
};
struct JtJJtrSO3 {
    float v[11];
};

// ---- containers: shared ownership, create() reallocates only on a size change ---------------------------------
namespace mmf {
inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        std::fprintf(stderr, "%s failed: %s\n", what, hipGetErrorString(e));
        std::exit(-1);
    }
}
inline std::shared_ptr<void> device_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes) hip_check(hipMalloc(&p, bytes), "hipMalloc");
    return std::shared_ptr<void>(p, [](void* q) {
        if (q) (void)hipFree(q);
    });
}
inline Context& default_context() {  // what the reference's default-stream calls amount to
    static Context ctx(0, nullptr, false);
    return ctx;
}
}  // namespace mmf

template <class T>
class DeviceArray {
   public:
    typedef T type;
    enum { elem_size = sizeof(T) };
    DeviceArray() {}
    explicit DeviceArray(size_t size) { create(size); }
    void create(size_t size) {
        if (size == size_ && mem_) return;
        mem_ = mmf::device_alloc(size * sizeof(T));
        size_ = size;
    }
    void release() {
        mem_.reset();
        size_ = 0;
    }
    void upload(const T* host_ptr, size_t size) {
        create(size);
        mmf::hip_check(hipMemcpy(ptr(), host_ptr, size * sizeof(T), hipMemcpyHostToDevice), "DeviceArray::upload");
    }
    void download(T* host_ptr) const {
        mmf::hip_check(hipMemcpy(host_ptr, ptr(), size_ * sizeof(T), hipMemcpyDeviceToHost), "DeviceArray::download");
    }
    void upload(const std::vector<T>& data) { upload(data.data(), data.size()); }
    void download(std::vector<T>& data) const {
        data.resize(size_);
        if (size_) download(data.data());
    }
    void swap(DeviceArray& o) {
        mem_.swap(o.mem_);
        std::swap(size_, o.size_);
    }
    T* ptr() { return static_cast<T*>(mem_.get()); }
    const T* ptr() const { return static_cast<const T*>(mem_.get()); }
    operator T*() { return ptr(); }
    operator const T*() const { return ptr(); }
    size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }

   private:
    std::shared_ptr<void> mem_;
    size_t size_ = 0;
};

template <class T>
class DeviceArray2D {  // dense rows here (step() == cols() * sizeof(T)); the C ABI takes any pitch
   public:
    typedef T type;
    enum { elem_size = sizeof(T) };
    DeviceArray2D() {}
    DeviceArray2D(int rows, int cols) { create(rows, cols); }
    void create(int rows, int cols) {
        if (rows == rows_ && cols == cols_ && mem_) return;
        mem_ = mmf::device_alloc((size_t)rows * cols * sizeof(T));
        rows_ = rows, cols_ = cols;
    }
    void release() {
        mem_.reset();
        rows_ = cols_ = 0;
    }
    void upload(const void* host_ptr, size_t host_step, int rows, int cols) {
        create(rows, cols);
        mmf::hip_check(hipMemcpy2D(ptr(), step(), host_ptr, host_step, (size_t)cols * sizeof(T), rows, hipMemcpyHostToDevice),
                       "DeviceArray2D::upload");
    }
    void download(void* host_ptr, size_t host_step) const {
        mmf::hip_check(hipMemcpy2D(host_ptr, host_step, ptr(), step(), (size_t)cols_ * sizeof(T), rows_, hipMemcpyDeviceToHost),
                       "DeviceArray2D::download");
    }
    void swap(DeviceArray2D& o) {
        mem_.swap(o.mem_);
        std::swap(rows_, o.rows_), std::swap(cols_, o.cols_);
    }
    T* ptr(int y = 0) { return static_cast<T*>(mem_.get()) + (size_t)y * cols_; }
    const T* ptr(int y = 0) const { return static_cast<const T*>(mem_.get()) + (size_t)y * cols_; }
    operator T*() { return ptr(); }
    operator const T*() const { return ptr(); }
    int cols() const { return cols_; }
    int rows() const { return rows_; }
    size_t step() const { return (size_t)cols_ * sizeof(T); }
    size_t elem_step() const { return (size_t)cols_; }
    bool empty() const { return rows_ == 0 || cols_ == 0; }

   private:
    std::shared_ptr<void> mem_;
    int rows_ = 0, cols_ = 0;
};

// ---- cudafuncs.cuh:64-193 ----------------------------------------------------------------------------------------
namespace mmf {
inline mmf_camera cam(const CameraModel& c) { return mmf_camera{c.fx, c.fy, c.cx, c.cy}; }
inline const float* f9(const mat33& m) { return reinterpret_cast<const float*>(m.data); }
inline const float* f3(const float3& v) { return &v.x; }
}  // namespace mmf

// vertex / normal maps are planar: 3 * rows x cols (reduce.cu:261-263)
inline void icpStep(const mat33& Rcurr, const float3& tcurr, const DeviceArray2D<float>& vmap_curr,
                    const DeviceArray2D<float>& nmap_curr, const mat33& Rprev_inv, const float3& tprev, const CameraModel& intr,
                    const DeviceArray2D<float>& vmap_g_prev, const DeviceArray2D<float>& nmap_g_prev, float distThres,
                    float angleThres, DeviceArray<JtJJtrSE3>& /*sum*/, DeviceArray<JtJJtrSE3>& /*out*/, float* matrixA_host,
                    float* vectorB_host, float* residual_host, int /*threads*/, int /*blocks*/,
                    DeviceArray2D<float>* icpErrorSurface = nullptr) {
    const int cols = vmap_curr.cols(), rows = vmap_curr.rows() / 3;
    const mmf_camera c = mmf::cam(intr);
    mmf::check(mmf_icp_step(mmf::default_context().get(), mmf::f9(Rcurr), mmf::f3(tcurr), vmap_curr.ptr(), vmap_curr.step(),
                            nmap_curr.ptr(), nmap_curr.step(), mmf::f9(Rprev_inv), mmf::f3(tprev), &c, vmap_g_prev.ptr(),
                            vmap_g_prev.step(), nmap_g_prev.ptr(), nmap_g_prev.step(), distThres, angleThres, cols, rows,
                            matrixA_host, vectorB_host, residual_host, icpErrorSurface ? icpErrorSurface->ptr() : nullptr,
                            icpErrorSurface ? icpErrorSurface->step() : 0),
               "icpStep");
}

inline void rgbStep(const DeviceArray2D<DataTerm>& corresImg, const float& sigma, const DeviceArray2D<float3>& cloud,
                    const float& fx, const float& fy, const DeviceArray2D<short>& dIdx, const DeviceArray2D<short>& dIdy,
                    const float& sobelScale, DeviceArray<JtJJtrSE3>& /*sum*/, DeviceArray<JtJJtrSE3>& /*out*/,
                    float* matrixA_host, float* vectorB_host, int /*threads*/, int /*blocks*/) {
    mmf::check(mmf_rgb_step(mmf::default_context().get(), reinterpret_cast<const mmf_dataterm*>(corresImg.ptr()), sigma,
                            reinterpret_cast<const float*>(cloud.ptr()), fx, fy, dIdx.ptr(), dIdx.step(), dIdy.ptr(),
                            dIdy.step(), sobelScale, dIdx.cols(), dIdx.rows(), matrixA_host, vectorB_host),
               "rgbStep");
}

inline void so3Step(const DeviceArray2D<unsigned char>& lastImage, const DeviceArray2D<unsigned char>& nextImage,
                    const mat33& imageBasis, const mat33& kinv, const mat33& krlr, DeviceArray<JtJJtrSO3>& /*sum*/,
                    DeviceArray<JtJJtrSO3>& /*out*/, float* matrixA_host, float* vectorB_host, float* residual_host,
                    int /*threads*/, int /*blocks*/) {
    mmf::check(mmf_so3_step(mmf::default_context().get(), lastImage.ptr(), lastImage.step(), nextImage.ptr(), nextImage.step(),
                            mmf::f9(imageBasis), mmf::f9(kinv), mmf::f9(krlr), lastImage.cols(), lastImage.rows(), matrixA_host,
                            vectorB_host, residual_host),
               "so3Step");
}

inline void computeRgbResidual(const float& minScale, const DeviceArray2D<short>& dIdx, const DeviceArray2D<short>& dIdy,
                               const DeviceArray2D<float>& lastDepth, const DeviceArray2D<float>& nextDepth,
                               const DeviceArray2D<unsigned char>& lastImage, const DeviceArray2D<unsigned char>& nextImage,
                               const DeviceArray2D<unsigned char>& /*lastMask*/, const DeviceArray2D<unsigned char>& /*nextMask*/,
                               DeviceArray2D<DataTerm>& corresImg, DeviceArray<int2>& /*sumResidual*/, const float maxDepthDelta,
                               const float3& kt, const mat33& krkinv, int& sigmaSum, int& count, int /*threads*/, int /*blocks*/,
                               DeviceArray2D<float>* rgbErrorSurface = nullptr, unsigned char /*maskID*/ = 0) {
    mmf::check(mmf_compute_rgb_residual(mmf::default_context().get(), minScale, dIdx.ptr(), dIdx.step(), dIdy.ptr(), dIdy.step(),
                                        lastDepth.ptr(), lastDepth.step(), nextDepth.ptr(), nextDepth.step(), lastImage.ptr(),
                                        lastImage.step(), nextImage.ptr(), nextImage.step(),
                                        reinterpret_cast<mmf_dataterm*>(corresImg.ptr()), maxDepthDelta, mmf::f3(kt),
                                        mmf::f9(krkinv), nextImage.cols(), nextImage.rows(), &sigmaSum, &count,
                                        rgbErrorSurface ? rgbErrorSurface->ptr() : nullptr,
                                        rgbErrorSurface ? rgbErrorSurface->step() : 0),
               "computeRgbResidual");
}

inline void createVMap(const CameraModel& intr, const DeviceArray2D<float>& depth, const DeviceArray2D<unsigned char>& /*mask*/,
                       DeviceArray2D<float>& vmap, const float depthCutoff, unsigned char /*maskID*/ = 0) {
    vmap.create(depth.rows() * 3, depth.cols());
    const mmf_camera c = mmf::cam(intr);
    mmf::check(mmf_create_vmap(mmf::default_context().get(), &c, depth.ptr(), depth.step(), depth.cols(), depth.rows(),
                               vmap.ptr(), vmap.step(), depthCutoff),
               "createVMap");
}

inline void createNMap(const DeviceArray2D<float>& vmap, DeviceArray2D<float>& nmap) {
    nmap.create(vmap.rows(), vmap.cols());
    mmf::check(mmf_create_nmap(mmf::default_context().get(), vmap.ptr(), vmap.step(), vmap.cols(), vmap.rows() / 3, nmap.ptr(),
                               nmap.step()),
               "createNMap");
}

inline void tranformMaps(const DeviceArray2D<float>& vmap_src, const DeviceArray2D<float>& nmap_src, const mat33& Rmat,
                         const float3& tvec, DeviceArray2D<float>& vmap_dst, DeviceArray2D<float>& nmap_dst) {
    vmap_dst.create(vmap_src.rows(), vmap_src.cols());
    nmap_dst.create(vmap_src.rows(), vmap_src.cols());
    mmf::check(mmf_transform_maps(mmf::default_context().get(), vmap_src.ptr(), nmap_src.ptr(), vmap_src.step(), vmap_src.cols(),
                                  vmap_src.rows() / 3, mmf::f9(Rmat), mmf::f3(tvec), vmap_dst.ptr(), nmap_dst.ptr(),
                                  vmap_dst.step()),
               "tranformMaps");
}

// vmap_src / nmap_src: dense RGBA32F images (the model prediction), vmap_dst / nmap_dst sized by the caller (3*rows x cols)
inline void copyMaps(const DeviceArray<float>& vmap_src, const DeviceArray<float>& nmap_src, DeviceArray2D<float>& vmap_dst,
                     DeviceArray2D<float>& nmap_dst) {
    mmf::check(mmf_copy_maps(mmf::default_context().get(), vmap_src.ptr(), nmap_src.ptr(), vmap_dst.cols(), vmap_dst.rows() / 3,
                             vmap_dst.ptr(), nmap_dst.ptr(), vmap_dst.step()),
               "copyMaps");
}

inline void resizeVMap(const DeviceArray2D<float>& input, DeviceArray2D<float>& output) {
    output.create((input.rows() / 3 / 2) * 3, input.cols() / 2);
    mmf::check(mmf_resize_vmap(mmf::default_context().get(), input.ptr(), input.step(), input.cols(), input.rows() / 3,
                               output.ptr(), output.step()),
               "resizeVMap");
}

inline void resizeNMap(const DeviceArray2D<float>& input, DeviceArray2D<float>& output) {
    output.create((input.rows() / 3 / 2) * 3, input.cols() / 2);
    mmf::check(mmf_resize_nmap(mmf::default_context().get(), input.ptr(), input.step(), input.cols(), input.rows() / 3,
                               output.ptr(), output.step()),
               "resizeNMap");
}

// cuArr of the reference: here `image` holds rows x (cols * channels) interleaved bytes; dst is sized by the caller
inline void imageBGRToIntensity(const DeviceArray2D<unsigned char>& image, int channels, DeviceArray2D<unsigned char>& dst) {
    mmf::check(mmf_image_bgr_to_intensity(mmf::default_context().get(), image.ptr(), image.step(), channels, dst.cols(), dst.rows(),
                                          dst.ptr(), dst.step()),
               "imageBGRToIntensity");
}

inline void verticesToDepth(DeviceArray<float>& vmap_src, DeviceArray2D<float>& dst, float cutOff) {
    mmf::check(mmf_vertices_to_depth(mmf::default_context().get(), vmap_src.ptr(), dst.cols(), dst.rows(), cutOff, dst.ptr(),
                                     dst.step()),
               "verticesToDepth");
}

inline void projectToPointCloud(const DeviceArray2D<float>& depth, const DeviceArray2D<float3>& cloud, CameraModel& intrinsics,
                                const int& level) {
    const mmf_camera c = mmf::cam(intrinsics);
    mmf::check(mmf_project_to_point_cloud(mmf::default_context().get(), depth.ptr(), depth.step(), depth.cols(), depth.rows(), &c,
                                          level, const_cast<float*>(reinterpret_cast<const float*>(cloud.ptr()))),
               "projectToPointCloud");
}

inline void pyrDownGaussF(const DeviceArray2D<float>& src, DeviceArray2D<float>& dst) {
    dst.create(src.rows() / 2, src.cols() / 2);
    mmf::check(mmf_pyr_down_gauss_f(mmf::default_context().get(), src.ptr(), src.step(), src.cols(), src.rows(), dst.ptr(),
                                    dst.step()),
               "pyrDownGaussF");
}

inline void pyrDownUcharGauss(const DeviceArray2D<unsigned char>& src, DeviceArray2D<unsigned char>& dst) {
    dst.create(src.rows() / 2, src.cols() / 2);
    mmf::check(mmf_pyr_down_uchar_gauss(mmf::default_context().get(), src.ptr(), src.step(), src.cols(), src.rows(), dst.ptr(),
                                        dst.step()),
               "pyrDownUcharGauss");
}

inline void computeDerivativeImages(DeviceArray2D<unsigned char>& src, DeviceArray2D<short>& dx, DeviceArray2D<short>& dy) {
    dx.create(src.rows(), src.cols());
    dy.create(src.rows(), src.cols());
    mmf::check(mmf_compute_derivative_images(mmf::default_context().get(), src.ptr(), src.step(), src.cols(), src.rows(), dx.ptr(),
                                             dx.step(), dy.ptr(), dy.step()),
               "computeDerivativeImages");
}
