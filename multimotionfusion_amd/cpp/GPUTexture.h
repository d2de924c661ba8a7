// GPUTexture.h -- handle type with the reference's name (Core/GPUTexture.h:27-66) for the device images that cross
// the Model / MultiMotionFusion API.  The reference's GPUTexture owns an OpenGL texture registered with CUDA; this
// path has no GL interop, so the handle is a VIEW of a dense image in HBM: (pointer, width, height, format).  It
// exists so that call sites written against the reference -- Model::fuse(time, GPUTexture* rgb, GPUTexture* mask, ...),
// textures[GPUTexture::RGB], getIndexMap().getSplatVertexConfTex() -- keep their shape.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

class GPUTexture {
   public:
    // the pixel formats the path uses (the reference passes GL enums: GL_RGBA32F / GL_LUMINANCE32F_ARB / GL_RGBA /
    // GL_R32F / GL_LUMINANCE8UI_EXT / GL_LUMINANCE16UI_EXT / GL_R32UI, MultiMotionFusion.cpp:160-176, ModelProjection.cpp:28-55)
    enum Format { RGB8, RGBA8, R32F, RGBA32F, R8UI, R16UI, R32UI };

    GPUTexture(const void* device_image, int width, int height, Format format, std::string name = "_")
        : myName(std::move(name)), width(width), height(height), format(format), data_(device_image) {}
    virtual ~GPUTexture() {}

    std::string myName;
    static const std::string RGB, DEPTH_METRIC, DEPTH_METRIC_FILTERED, MASK;  // GPUTexture.h:49

    const int width;
    const int height;
    const Format format;

    const void* data() const { return data_; }
    template <typename T>
    const T* ptr() const { return static_cast<const T*>(data_); }
    void rebind(const void* device_image) { data_ = device_image; }  // the frame's images move between double buffers
    static size_t bytesPerPixel(Format f) {
        switch (f) {
            case RGB8: return 3;
            case RGBA8: return 4;
            case R32F: return 4;
            case RGBA32F: return 16;
            case R8UI: return 1;
            case R16UI: return 2;
            case R32UI: return 4;
        }
        return 0;
    }
    size_t bytes() const { return (size_t)width * height * bytesPerPixel(format); }
    // cv::Mat downloadTexture() of the reference (GPUTexture.h:44): a host copy of the image (the caller synchronised
    // the producing call; every public MultiMotionFusion / Model call leaves the context's stream ordered after its work)
    std::vector<uint8_t> downloadTexture() const {
        std::vector<uint8_t> host(bytes());
        if (data_ && hipMemcpy(host.data(), data_, host.size(), hipMemcpyDeviceToHost) != hipSuccess) host.clear();
        return host;
    }

   private:
    const void* data_;
};

inline const std::string GPUTexture::RGB = "RGB";
inline const std::string GPUTexture::DEPTH_METRIC = "DEPTH_METRIC";
inline const std::string GPUTexture::DEPTH_METRIC_FILTERED = "DEPTH_METRIC_FILTERED";
inline const std::string GPUTexture::MASK = "MASK";
