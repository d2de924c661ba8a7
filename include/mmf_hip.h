/*
 * mmf_hip.h -- C ABI of the MI355X (gfx950) dense-tracking hot path of MultiMotionFusion.
 *
 * This is the drop-in boundary: plain C, raw device/host pointers and sizes, no C++ / torch /
 * Eigen / OpenCV types.  Each entry point names the reference interface it replaces
 * (paths relative to the reference tree).  The C++ shim classes with the reference's own
 * names (multimotionfusion_amd/cpp/) and the Python ctypes mirror
 * (multimotionfusion_amd/_capi.py) are thin forwards onto this header.
 *
 * Conventions
 *   - every `*_dev` / "device" pointer is HBM memory of the context's device; everything else
 *     is host memory.  2-D images are passed as (pointer, step_bytes) like the reference's
 *     DeviceArray2D / PtrStep (Core/Cuda/containers/kernel_containers.hpp:48-91); step 0
 *     means dense (step = cols * sizeof(element)).
 *   - vertex / normal maps are planar float32 [3*rows][cols] (x plane, y plane, z plane;
 *     Core/Cuda/reduce.cu:261-263), invalid = NaN in the x plane.
 *   - 3x3 matrices are row major float[9] (types.cuh:61-73), poses row major float[16].
 *   - all work is enqueued on the context's stream; functions that return results in host
 *     memory synchronise that stream before returning (the reference's *Step functions do the
 *     same, reduce.cu:452-456); the others are asynchronous on the stream -- call
 *     mmf_ctx_synchronize() (the reference relies on the default stream's ordering).
 *   - return value: MMF_OK (0) or a negative mmf_status; mmf_last_error() gives the message for
 *     the calling thread.  (The reference prints and exit(-1)s, convenience.cuh:74-83; the C++
 *     shims reproduce that on a non-zero status.)
 */
#ifndef MMF_HIP_H_
#define MMF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMF_ABI_VERSION 3

typedef enum {
    MMF_OK = 0,
    MMF_ERR_INVALID = -1, /* bad argument (null pointer, size, alignment) */
    MMF_ERR_HIP = -2,     /* a HIP runtime call failed                     */
    MMF_ERR_NO_DEVICE = -3,
    MMF_ERR_STATE = -4 /* call order violated (e.g. initRGB before initICP) */
} mmf_status;

typedef struct mmf_ctx mmf_ctx;   /* device + stream + reduction scratch              */
typedef struct mmf_odom mmf_odom; /* replaces class RGBDOdometry (RGBDOdometry.h:31)  */

/* types.cuh:75-81 (DataTerm), 16 bytes */
typedef struct {
    int16_t zero_x, zero_y;
    int16_t one_x, one_y;
    float diff;
    uint8_t valid;
    uint8_t pad_[3];
} mmf_dataterm;

/* types.cuh:83-99 (CameraModel) */
typedef struct {
    float fx, fy, cx, cy;
} mmf_camera;

int mmf_abi_version(void);
const char *mmf_last_error(void);

/* private_stream != 0: create and own a non-blocking stream (`stream` is ignored).
 * private_stream == 0: run on the caller's hipStream_t `stream`; NULL is the device's default
 * (null) stream, which is what the reference uses throughout. */
int mmf_ctx_create(int device, void *stream, int private_stream, mmf_ctx **out);
void mmf_ctx_destroy(mmf_ctx *ctx);
int mmf_ctx_synchronize(mmf_ctx *ctx);
void *mmf_ctx_stream(mmf_ctx *ctx);
/* name of the device the context runs on, e.g. "gfx950:sramecc+:xnack-" */
int mmf_ctx_device_name(mmf_ctx *ctx, char *buf, size_t buflen);

/* ---------------------------------------------------------------------------------------
 * Device entry points: one per host function of Core/Cuda/cudafuncs.cuh:64-193.
 * (`pyrDown(ushort)` cudafuncs.cuh:167 has no caller in the reference and is not provided.)
 * `threads, blocks` of the reference signatures are chosen internally for gfx950.
 * ------------------------------------------------------------------------------------- */

/* icpStep, cudafuncs.cuh:64-82 / reduce.cu:399-473.
 * A_host[36] row major symmetric, b_host[6], residual_host[2] = {sum r^2, inliers}.
 * err_map_dev: optional cols x rows float32 image standing in for icpErrorSurface. */
int mmf_icp_step(mmf_ctx *ctx, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                 size_t vmap_curr_step, const float *nmap_curr, size_t nmap_curr_step,
                 const float Rprev_inv[9], const float tprev[3], const mmf_camera *intr,
                 const float *vmap_g_prev, size_t vmap_g_prev_step, const float *nmap_g_prev,
                 size_t nmap_g_prev_step, float dist_thres, float angle_thres, int cols, int rows,
                 float *A_host, float *b_host, float *residual_host, float *err_map_dev,
                 size_t err_map_step);

/* computeRgbResidual, cudafuncs.cuh:113-132 / reduce.cu:867-945.  last/next masks are not
 * read by the reference (MASK_RGB_RESIDUAL undefined) and are not taken. */
int mmf_compute_rgb_residual(mmf_ctx *ctx, float min_scale, const int16_t *dIdx, size_t dIdx_step,
                             const int16_t *dIdy, size_t dIdy_step, const float *last_depth,
                             size_t last_depth_step, const float *next_depth,
                             size_t next_depth_step, const uint8_t *last_image,
                             size_t last_image_step, const uint8_t *next_image,
                             size_t next_image_step, mmf_dataterm *corres_dev,
                             float max_depth_delta, const float kt[3], const float krkinv[9],
                             int cols, int rows, int *sigma_sum_host, int *count_host,
                             float *err_map_dev, size_t err_map_step);

/* rgbStep, cudafuncs.cuh:84-97 / reduce.cu:609-661.  cloud = AoS float3 (dense). */
int mmf_rgb_step(mmf_ctx *ctx, const mmf_dataterm *corres_dev, float sigma, const float *cloud_dev,
                 float fx, float fy, const int16_t *dIdx, size_t dIdx_step, const int16_t *dIdy,
                 size_t dIdy_step, float sobel_scale, int cols, int rows, float *A_host,
                 float *b_host);

/* so3Step, cudafuncs.cuh:99-110 / reduce.cu:1092-1150.  A_host[9], b_host[3], residual[2]. */
int mmf_so3_step(mmf_ctx *ctx, const uint8_t *last_image, size_t last_image_step,
                 const uint8_t *next_image, size_t next_image_step, const float image_basis[9],
                 const float kinv[9], const float krlr[9], int cols, int rows, float *A_host,
                 float *b_host, float *residual_host);

/* createVMap / createNMap, cudafuncs.cuh:134-142 / cudafuncs.cu:136-205 (mask unused, :119) */
int mmf_create_vmap(mmf_ctx *ctx, const mmf_camera *intr, const float *depth, size_t depth_step,
                    int cols, int rows, float *vmap, size_t vmap_step, float depth_cutoff);
int mmf_create_nmap(mmf_ctx *ctx, const float *vmap, size_t vmap_step, int cols, int rows,
                    float *nmap, size_t nmap_step);
/* tranformMaps (sic), cudafuncs.cuh:144-149; src may equal dst */
int mmf_transform_maps(mmf_ctx *ctx, const float *vmap_src, const float *nmap_src, size_t src_step,
                       int cols, int rows, const float R[9], const float t[3], float *vmap_dst,
                       float *nmap_dst, size_t dst_step);
/* copyMaps, cudafuncs.cuh:151-154: RGBA32F interleaved (dense) -> planar */
int mmf_copy_maps(mmf_ctx *ctx, const float *vmap_rgba, const float *nmap_rgba, int cols, int rows,
                  float *vmap_dst, float *nmap_dst, size_t dst_step);
/* resizeVMap / resizeNMap, cudafuncs.cuh:156-160; output is (in_cols/2) x (in_rows/2) */
int mmf_resize_vmap(mmf_ctx *ctx, const float *in, size_t in_step, int in_cols, int in_rows,
                    float *out, size_t out_step);
int mmf_resize_nmap(mmf_ctx *ctx, const float *in, size_t in_step, int in_cols, int in_rows,
                    float *out, size_t out_step);
/* imageBGRToIntensity, cudafuncs.cuh:162-163; `channels` = 3 or 4 interleaved u8 */
int mmf_image_bgr_to_intensity(mmf_ctx *ctx, const uint8_t *img, size_t img_step, int channels,
                               int cols, int rows, uint8_t *dst, size_t dst_step);
/* verticesToDepth, cudafuncs.cuh:165-167; vmap_rgba dense RGBA32F */
int mmf_vertices_to_depth(mmf_ctx *ctx, const float *vmap_rgba, int cols, int rows, float cutoff,
                          float *dst, size_t dst_step);
/* projectToPointCloud, cudafuncs.cuh:170-173; cloud dense AoS float3 */
int mmf_project_to_point_cloud(mmf_ctx *ctx, const float *depth, size_t depth_step, int cols,
                               int rows, const mmf_camera *intr, int level, float *cloud);
/* pyrDownGaussF / pyrDownUcharGauss, cudafuncs.cuh:182-186; dst is (cols/2) x (rows/2) */
int mmf_pyr_down_gauss_f(mmf_ctx *ctx, const float *src, size_t src_step, int src_cols,
                         int src_rows, float *dst, size_t dst_step);
int mmf_pyr_down_uchar_gauss(mmf_ctx *ctx, const uint8_t *src, size_t src_step, int src_cols,
                             int src_rows, uint8_t *dst, size_t dst_step);
/* computeDerivativeImages, cudafuncs.cuh:191-193 */
int mmf_compute_derivative_images(mmf_ctx *ctx, const uint8_t *src, size_t src_step, int cols,
                                  int rows, int16_t *dx, size_t dx_step, int16_t *dy,
                                  size_t dy_step);

/* ---------------------------------------------------------------------------------------
 * RGBDOdometry (Core/Utils/RGBDOdometry.h:31-137).  GPUTexture* arguments of the reference
 * become dense device images; Eigen types become float arrays.
 * ------------------------------------------------------------------------------------- */
#define MMF_NUM_PYRS 3 /* RGBDOdometry.h:72 */

/* RGBDOdometry::RGBDOdometry, RGBDOdometry.h:34-36 (maskID is unused by the kernels) */
int mmf_odom_create(mmf_ctx *ctx, int width, int height, float cx, float cy, float fx, float fy,
                    float dist_thresh, float angle_thresh, mmf_odom **out);
void mmf_odom_destroy(mmf_odom *o);

/* Model::generateCUDATextures (Core/Model/Model.cpp:359-388): level-0 filtered depth ->
 * 3-level pyramid owned by the odometry object (returned for sharing between models). */
int mmf_odom_build_depth_pyramid(mmf_odom *o, const float *depth_l0, size_t step);
/* RGBDOdometry::initICP(depthPyramid, maskPyramid, cutoff), RGBDOdometry.h:41-43.  depth_pyr
 * may be NULL to use the pyramid built by mmf_odom_build_depth_pyramid. */
int mmf_odom_init_icp(mmf_odom *o, const float *const depth_pyr[MMF_NUM_PYRS],
                      const size_t steps[MMF_NUM_PYRS], float depth_cutoff);
/* RGBDOdometry::initICP(GPUTexture*,GPUTexture*,cutoff), RGBDOdometry.h:44 */
int mmf_odom_init_icp_from_prediction(mmf_odom *o, const float *vert_rgba, const float *norm_rgba,
                                      float depth_cutoff);
/* RGBDOdometry::initICPModel, RGBDOdometry.h:47 */
int mmf_odom_init_icp_model(mmf_odom *o, const float *vert_rgba, const float *norm_rgba,
                            float depth_cutoff, const float pose[16]);
/* RGBDOdometry::initRGB / initRGBModel / initFirstRGB, RGBDOdometry.h:49-53.
 * Must follow the matching initICP* call (RGBDOdometry.cpp:197,202). */
int mmf_odom_init_rgb(mmf_odom *o, const uint8_t *rgb, size_t step, int channels);
int mmf_odom_init_rgb_model(mmf_odom *o, const uint8_t *rgb, size_t step, int channels);
int mmf_odom_init_first_rgb(mmf_odom *o, const uint8_t *rgb, size_t step, int channels);

/* RGBDOdometry::getIncrementalTransformation, RGBDOdometry.h:56-58.
 * trans[3] / rot[9] in-out (host).  icp_err_dev / rgb_err_dev: width x height float32 device
 * images (dense) standing in for the two cudaSurfaceObject_t, or NULL.
 * The whole Gauss-Newton schedule runs device-resident (no host round trip per iteration);
 * the call returns after the stream has drained and trans/rot/stats are valid. */
int mmf_odom_get_incremental_transformation(mmf_odom *o, float trans[3], float rot[9],
                                            int rgb_only, float icp_weight, int pyramid,
                                            int fast_odom, int so3, float *icp_err_dev,
                                            float *rgb_err_dev);

/* public result members, RGBDOdometry.h:62-69 */
typedef struct {
    float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
    double lastA[36];
    double lastb[6];
    int iterations_run;
    int so3_iterations_run;
} mmf_odom_stats;
int mmf_odom_get_stats(mmf_odom *o, mmf_odom_stats *out);
/* RGBDOdometry::getCovariance, RGBDOdometry.h:60: inverse of lastA (6x6, row major) */
int mmf_odom_get_covariance(mmf_odom *o, double cov[36]);

/* Introspection for parity tests: device pointer of an internal pyramid buffer.
 * names: vmaps_curr nmaps_curr vmaps_g_prev nmaps_g_prev last_depth next_depth depth_pyr cloud
 *        last_image next_image last_next_image dIdx dIdy corres */
int mmf_odom_buffer(mmf_odom *o, const char *name, int level, void **dev_ptr, size_t *bytes);
/* synchronous copy of that buffer into host memory (host_bytes must equal its size) */
int mmf_odom_download(mmf_odom *o, const char *name, int level, void *host_dst, size_t host_bytes);

/* Timing hook for bench.py: enqueue `reps` back-to-back launches of the level-`level` ICP
 * reduction kernel on the odometry object's current maps and pose (no host work in
 * between), bracketed by HIP events on the context's stream; returns the mean time per launch.
 * variant: 0 = the shipped launch geometry, else PX * 10000 + BLOCK (tuning sweeps). */
int mmf_odom_time_icp_kernel(mmf_odom *o, int level, int reps, int variant, float *mean_us_out);
/* Measurement mode for bench.py: while on, every launch of the Gauss-Newton loop's two kernels -- the producer
 * (ICP J^T J reduction + photometric correspondence pass of one iteration) and the photometric Jacobian /
 * solve step -- carries its own start / stop HIP events (the dispatch's begin / end timestamps, the same
 * ones a kernel trace reports), and the whole chain of a getIncrementalTransformation two more.  Sums, minima and
 * launch counts accumulate per pyramid level until the mode is switched (which also clears them). */
typedef struct {
    double producer_us_sum[MMF_NUM_PYRS], producer_us_min[MMF_NUM_PYRS];
    double rgb_step_us_sum[MMF_NUM_PYRS], rgb_step_us_min[MMF_NUM_PYRS];
    int producer_launches[MMF_NUM_PYRS], rgb_step_launches[MMF_NUM_PYRS];
    double chain_us_sum; /* odom_begin .. last step of one tracking call, on the device */
    int chains;
} mmf_odom_timing;
int mmf_odom_enable_timing(mmf_odom *o, int mode); /* 0 off, 1 the chain only (two events per call), 2 every kernel too */
int mmf_odom_get_timing(mmf_odom *o, mmf_odom_timing *out);

/* ---------------------------------------------------------------------------------------
 * Surfel model: Core/Model/Model.{h,cpp} (store, fuse, clean, initialise) and
 * Core/Model/ModelProjection.{h,cpp} (index map, splat prediction), without OpenGL.
 * The surfel store is a structure of float4 arrays in HBM; "textures" are dense device images.
 * All images are cols x rows, dense; rgb is interleaved u8 x 3 as uploaded from FrameData.rgb,
 * depth is float32 metres (0 = invalid), mask is u8.
 * ------------------------------------------------------------------------------------- */
typedef struct mmf_model mmf_model; /* replaces class Model + its ModelProjection */

/* Model::Model (Model.h:120-130): id doubles as the mask value of the model's pixels;
 * conf_threshold = confGlobalInit (10) / confObjectInit (0.01); max_surfels 0 = 1024*1024
 * (Model::MAX_VERTICES, Model.cpp:119-126). */
int mmf_model_create(mmf_ctx *ctx, int width, int height, float cx, float cy, float fx, float fy,
                     unsigned char id, float conf_threshold, int max_surfels, mmf_model **out);
void mmf_model_destroy(mmf_model *m);
/* Model::overridePose / getPose (Model.h:196-205): row-major camera-to-world 4x4 */
int mmf_model_set_pose(mmf_model *m, const float pose[16]);
int mmf_model_get_pose(mmf_model *m, float pose[16]);
/* Model::lastCount (Model.h:227) */
int mmf_model_count(mmf_model *m, unsigned *count);

/* MultiMotionFusion::filterDepth (MultiMotionFusion.cpp:897-904, depth_bilateral_metric.frag) */
int mmf_filter_depth(mmf_ctx *ctx, const float *depth, int cols, int rows, float max_depth, float *out);

/* computeFeedbackBuffers + Model::initialise (MultiMotionFusion.cpp:197-205, Model.cpp:267-312) */
int mmf_model_initialise(mmf_model *m, const uint8_t *rgb, const float *depth_raw,
                         const float *depth_filtered, int time, float max_depth);
/* Model::predictIndices -> ModelProjection::predictIndices (ModelProjection.cpp:94-143) */
int mmf_model_predict_indices(mmf_model *m, int time, float depth_cutoff, int time_delta);
/* Model::combinedPredict(ACTIVE) -> ModelProjection::combinedPredict (ModelProjection.cpp:187-269) */
int mmf_model_combined_predict(mmf_model *m, float depth_cutoff, int time, int max_time, int time_delta);
/* ModelProjection::synthesizeDepth (ModelProjection.cpp:275-335, depth_splat.frag): the splat's depth
 * only, into the model's "depth" image (float32 metres, 0 where no surfel lands); conf_threshold is
 * an argument here as in the reference (MultiMotionFusion.cpp:809-810 passes initConfThresGlobal) */
int mmf_model_synthesize_depth(mmf_model *m, float depth_cutoff, float conf_threshold, int time, int max_time,
                               int time_delta);
/* Model::fuse (Model.cpp:893-1048); weighting = Model::computeFusionWeight(weightMultiplier) */
int mmf_model_fuse(mmf_model *m, int time, const uint8_t *rgb, const uint8_t *mask, const float *depth_raw,
                   const float *depth_filtered, float depth_cutoff, float weighting);
/* Model::clean (Model.cpp:1050-1182); outlier_coeff = GPUSetup::outlierCoefficient (GUI default 3) */
int mmf_model_clean(mmf_model *m, int time, int time_delta, float depth_cutoff, const float *depth_filtered,
                    const uint8_t *mask, float outlier_coeff);
/* Model::performFillIn (Model.cpp:1607-1616) and MultiMotionFusion::requiresFillIn (:877-895) */
int mmf_model_perform_fill_in(mmf_model *m, const uint8_t *rgb, const float *depth_filtered,
                              int frame_to_frame_rgb, int lost);
int mmf_model_requires_fill_in(mmf_model *m, float ratio, int *result);
/* Model::downloadMap (Model.cpp:1353-1384): count_out surfels of 12 floats
 * {x y z conf | colour24 unused initTime timestamp | nx ny nz radius} (Vertex::SIZE = 48) */
int mmf_model_download_map(mmf_model *m, float *host_aos, unsigned max_surfels, unsigned *count_out);
/* Model::getModel() (Model.h:297; OutputBuffer of Core/Model/Buffers.h:3-6): the surfel store as it stands, in place --
 * three device arrays of float4 {xyz, confidence}, {colour24, unused, initTime, timestamp}, {normal xyz, radius} and the
 * number of surfels; valid until this model's next fuse / clean / initialise. */
int mmf_model_surfel_arrays(mmf_model *m, const float **pos_conf, const float **colour_time, const float **normal_radius,
                            unsigned *count);
/* inverse of download (tests, -restore) */
int mmf_model_upload_map(mmf_model *m, const float *host_aos, unsigned count);
/* device image behind a GPUTexture getter (ModelProjection.h:52-77, Model.h:232-244).
 * names: index(u32) vertConf colorTime normRad (float4) -- sparse index map;
 *        image(rgba8) vertexConf normalRadius (float4) time(u16) -- splat prediction;
 *        fillVertex fillNormal (float4) fillImage (rgba8) -- fill-in */
int mmf_model_texture(mmf_model *m, const char *name, void **dev_ptr, size_t *bytes);
/* Model::setMaxDepth / setConfidenceThreshold / getConfidenceThreshold / getID (Model.h:222-230, 307) */
int mmf_model_set_max_depth(mmf_model *m, float max_depth);
int mmf_model_set_confidence_threshold(mmf_model *m, float conf_threshold);
float mmf_model_confidence_threshold(mmf_model *m);
int mmf_model_id(mmf_model *m);

/* ---------------------------------------------------------------------------------------
 * Orchestrator: MultiMotionFusion::processFrame / predict (Core/MultiMotionFusion.h:78-86,
 * .cpp:207-854, 863-875): the global (camera) model plus the object models of the `models` list, each
 * with its own RGBDOdometry, tracked / predicted / fused / cleaned per frame.  With
 * enable_multiple_models == 0 this is the static-scene configuration (all-zero mask, :268-275).
 * The segmentation proper (gSLICr + dense CRF, or the ground-truth id image of
 * Segmentation.cpp:89-150), relocalisation and loop closure stay in the reference's front-end: the
 * segmentation RESULT is handed in per frame (mmf_segmentation) or pulled through a callback at the point
 * where the reference calls performSegmentation (:412).
 * Every model runs on its own stream ("lane"); every public call returns with the fusion's own
 * stream (the context's) ordered after all lanes.
 * ------------------------------------------------------------------------------------- */
typedef struct mmf_fusion mmf_fusion;

typedef struct {
    int time_delta;            /* MultiMotionFusion ctor timeDelta (GUI default 200) */
    float conf_global_init;    /* confGlobalInit, GUI default 10 */
    float icp_weight;          /* GUI default 10 */
    float depth_cutoff;        /* bilateral filter maxD, GUI default */
    float max_depth_processed; /* 20 (MultiMotionFusion.cpp:53) */
    int rgb_only, pyramid, fast_odom, so3, frame_to_frame_rgb;
    float outlier_coeff;       /* GPUSetup::outlierCoefficient, GUI default 3 */
    int fill_in;               /* global model is created with fill-in enabled */
    int max_surfels;           /* 0 = Model::MAX_VERTICES */
    float conf_object_init;    /* confObjectInit, GUI default 0.01 (-confO) */
    int enable_multiple_models; /* setEnableMultipleModels (MultiMotionFusion.h:262) */
    int preallocated_models;   /* preallocateModels(count) (:125-131; -a) */
    int error_recording;       /* Model(..., enableErrorRecording): per-model ICP / RGB error images */
    int pose_logging;          /* enablePoseLogging: Model::poseLog, exportPoses */
    int max_object_surfels;    /* capacity of an object model's store, 0 = max_surfels */
    int batch_tracking;        /* 1: the Gauss-Newton chains of all models of a frame run as ONE chain of launches with
                                  gridDim.y = model (default); 0: one chain per model on the model's own stream */
} mmf_fusion_config;

/* SegmentationResult (Core/Segmentation/Segmentation.h:32-70) as far as processFrame consumes it */
typedef struct {
    unsigned id;                /* ModelData::id */
    unsigned super_pixel_count; /* 0: the model was not seen in this frame (:607) */
    float avg_confidence;       /* raises the object's confidence threshold, capped at 9 (:616-620) */
    float depth_mean, depth_std; /* Model::setMaxDepth(depth_mean + 1.2 depth_std) (:409, :486, :586) */
} mmf_segmentation_model;

typedef struct {
    const uint8_t *mask;  /* DEVICE, width*height u8: fullSegmentation = model id per pixel */
    int has_new_label;    /* hasNewLabel: spawn an object model for the id mmf_fusion_next_model_id() (:469-487) */
    int n_models;         /* entries of model_data: the active models in list order, then the new label's */
    const mmf_segmentation_model *model_data; /* HOST; NULL: no max-depth / confidence / unseen updates */
} mmf_segmentation;

/* one processFrame call (MultiMotionFusion.h:78-80): FrameData + the optional arguments */
typedef struct {
    const uint8_t *rgb;  /* DEVICE u8 x 3 interleaved (FrameData::rgb) */
    const float *depth;  /* DEVICE float32 metres (FrameData::depth) */
    long long timestamp;
    const float *in_pose; /* HOST 4x4 or NULL */
    float weight_multiplier;
    int bootstrap;
    /* odom_cfg.init == "kp" (:312-384): one 4x4 (HOST) per active model in list order = the
     * RigidRANSAC transformation of Model::getLastTrackTransform; NULL = no pose initialisation */
    const float *init_transforms;
    int n_init_transforms;
    int icp_refine; /* odom_cfg.icp_refine */
    /* enable_multiple_models: the result of performSegmentation(frame) for THIS frame; NULL = ask the callback */
    const mmf_segmentation *segmentation;
    /* optional hint (both or neither): the buffers the NEXT processFrame call will be given.  Their sensor-side
     * preparation (mmf_fusion_prefetch_frame) is then enqueued inside this call, while the host waits for this frame's
     * pose, instead of by a separate call afterwards.  Same contract: the next call gets these buffers, unchanged. */
    const uint8_t *next_rgb;
    const float *next_depth;
} mmf_frame;

/* called where the reference calls performSegmentation (:412): every model of this frame has been tracked
 * (poses, ICP / RGB error images and predictions of the previous fusion are readable); fill *out. */
typedef int (*mmf_segmentation_fn)(void *user, mmf_fusion *f, const mmf_frame *frame, mmf_segmentation *out);

int mmf_fusion_default_config(mmf_fusion_config *cfg);
int mmf_fusion_create(mmf_ctx *ctx, int width, int height, float cx, float cy, float fx, float fy,
                      const mmf_fusion_config *cfg, mmf_fusion **out);
void mmf_fusion_destroy(mmf_fusion *f);
int mmf_fusion_preallocate_models(mmf_fusion *f, unsigned count); /* preallocateModels (:125-131) */
/* processFrame(frame, inPose, weightMultiplier, gt_pose, bootstrap): rgb = u8 x 3 interleaved and
 * depth = float32 metres, both already in HBM (the reference uploads FrameData to GL textures
 * here, :221,261); in_pose may be NULL.  Returns MMF_ERR_INVALID with "invalid image data" where
 * the reference returns false (:209-212). */
int mmf_fusion_process_frame(mmf_fusion *f, const uint8_t *rgb, const float *depth, long long timestamp,
                             const float *in_pose, float weight_multiplier, int bootstrap);
/* processFrame with every optional input (multiple models, pose initialisation of every model) */
int mmf_fusion_process_frame_ex(mmf_fusion *f, const mmf_frame *frame);
/* processFrame(const FrameData&) with the frame in HOST memory: rgb, depth and the optional id image
 * (mask_host != NULL: FrameData::mask, already mapped to model ids) are staged through pinned double buffers
 * and uploaded on the fusion's stream (:221, :261, :416). */
int mmf_fusion_process_frame_host(mmf_fusion *f, const uint8_t *rgb_host, const float *depth_host,
                                  const uint8_t *mask_host, int has_new_label, long long timestamp,
                                  const float *in_pose, float weight_multiplier, int bootstrap);
/* The same with the NEXT call's host frame announced (both pointers or neither; the next call must be given exactly these
 * pointers, contents unchanged): the host-memory form of mmf_frame::next_rgb / next_depth.  That frame is staged and
 * uploaded on a stream of its own while this frame is tracked, and its sensor-side preparation overlaps this frame's
 * fusion -- what a front-end that reads frames ahead (GUI/MainController.cpp:547-590: logReader->getNext()) gets for free. */
int mmf_fusion_process_frame_host_next(mmf_fusion *f, const uint8_t *rgb_host, const float *depth_host,
                                       const uint8_t *mask_host, int has_new_label, long long timestamp,
                                       const float *in_pose, float weight_multiplier, int bootstrap,
                                       const uint8_t *next_rgb_host, const float *next_depth_host);
/* processFrame with the tracker initialised from keypoint tracks: odom_cfg.init == "kp"
 * (MultiMotionFusion.cpp:312-384).  init_transform = RigidRANSAC::Result::transformation of
 * Model::getLastTrackTransform (row-major 4x4, mmf_ransac_estimate): the camera model's pose becomes
 * pose * init_transform (:331), the map is predicted / fused / cleaned once at that pose with
 * Model::computeFusionWeight(weight_multiplier) (:352-366, Model.cpp:918), then the dense tracker refines the
 * pose when icp_refine != 0 (:377-381; odom_cfg.icp_refine), else the initial pose is kept (:382-385).  On the
 * first frame (tick 1) the transformation is ignored, as in the reference.  Fails in frame-to-frame RGB mode (:370). */
int mmf_fusion_process_frame_init(mmf_fusion *f, const uint8_t *rgb, const float *depth, long long timestamp,
                                  const float *init_transform, int icp_refine, float weight_multiplier);
/* Overlap across frames: enqueue the work of the NEXT frame that depends on the sensor frame only -- the depth
 * filter, vertex / normal maps and the depth pyramid on one side stream; the intensity pyramid, the gradients and
 * the SO3 pre-alignment (last frame's image against this one's) on another -- where they run while the current
 * frame is still being fused.  The following
 * mmf_fusion_process_frame* call with the SAME rgb / depth pointers picks the results up instead of
 * recomputing them (same kernels, same bits); with other pointers the prefetch is discarded.  rgb / depth must
 * stay unchanged until that call.  Optional: without it every frame is processed on its own. */
int mmf_fusion_prefetch_frame(mmf_fusion *f, const uint8_t *rgb, const float *depth);
/* predict() (MultiMotionFusion.h:86, .cpp:863-875): re-render every model's prediction at its current pose */
int mmf_fusion_predict(mmf_fusion *f);
/* getCurrPose / getTick / setTick / getBackgroundModel / getModels (MultiMotionFusion.h:100-216) */
int mmf_fusion_reset(mmf_fusion *f); /* empty maps, identity poses, tick = 1, only the global model active */
int mmf_fusion_get_pose(mmf_fusion *f, float pose[16]);
int mmf_fusion_tick(mmf_fusion *f);
int mmf_fusion_set_tick(mmf_fusion *f, int tick);
mmf_model *mmf_fusion_model(mmf_fusion *f);   /* getBackgroundModel(), = getIndexMap()'s owner */
mmf_odom *mmf_fusion_odometry(mmf_fusion *f); /* its frameToModel */
int mmf_fusion_num_models(mmf_fusion *f);     /* getModels().size() */
mmf_model *mmf_fusion_model_at(mmf_fusion *f, int index);
mmf_odom *mmf_fusion_odometry_at(mmf_fusion *f, int index);
int mmf_fusion_num_inactive_models(mmf_fusion *f);
mmf_model *mmf_fusion_inactive_model_at(mmf_fusion *f, int index);
int mmf_fusion_next_model_id(mmf_fusion *f); /* getNextModelID(false): the id a new label must carry in the mask */
int mmf_fusion_schedule_deactivation(mmf_fusion *f, int id); /* scheduleDeactivation: applied at the next frame */
/* Model::getICPErrorTexture (which = 0) / getRGBErrorTexture (which = 1) of the index-th active model:
 * width*height float32, written by the last level-0 iteration of its tracking */
int mmf_fusion_error_texture(mmf_fusion *f, int index, int which, float **dev_ptr);
/* getTextures() (MultiMotionFusion.h:124): "RGB" (u8 x 3), "DEPTH_METRIC", "DEPTH_METRIC_FILTERED" (float32),
 * "MASK" (u8) of the current frame as device images */
int mmf_fusion_texture(mmf_fusion *f, const char *name, const void **dev_ptr, size_t *bytes);
const float *mmf_fusion_depth_filtered(mmf_fusion *f);
/* the runtime setters the front-end pushes every GUI tick (MultiMotionFusion.cpp:1064-1116,
 * GUI/MainController.cpp:641-670); they take effect at the next processFrame */
int mmf_fusion_set_rgb_only(mmf_fusion *f, int val);
int mmf_fusion_set_icp_weight(mmf_fusion *f, float val);
int mmf_fusion_set_outlier_coefficient(mmf_fusion *f, float val);
int mmf_fusion_set_pyramid(mmf_fusion *f, int val);
int mmf_fusion_set_fast_odom(mmf_fusion *f, int val);
int mmf_fusion_set_so3(mmf_fusion *f, int val);
int mmf_fusion_set_frame_to_frame_rgb(mmf_fusion *f, int val);
int mmf_fusion_set_depth_cutoff(mmf_fusion *f, float val);
int mmf_fusion_set_confidence_threshold(mmf_fusion *f, float val);
int mmf_fusion_set_enable_multiple_models(mmf_fusion *f, int val);
int mmf_fusion_get_config(mmf_fusion *f, mmf_fusion_config *out);
/* Per-rigid-body shard (one process per GPU): this process runs the GPU work of the models whose id has
 * id % world == rank (fixed when the model is created: a model leaving the list moves nobody else); the other models exist
 * as bookkeeping only (ids, thresholds, poses).  The sensor-side
 * preparation of a frame runs on every rank.  Poses of remote models are handed in by the caller (the all-gather
 * of mmf_shard_* or of the host framework). */
int mmf_fusion_set_shard(mmf_fusion *f, int rank, int world);
int mmf_fusion_owns_model(mmf_fusion *f, int index);
int mmf_fusion_set_model_pose(mmf_fusion *f, int index, const float pose[16]);
/* The shard's two exchanges over RCCL, for a front-end that runs one process per GPU (the reference is
 * single GPU and walks its Model list serially, MultiMotionFusion.cpp:312, 793-816).  RCCL is bound at run time
 * (dlopen of librccl.so.1); all collectives run, in program order, on a stream the shard owns, tied to the context's
 * stream by events where data crosses.
 *   mmf_shard_unique_id  rank 0: ncclGetUniqueId; ship the 128 bytes to the other ranks by any means
 *   mmf_shard_create     ncclCommInitRank(world, id, rank) on the context's device
 *   mmf_shard_attach     use the caller's ncclComm_t instead (not destroyed by mmf_shard_destroy)
 *   mmf_shard_broadcast_frame  the root's rgb (u8 x 3) / depth (f32) / id image (u8, may be NULL) into the same
 *                        buffers of every rank: 8 B/px, stream ordered on the context's stream -- call before processFrame
 *   mmf_shard_post_frame / mmf_shard_wait_frame   the same exchange in two halves: post starts it on the shard's own stream
 *                        (behind what the context's stream holds now) so that it overlaps the frame being processed, wait
 *                        makes the context's stream wait for the exchange posted into that slot (0 .. 7)
 *   mmf_shard_gather_poses     after processFrame: all-gather of {pose, lastICPError, lastICPCount} (18 floats per
 *                        model slot); the poses of models other ranks own land in this rank's bookkeeping.  Blocking.
 *   mmf_shard_gather_poses_begin / _end   the same exchange without stalling a frame: _begin enqueues it (pinned
 *                        buffers, an event, no synchronisation), _end waits for the OLDEST one in flight and applies it
 *                        (call it one or two frames later; up to three may be in flight)
 *                        STALENESS CONTRACT: a rank's copy of a model it does not own carries the pose of the newest
 *                        exchange that rank has applied -- the current frame's under the blocking form, one to three frames
 *                        old under _begin / _end.  Nothing on the data path reads those copies (a rank tracks and fuses its
 *                        own models only).  The pose log (mmf_fusion_pose_log / export_poses) is written by a model's OWNER
 *                        only; an object model's entry is global_pose * inverse(pose) (MultiMotionFusion.cpp:829-846) with
 *                        the global pose as that rank holds it, so on ranks other than the global model's it is exact only
 *                        under the blocking form.
 *   mmf_shard_gather_maps      what the segmentation reads of every model (Segmentation.cpp:214-223): the ICP-error image
 *                        and the confidence channel of the splat's vertex image, averaged per super-pixel
 *                        (mmf_slic_downsample) where the model lives and all-gathered: out_dev[n_models][2][nspix] on every
 *                        rank, list order, {icp, confidence}; labels = device int32 super-pixel index image
 * A model belongs to rank (id % world), whatever its position in the list. */
typedef struct mmf_shard mmf_shard;
int mmf_shard_unique_id(char id[128]);
int mmf_shard_create(mmf_ctx *ctx, int rank, int world, const char id[128], mmf_shard **out);
int mmf_shard_attach(mmf_ctx *ctx, int rank, int world, void *nccl_comm, mmf_shard **out);
void mmf_shard_destroy(mmf_shard *s);
int mmf_shard_broadcast_frame(mmf_shard *s, uint8_t *rgb, float *depth, uint8_t *mask, int width, int height, int root);
int mmf_shard_post_frame(mmf_shard *s, uint8_t *rgb, float *depth, uint8_t *mask, int width, int height, int root, int slot);
int mmf_shard_wait_frame(mmf_shard *s, int slot);
int mmf_shard_gather_poses(mmf_shard *s, mmf_fusion *f);
int mmf_shard_gather_poses_begin(mmf_shard *s, mmf_fusion *f);
int mmf_shard_gather_poses_end(mmf_shard *s, mmf_fusion *f);
int mmf_shard_gather_maps(mmf_shard *s, mmf_fusion *f, const int *labels, int spixel_size, float *out_dev);
/* host wall clock of the last processFrame call: the tracking phase (first enqueue .. last result) and the whole call */
int mmf_fusion_last_timings(mmf_fusion *f, double *tracking_s, double *frame_s);
int mmf_fusion_set_segmentation_callback(mmf_fusion *f, mmf_segmentation_fn fn, void *user);
/* exportPoses() (:1020-1045): `poses-<id>.txt` per pose-logging model under export_dir (which ends in '/') */
int mmf_fusion_export_poses(mmf_fusion *f, const char *export_dir);
/* Model::getPoseLog() of the index-th active model: ts[i], p7[7 i ..] = x y z qx qy qz qw */
int mmf_fusion_pose_log(mmf_fusion *f, int index, long long *ts, float *p7, int max_entries, int *n_out);
/* test hook: the device build's mmf_expf and the packed exponential of the two-pixel bilateral filter on n arguments
 * (the packed one is defined for x <= 0 or NaN only) */
int mmf_debug_expf(mmf_ctx *ctx, const float *x_dev, int n, float *out_mmf_dev, float *out_packed_dev);
/* test / A-B hook: 1 = run the Gauss-Newton chain as one launch per iteration where it applies (the default), 0 = always as
 * producer + step launches, -1 = what the environment says (MMF_GN_FUSED=0 turns it off).  Process wide. */
int mmf_debug_set_gn_fused(int on);
/* The one-launch chain spins on its own workgroups (a count barrier inside every launch); the library checks the launch's
 * occupancy before it uses that chain, but another process on the same GPU can still keep a launch from becoming resident as
 * a whole.  A launch that gives up marks the chain's result void; the call that waits for it tracks the frame again on the
 * two-launch chain from the pose it started with (RGBDOdometry.cpp:464-467: the call returns a pose, it never aborts) and the
 * process stops using the one-launch chain.  recoveries: how often that happened; one_launch_chain_in_use: 0 afterwards.
 * Either pointer may be NULL. */
int mmf_gn_chain_status(int *recoveries, int *one_launch_chain_in_use);
/* test hook: the next n one-launch chains of this process give up at their third launch */
int mmf_debug_force_gn_fault(int n);
/* test hooks of the object models' walk in the one-launch chain (csrc/gn_fused.hpp: an object model's ICP term covers only the
 * rectangle of sensor pixels its prediction can reach under the iteration's pose).  Checking mode (process wide): such models
 * walk the WHOLE image and count the correspondences icpStep accepts outside that rectangle.  mmf_debug_odom_sparse_outside:
 * that count over the last chain of an odometry (0 or the rectangle is wrong), whether the chain walked it by its extents, and
 * rect = the level-0 rectangle of the chain's last launch {x0, y0, width, rows}, the most passes a level-0 rectangle took, and
 * whether the rectangles were derived (1) or the whole image (0), then the lanes the box of the model's own depth needed at
 * levels 0 / 1 / 2 (what its next chain is sized by).  Any pointer may be NULL. */
int mmf_debug_set_sparse_check(int on);
/* test hook: object models get n workgroups per launch of the one-launch chain at every level (0: the default, 32 / 24 / 16);
 * with n = 1 no object's extent fits, the chain reports it, the frame is tracked again on the two-launch chain and the model's
 * next chain is sized by what its box needed -- the one-launch chain stays in use (mmf_gn_chain_status) */
int mmf_debug_set_sparse_groups(int n);
int mmf_debug_odom_sparse_outside(mmf_odom *o, unsigned *outside, int *walked_by_extent, int rect[9]);
/* test / A-B hook: 1 = enqueue the reference's first predict() of a frame (MultiMotionFusion.cpp:675) although nothing
 * inside this library reads its images before the frame's second predict() (:821) overwrites them, 0 = leave it out (the
 * default; -1 = the default).  Process wide. */
int mmf_debug_set_mid_predict(int on);
/* test / A-B hook: the early depth test of combinedPredict (splat_kernel<true>: the rasterising pass reads a pixel's key before it
 * evaluates a fragment there and skips what cannot win; same images).  1 = always, 0 = never, -1 = when the store holds two
 * surfels per pixel or more (the default).  Process wide. */
int mmf_debug_set_splat_bound(int mode);
/* test / A-B hook: the projection / fuse / clean / predict passes of the OBJECT models of a frame: 0 = model by model on the models' own
 * streams, 1 = one launch per pass for all of them, each covering the whole frame (csrc/surfel_kernels.hpp:
 * *_batched_kernel -- the same kernel bodies, gridDim.y = model), 2 = one launch per pass restricted to where each model is
 * (csrc/pass_rect.hpp: the boxes of its key-image writes, of its non-zero images and of its id in the id image), -1 = the
 * default (MMF_PASS_BATCH, or by the number of object models a GPU runs: 2 from four on, 0 below).  Same maps and images, bit
 * for bit, in all three, also across a frame in which the mode changes.  Process wide. */
int mmf_debug_set_pass_batch(int mode);
/* test / A-B hook: an OBJECT model's model-side preparation (Model::initICPModel / initRGBModel's pyramids, records and point
 * clouds) covers only the box its prediction is non-zero in -- the hull of that box now and at its previous preparation, so
 * that every buffer stays what a whole-frame preparation writes (1, the default) -- or the whole frame (0); -1 = the default
 * (MMF_PREP_RECT).  Same buffers, bit for bit.  Process wide. */
int mmf_debug_set_prep_rect(int on);
/* test / A-B hook: when a frame's model side is prepared at the end of the call before it (one model per process, the next frame
 * handed in), the beginning of its tracking (odom_begin_kernel: RGBDOdometry.cpp:221-228, 237, 252-255, 316-328) rides the
 * preparation's last launch on one more workgroup (1, the default: csrc/track_kernels.hpp, prep_batch_begin_kernel) or is a launch
 * of its own in front of the first Gauss-Newton iteration (0); -1 = the default (MMF_BEGIN_RIDER).  Same bits.  Process wide.
 * mmf_debug_begin_rider_count: chains of this process that found their beginning done. */
/* test / A-B hook: every XCD works on one contiguous eighth of a surfel pass's blocks (1, the default: csrc/surfel_kernels.hpp,
 * xcd_block) or the blocks are dealt round-robin as the workgroups are (0); -1 = the default (MMF_XCD).  Same bits.  Synchronises
 * the device; process wide. */
int mmf_debug_set_xcd(int on);
/* the block workgroup `block` of a launch of `blocks` works on under that mapping (a host function: needs no device) */
unsigned mmf_debug_xcd_block(unsigned block, unsigned blocks);
int mmf_debug_set_begin_rider(int on);
int mmf_debug_begin_rider_count(void);
/* test / A-B hook: object models in the producer + step chain (csrc/extent.hpp, ChainGeom: their passes skip the blocks outside
 * the model's own depth and walk the images with a quarter of the workgroups).  1 = on, 0 = every model is tracked like the
 * camera model, -1 = the default (MMF_TRACK_CULL, on).  Process wide. */
int mmf_debug_set_track_cull(int mode);
/* test hook: the 24-bit depth key of combo_splat.frag's gl_FragDepth for n depths, as the rasterising pass computes it (the
 * division by 2 max_depth as three multiply-adds, csrc/surfel_kernels.hpp: splat_depth24_fast) and with the division */
int mmf_debug_depth_keys(mmf_ctx *ctx, const float *z_dev, int n, float max_depth, unsigned *fast_dev, unsigned *divided_dev);
/* Model::computeFusionWeight (Model.cpp:876-891) for pose / lastPose (host 4x4), exported for tests */
int mmf_compute_fusion_weight(const float pose[16], const float last_pose[16], float multiplier, float *out);

/* ---- keypoint descriptor matching ------------------------------------
 * PointTracker::addKeypoints, Core/Utils/PointTracker.cpp:100-114:
 *     cv::BFMatcher(cv::NORM_L2, true).match(current, previous, matches);   // query, train
 *     keep when min_feature_distance < epsilon || match.distance <= min_feature_distance
 * query [nq x dim], train [nt x dim]: dense float32 rows on the device (SuperPoint: dim = 256);
 * dim must be a multiple of 8.  train_idx [nq] receives the matched train row or -1, distance [nq]
 * the L2 distance of a match (0 when unmatched); both on the device.  The nq x nt Gram matrix runs
 * on the f32 matrix cores (csrc/match_kernels.hpp).  Asynchronous on the context's stream. */
int mmf_match_descriptors(mmf_ctx *ctx, const float *query, int nq, const float *train, int nt, int dim,
                          float max_distance, int *train_idx, float *distance);

/* ---- SuperPoint keypoint network --------------------------------------------
 * Replaces the un-vendored super_point_inference package as the reference uses it:
 *     kp_predictor = std::make_shared<SuperPoint>(keypoint_predictor_path);       Core/MultiMotionFusion.cpp:78
 *     std::tie(coordinates[i], descriptors[i]) = kp_predictor->getFeatures(img);  Core/MultiMotionFusion.cpp:233
 * weights: 24 HOST arrays {weight, bias} of conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa
 * convPb convDa convDb in PyTorch layout [Cout][Cin][k][k] / [Cout] (the state dict of SuperPointNet.pt;
 * reading the file is the caller's business).  The object serves images up to max_width x max_height
 * (sides multiples of 8) and returns at most max_keypoints keypoints.  The 3x3 and 1x1 convolutions run
 * on the f32 matrix cores (csrc/superpoint_kernels.hpp). */
typedef struct mmf_superpoint mmf_superpoint;
int mmf_superpoint_create(mmf_ctx *ctx, const float *const *weights, int max_width, int max_height,
                          int max_keypoints, mmf_superpoint **out);
void mmf_superpoint_destroy(mmf_superpoint *sp);
/* the network alone: image = DEVICE pointer to interleaved u8 (1 = grey, 3 / 4 = RGB[A]); leaves the detector
 * logits [H/8][W/8][65], the normalised coarse descriptors [H/8][W/8][256] and the heat map [H][W] on the
 * device.  Asynchronous on the context's stream. */
int mmf_superpoint_forward(mmf_superpoint *sp, const uint8_t *image, int width, int height, int channels);
/* copy a result of the last forward pass to the host: which = 0 logits, 1 coarse descriptors, 2 heat map;
 * count must be the exact number of floats.  Synchronous. */
int mmf_superpoint_download(mmf_superpoint *sp, int which, float *host, size_t count);
/* SuperPoint::getFeatures: forward pass, heat >= conf_thresh, greedy non-maximum suppression (Chebyshev
 * radius nms_dist, strongest first), border band removed, descriptors sampled bilinearly and normalised.
 * xy [max_keypoints][2] (pixels), conf [max_keypoints], desc [max_keypoints][256]: HOST arrays, strongest
 * keypoint first; *count = keypoints written.  The reference's `coordinates` are xy / (width, height)
 * (PointTracker.cpp:40-41).  Synchronous. */
int mmf_superpoint_get_features(mmf_superpoint *sp, const uint8_t *image, int width, int height, int channels,
                                float conf_thresh, int nms_dist, int border, int *xy, float *conf, float *desc,
                                int *count);
/* one convolution layer (tests / tools): in [H][W][cin], out [H][W][cout] (or [H/2][W/2][cout] with pool) on
 * the device, channels-last; w [cout][cin][k][k], bias [cout] on the HOST; taps = 9 (3x3, zero pad 1) or 1;
 * cin a multiple of 32; nt = output-channel tiles of 32 per workgroup (1, 2, 4; 0 = automatic). Synchronous. */
int mmf_superpoint_conv(mmf_ctx *ctx, const float *in, int height, int width, int cin, const float *w,
                        const float *bias, int cout, int taps, int relu, int pool, int nt, float *out);

/* ---- super-pixel resampling for the segmentation ------------------------------
 * Slic::downsample<float>(image, channel) (Core/Segmentation/Slic.h:48-83), Slic::downsampleThresholded<float>
 * (:87-126), Slic::downsample() (Slic.cpp:82-112) and Slic::upsample<unsigned char> (Slic.h:133-146), which
 * Segmentation.cpp:177-178,218-221,683 runs on the CPU after downloading the full-resolution ICP-error and
 * vertex-confidence textures of every model.  labels = gSLICr's segmentation mask (int32 [height][width],
 * values in [0, n), n = (width / spixel_size) * (height / spixel_size)); everything is a DEVICE pointer and
 * the calls are asynchronous on the context's stream.  Results are those of the reference's loops bit for
 * bit: float sums run in pixel order, the in-place division and the empty-super-pixel substitution
 * (resampleEmptyIndex, including its spixelY quirk) are reproduced.
 *   image [height][width][channels] float32, `channel` selected (the ICP-error map: channels 1; the
 *   vertex-confidence map: channels 4, channel 3); thresholded != 0: only values > min_threshold count
 *   (the depth map with 0.02); out [n] float32; counts_out (optional) [n] int32 = spixelCounts. */
int mmf_slic_downsample(mmf_ctx *ctx, const int *labels, int width, int height, int spixel_size, const float *image,
                        int channels, int channel, int thresholded, float min_threshold, float *out, int *counts_out);
/* rgb [height][width][channels] u8 (3 or 4); out [n][3] u8 = integer means of input channels (2, 1, 0) */
int mmf_slic_downsample_rgb(mmf_ctx *ctx, const int *labels, int width, int height, int spixel_size, const uint8_t *rgb,
                            int channels, uint8_t *out);
/* out[i] = map[labels[i]] */
int mmf_slic_upsample_u8(mmf_ctx *ctx, const int *labels, int width, int height, const uint8_t *map, int nspix,
                         uint8_t *out);

/* ---- keypoint-based pose initialisation: RigidRANSAC (Core/Utils/RigidRANSAC.h:6-32, .cpp:73-180) ----
 * Host code, like the reference's (a few dozen keypoint tracks): p0, p1 are HOST arrays of n 3-D points
 * (row-major n x 3), mask an optional n-byte selection; T receives the row-major 4x4 of T_01 with
 * p0 ~ T_01 p1.  mmf_rigid_fit = the free function fit() (RigidRANSAC.cpp:73-120).  A mmf_ransac object owns
 * the std::default_random_engine of the reference's class, so successive estimates continue its sequence.
 * estimate(): error = mean inlier distance of the best model (+inf when none beat the initial all-points
 * fit); inlier (optional, n bytes) flags the rows of the HASH-SORTED correspondence order used internally
 * (RigidRANSAC.cpp:36-58), exactly what Result::inlier holds in the reference; *has_inlier = 0 when empty. */
typedef struct mmf_ransac mmf_ransac;
int mmf_rigid_fit(const float *p0, const float *p1, int n, const unsigned char *mask, float T[16]);
int mmf_rigid_apply(const float T[16], const float *p0, const float *p1, int n, float *distance);
int mmf_ransac_create(int iterations, float inlier_threshold, float inlier_fraction, mmf_ransac **out);
void mmf_ransac_destroy(mmf_ransac *r);
int mmf_ransac_estimate(mmf_ransac *r, const float *p0, const float *p1, int n, const unsigned char *mask,
                        float T[16], float *error, unsigned char *inlier, int *has_inlier);

#ifdef __cplusplus
}
#endif
#endif /* MMF_HIP_H_ */
