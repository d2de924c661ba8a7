/*
 * mmf_oracle.h -- CPU restatement ("oracle") of MultiMotionFusion's dense tracking hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * there only as the checker / the reported CPU baseline.  The shipped path is the HIP
 * library behind include/mmf_hip.h and it fails loudly when that library is missing.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, KATs or tests for this path
 *  and its CUDA+OpenGL implementation cannot be built or run in
 * this pipeline.  This file restates the reference algorithm from reading its sources; each
 * function cites the file:line (relative to the reference tree) it follows.  Fixtures under
 * tests/golden/ are produced by THIS oracle (tests/golden/make_golden.py).
 *
 * Conventions
 *   - all images are dense row-major (pitch == cols); vertex / normal maps are planar
 *     [3*rows][cols] float32: x plane, then y plane, then z plane (reduce.cu:261-263);
 *     invalid = NaN in the x plane (cudafuncs.cu:131).
 *   - per-pixel arithmetic is float32 in the reference's operation order, built with
 *     -ffp-contract=off so that the HIP kernels (also built without contraction) can be
 *     compared bit-for-bit per pixel; reductions accumulate in double (the reference's
 *     float tree order is not reproducible on another machine, so sums are compared within
 *     a stated tolerance instead).
 *   - normalisation uses 1/sqrtf (correctly rounded) where the reference uses CUDA's
 *     approximate rsqrtf (operators.cuh:80-84); the difference is <= 2 ulp per component.
 */
#ifndef MMF_ORACLE_H_
#define MMF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* types.cuh:75-81 -- 16-byte correspondence record written for every pixel */
typedef struct {
    int16_t zero_x, zero_y; /* pixel in the last (model) image   */
    int16_t one_x, one_y;   /* pixel in the next (current) image */
    float diff;             /* next - last intensity             */
    uint8_t valid;
    uint8_t pad_[3];
} orc_dataterm;

/* ---- map kernels (cudafuncs.cu) ------------------------------------------------------- */
void orc_create_vmap(const float *depth, int cols, int rows, float fx, float fy, float cx, float cy,
                     float depth_cutoff, float *vmap);
void orc_create_nmap(const float *vmap, int cols, int rows, float *nmap);
void orc_transform_maps(const float *vsrc, const float *nsrc, int cols, int rows, const float R[9],
                        const float t[3], float *vdst, float *ndst);
void orc_copy_maps(const float *vsrc_rgba, const float *nsrc_rgba, int cols, int rows, float *vdst,
                   float *ndst);
void orc_resize_map(const float *in, int in_cols, int in_rows, int normalize, float *out);
void orc_pyrdown_gauss_f(const float *src, int src_cols, int src_rows, float *dst);
void orc_pyrdown_uchar_gauss(const uint8_t *src, int src_cols, int src_rows, uint8_t *dst);
void orc_vertices_to_depth(const float *vmap_rgba, int cols, int rows, float cutoff, float *dst);
void orc_image_to_intensity(const uint8_t *img, int channels, int cols, int rows, uint8_t *dst);
void orc_derivative_images(const uint8_t *src, int cols, int rows, int16_t *dx, int16_t *dy);
void orc_project_to_cloud(const float *depth, int cols, int rows, float fx, float fy, float cx,
                          float cy, float *cloud_xyz);

/* ---- reductions (reduce.cu) ----------------------------------------------------------- */
/* out29: 27 upper-triangular products of the 6x7 system + residual + inliers, in double.
 * err_map (may be NULL): what the reference writes to the ICP error surface.            */
void orc_icp_step(const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                  const float *nmap_curr, const float Rprev_inv[9], const float tprev[3], float fx,
                  float fy, float cx, float cy, const float *vmap_g_prev, const float *nmap_g_prev,
                  float dist_thres, float angle_thres, int cols, int rows, double out29[29],
                  float *err_map);
/* float-accumulating OpenMP variant: the "naive OpenMP CPU run" BASELINE.md section 3 asks for */
void orc_icp_step_omp_f32(const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                          const float *nmap_curr, const float Rprev_inv[9], const float tprev[3],
                          float fx, float fy, float cx, float cy, const float *vmap_g_prev,
                          const float *nmap_g_prev, float dist_thres, float angle_thres, int cols,
                          int rows, float out29[29]);
int orc_omp_threads(void);

void orc_rgb_residual(float min_scale, const int16_t *dIdx, const int16_t *dIdy,
                      const float *last_depth, const float *next_depth, const uint8_t *last_image,
                      const uint8_t *next_image, orc_dataterm *corres, float max_depth_delta,
                      const float kt[3], const float krkinv[9], int cols, int rows, int *sigma_sum,
                      int *count, float *err_map);
void orc_rgb_step(const orc_dataterm *corres, float sigma, const float *cloud_xyz, float fx, float fy,
                  const int16_t *dIdx, const int16_t *dIdy, float sobel_scale, int cols, int rows,
                  double out29[29]);
void orc_so3_step(const uint8_t *last_image, const uint8_t *next_image, const float image_basis[9],
                  const float kinv[9], const float krlr[9], int cols, int rows, double out11[11]);

/* unpack the 29 / 11 sums the way the reference host code does (reduce.cu:458-472, 1135-1149) */
void orc_unpack_se3(const double out29[29], float A[36], float b[6], float residual[2]);
void orc_unpack_so3(const double out11[11], float A[9], float b[3], float residual[2]);

/* ---- host algebra (RGBDOdometry.cpp / OdometryProvider.h) ------------------------------ */
void orc_rodrigues(const double r[3], double R[9]);
int orc_ldlt_solve(int n, const double *A, const double *b, double *x);
void orc_inverse3f(const float m[9], float inv[9]);
void orc_inverse4d(const double m[16], double inv[16]);

/* ---- whole odometry object (RGBDOdometry.{h,cpp}) ------------------------------------- */
typedef struct orc_odometry orc_odometry;

orc_odometry *orc_odom_create(int width, int height, float cx, float cy, float fx, float fy,
                              float dist_thresh, float angle_thresh);
void orc_odom_destroy(orc_odometry *o);
/* frame side: depth pyramid -> vmaps_curr/nmaps_curr (RGBDOdometry.cpp:110-118) */
void orc_odom_init_icp(orc_odometry *o, const float *depth_l0, float depth_cutoff);
/* model side: predicted vertex/normal RGBA32F images (RGBDOdometry.cpp:143-175) */
void orc_odom_init_icp_model(orc_odometry *o, const float *vert_rgba, const float *norm_rgba,
                             const float pose[16]);
/* model-to-model variant (RGBDOdometry.cpp:120-141) */
void orc_odom_init_icp_from_prediction(orc_odometry *o, const float *vert_rgba,
                                       const float *norm_rgba);
void orc_odom_init_rgb(orc_odometry *o, const uint8_t *rgb, int channels);
void orc_odom_init_rgb_model(orc_odometry *o, const uint8_t *rgb, int channels);
void orc_odom_init_first_rgb(orc_odometry *o, const uint8_t *rgb, int channels);
/* trans[3], rot[9] (row major) in/out.  icp_err / rgb_err: width*height floats or NULL */
void orc_odom_get_incremental_transformation(orc_odometry *o, float trans[3], float rot[9],
                                             int rgb_only, float icp_weight, int pyramid,
                                             int fast_odom, int so3, float *icp_err,
                                             float *rgb_err);
/* public result members of the reference class (RGBDOdometry.h:62-69) */
typedef struct {
    float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
    double lastA[36];
    double lastb[6];
    int iterations_run; /* diagnostic: GN iterations actually executed */
    int so3_iterations_run;
} orc_odom_stats;
void orc_odom_get_stats(const orc_odometry *o, orc_odom_stats *s);
/* internal buffers for kernel-level comparison; level 0..2; returns pointer into the object */
const float *orc_odom_buffer_f32(const orc_odometry *o, const char *name, int level);
const uint8_t *orc_odom_buffer_u8(const orc_odometry *o, const char *name, int level);
const int16_t *orc_odom_buffer_i16(const orc_odometry *o, const char *name, int level);

/* ---- surfel path (mmf_oracle_surfel.c): index map, splat, fuse, clean, init, filter ------ */
/* Vertex::SIZE = 48 bytes (Core/Shaders/Vertex.cpp:21-43, Model.h:247-264):
 * pos = xyz + confidence; col = {colour24-as-float, unused, initTime, timestamp}; nrm = xyz + radius */
typedef struct {
    float pos[4], col[4], nrm[4];
} orc_surfel;

void orc_inverse4f(const float m[16], float inv[16]);
/* host pose algebra of the surfel passes (mmf_oracle_pose.c; Model.cpp:876-891, 1301-1342, Eigen JacobiSVD) */
void orc_jacobi_svd3f(const float a[9], float U[9], float sv[3], float V[9]);
void orc_rodrigues2(const float matrix[9], float out[3]);
void orc_matmul4f(const float a[16], const float b[16], float out[16]);
float orc_compute_fusion_weight(const float pose[16], const float last_pose[16], float weight_multiplier);
void orc_bilateral_filter(const float *depth, int cols, int rows, float maxD, float *out);
int orc_surfel_initialise(const uint8_t *rgb, const float *depth_raw, const float *depth_filtered, int cols,
                          int rows, float cx, float cy, float fx, float fy, int time, float maxDepth,
                          orc_surfel *out);
void orc_predict_indices(const orc_surfel *s, int count, const float pose[16], float cx, float cy, float fx,
                         float fy, int cols, int rows, float maxDepth, int time, int timeDelta,
                         uint32_t *index, float *vertConf, float *colorTime, float *normRad);
void orc_combined_predict(const orc_surfel *s, int count, const float pose[16], float cx, float cy, float fx,
                          float fy, int cols, int rows, float maxDepth, float confThreshold, int time,
                          int maxTime, int timeDelta, uint8_t *image_rgba, float *vertexConf,
                          float *normalRadius, uint16_t *time_out);
void orc_synthesize_depth(const orc_surfel *s, int count, const float pose[16], float cx, float cy, float fx,
                          float fy, int cols, int rows, float maxDepth, float confThreshold, int time,
                          int maxTime, int timeDelta, float *depth_out);
/* keypoint descriptor matcher (mmf_oracle_match.c): cv::BFMatcher(NORM_L2, crossCheck).match + distance gate */
int orc_match_descriptors(const float *query, int nq, const float *train, int nt, int dim, float max_distance,
                          int *train_idx, float *distance);
/* SuperPoint network + keypoint post-processing (mmf_oracle_superpoint.c; channels-last activations) */
void orc_sp_conv(const float *in, int H, int W, int in_stride, int cin, const float *w, const float *bias, int cout, int taps,
                 int relu, float *out);
void orc_sp_maxpool2(const float *in, int H, int W, int C, float *out);
void orc_sp_input(const uint8_t *img, int H, int W, int channels, float *out);
void orc_sp_l2_normalize(float *desc, int npix, int C);
int orc_sp_forward(const float *input, int H, int W, const float *const *weights, float *semi, float *desc);
void orc_sp_heatmap(const float *semi, int Hc, int Wc, float *heat);
int orc_sp_keypoints(const float *heat, int H, int W, float conf_thresh, int nms_dist, int border, int max_out, int *xy,
                     float *conf);
void orc_sp_sample_descriptors(const float *desc, int Hc, int Wc, const int *xy, int n, int H, int W, float *out);
/* super-pixel resampling for the segmentation (mmf_oracle_slic.c; Slic.h:48-146, Slic.cpp:72-112) */
void orc_slic_counts(const int *labels, int npix, int nspix, int *counts);
void orc_slic_downsample(const int *labels, int width, int height, int S, const float *image, int channels, int channel,
                         int thresholded, float min_threshold, float *out);
void orc_slic_downsample_rgb(const int *labels, int width, int height, int S, const uint8_t *rgb, int channels, uint8_t *out);
void orc_slic_upsample_u8(const int *labels, int npix, const uint8_t *map, uint8_t *out);
int orc_fuse(orc_surfel *s, int count, const uint8_t *rgb, const float *depth_raw, const float *depth_filtered,
             const uint8_t *mask, const uint32_t *index, const float *vertConf, const float *normRad,
             const float pose[16], float cx, float cy, float fx, float fy, int cols, int rows, int time,
             float weighting, uint8_t maskID, float maxDepth, orc_surfel *new_out);
int orc_clean(const orc_surfel *s, int count, const orc_surfel *new_unstable, int nnew, const float pose[16],
              float cx, float cy, float fx, float fy, int cols, int rows, int time, int timeDelta,
              float confThreshold, float outlierCoeff, uint8_t maskID, const uint32_t *index,
              const float *vertConf, const float *colorTime, const float *depth_filtered, const uint8_t *mask,
              orc_surfel *out);
void orc_fill_in(const float *vertex_pred, const float *normal_pred, const uint8_t *image_pred_rgba,
                 const float *depth_filtered, const uint8_t *rgb, int cols, int rows, float cx, float cy,
                 float fx, float fy, int passthrough_geom, int passthrough_rgb, float *vertex_out,
                 float *normal_out, uint8_t *image_out_rgba);
int orc_requires_fill_in(const uint8_t *image_pred_rgba, int cols, int rows, float ratio);

#ifdef __cplusplus
}
#endif
#endif /* MMF_ORACLE_H_ */
