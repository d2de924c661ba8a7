// prep_batch.hpp -- the per-frame preparation of the dense tracker (vertex / normal maps, depth and
// intensity pyramids, gradients, point clouds, the model maps in the global frame) as FOUR launches.
//
// Each of those ~40 per-frame jobs (cudafuncs.cu:109-762 called from RGBDOdometry.cpp:108-235 and
// Model.cpp:359-407) is a one-touch kernel over at most 640x480 pixels: 3-5 us each, almost all of
// it the dependent-launch floor of the stream (~2.6 us).  They form a dependency chain only four deep
// (inputs -> level 0 -> level 1 -> level 2), so every stage of the chain runs as ONE launch whose
// workgroups pick their job from a table by block index.  The jobs call the same per-pixel functions
// (map_kernels.hpp) as the stand-alone kernels, so the results are bit-identical.
//
// Round 3: the chains got shorter by computing a stage's input in the job that needs it instead of storing it first -- the
// model side (prediction -> pyramids, global-frame maps, point clouds) is TWO dependent launches (PREP_TEX_*, PREP_RESIZE_TP,
// PREP_PYR_PROJECT), the depth side of the sensor frame THREE (PREP_VMAP_NMAP), its image side four; a launch on the
// model's stream costs ~4.5 us before it does anything, which is most of what a small stage takes.
#pragma once
#include "extent.hpp"
#include "icp_kernels.hpp"
#include "map_kernels.hpp"

namespace mmf {

enum PrepOp : int {
    PREP_VMAP,            // src0 depth -> dst0 vmap                          f = {1/fx, 1/fy, cx, cy, cutoff}
    PREP_NMAP,            // src0 vmap -> dst0 nmap
    PREP_TRANSFORM_PACK,  // src0 vmap, src1 nmap -> dst0 vmap, dst1 nmap, dst2 packed records   f = {R[9], t[3]}
    PREP_COPY_MAPS,       // src0, src1 RGBA32F prediction -> dst0, dst1 planar
    PREP_RESIZE_V,        // src0 (scols x srows) -> dst0
    PREP_RESIZE_N,
    PREP_PYRDOWN_F,       // src0 (scols x srows) -> dst0
    PREP_PYRDOWN_U8,
    PREP_V2D,             // src0 RGBA32F vertices -> dst0 depth                f = {cutoff}
    PREP_INTENSITY,       // src0 interleaved u8 (stride scols bytes, `channels`) -> dst0
    PREP_DERIV,           // src0 u8 -> dst0 dIdx, dst1 dIdy
    PREP_PROJECT,         // src0 depth -> dst0 AoS cloud, dst1 {X, Y, Z, 1/Z} records   f = {1/fx, 1/fy, cx, cy}
    // The coarsest level's products straight from the level above (the pyramid step's value stays in a register), so that the
    // chain of stages ends one launch earlier:
    PREP_RESIZE_TP,       // PREP_RESIZE_V + PREP_RESIZE_N + PREP_TRANSFORM_PACK: src0 vmap, src1 nmap (scols x srows) -> dst0, dst1, dst2   f = {R[9], t[3]}
    PREP_PYR_PROJECT,     // PREP_PYRDOWN_F + PREP_PROJECT: src0 depth (scols x srows) -> dst2 depth, dst0 cloud, dst1 records   f = {1/fx, 1/fy, cx, cy}
    // Level 0 and the first pyramid step straight from the prediction's RGBA32F / RGBA8 images (what PREP_V2D, PREP_INTENSITY
    // and PREP_COPY_MAPS would have written to memory first stays in registers), so that the chain of stages STARTS one launch
    // later: the model side is two dependent launches.  Same per-pixel functions on the same values: same bits.
    PREP_TEX_TP,          // PREP_COPY_MAPS + PREP_TRANSFORM_PACK: src0, src1 prediction -> dst0, dst1, dst2   f = {R[9], t[3]}
    PREP_TEX_PROJECT,     // PREP_V2D + PREP_PROJECT: src0 vertices -> dst2 depth, dst0 cloud, dst1 records   f = {1/fx, 1/fy, cx, cy, cutoff}
    PREP_TEX_PYR_F,       // PREP_V2D + PREP_PYRDOWN_F: src0 vertices (scols x srows) -> dst0 depth   f = {cutoff}
    PREP_TEX_PYR_U8,      // PREP_INTENSITY + PREP_PYRDOWN_U8: src0 image (scols x srows pixels of `channels` bytes) -> dst0
    PREP_VMAP_NMAP,       // PREP_VMAP + PREP_NMAP of the same level: src0 depth -> dst0 vmap, dst1 nmap   f = {1/fx, 1/fy, cx, cy, cutoff}
    PREP_TEX_RESIZE,      // PREP_COPY_MAPS + PREP_RESIZE_V + PREP_RESIZE_N: src0, src1 prediction (scols x srows) -> dst0 vmap, dst1 nmap
};

struct PrepJob {
    int op;
    int gx;           // 64-pixel tiles per row of this job's grid
    int reps;         // 64 x 4 tiles a workgroup handles, one below the other (a full-size job as 1 200 workgroups of one
                      // tile each is bound by the dispatcher: ~3 ns per workgroup, longer than the work)
    int first_block;  // first workgroup of the launch that belongs to this job
    int cols, rows;   // destination size in pixels (all internal buffers are dense: stride = cols)
    int scols, srows; // source size where it differs (pyramid steps), or the source byte stride (PREP_INTENSITY)
    int channels;
    const void *src0, *src1;
    void *dst0, *dst1, *dst2;
    // optional device-side choice of the sources: when sel != nullptr and *sel != 0 the job reads
    // alt0 / alt1 instead of src0 / src1 (the fill-in decision, Model.cpp:380, taken on the device)
    const int* sel;
    const void *alt0, *alt1;
    // sel_total != 0: *sel is a COUNT of covered thumbnail samples out of sel_total, and the alt_* sources are read when
    // count / total < sel_ratio (requiresFillIn, MultiMotionFusion.cpp:877-895: surfel_kernels.hpp: thumbnail_count_px)
    int sel_total;
    float sel_ratio;
    float f[12];
    // non-null: the job notes a bounding box of the valid depths it writes (extent.hpp: extent_of_level) -- all of them at the
    // coarsest level of an object model's depth pyramid, those in the last column / row at the two finer ones
    unsigned long long* ext;
    unsigned ext_gen;
    // non-null (PREP_TEX_TP of an object model): the job notes the pixel box and the depth range of the valid vertices it
    // reads into these extent words (extent.hpp: aabb_note), with ext_gen
    unsigned long long* aabb;
    // non-null (the sensor side's level-0 PREP_VMAP / PREP_VMAP_NMAP): the job notes the smallest valid depth it reads into the
    // extent words (extent.hpp: sensor_zmin_note) with zmin_gen
    unsigned long long* zmin;
    unsigned zmin_gen;
    // rect_now != nullptr (the model-side jobs of an OBJECT model): the job covers only the tiles that can differ from what
    // its outputs already hold -- the hull of the box the prediction's images are non-zero in NOW (rect_now: level-0 pixels
    // {x0, y0, x1, y1}, PassBoxes::spl_nz) and of that box at the model's previous preparation (rect_prev; null: unknown, the
    // whole image), taken to the job's level with the pyramid windows' reach -- on rect_groups workgroups that stride over
    // them.  Outside the hull the inputs are empty now and were empty then: the outputs there are the "nothing here" values
    // already.  rect_store: the job's first workgroup keeps rect_now there for the next preparation.
    const int *rect_now, *rect_prev;
    int* rect_store;
    int rect_level, rect_groups;
};

constexpr int kMaxPrepJobs = 24;  // 24 x 232 B of kernel arguments: what one model's stage (and a sensor side) needs
// Several models' jobs of a stage share a launch (eight models: 24 + 64 jobs in two stages): the wide table carries a stage
// of eight models in ONE launch -- as three launches of 24 the second stage was three launch latencies for jobs that do not
// depend on each other (8 models: 31 + 22 + 14 + 11 us on the stream the next chain waits for).  Longer lists go out in chunks.
constexpr int kMaxPrepJobsWide = 72;
template <int CAP>
struct PrepBatchT {
    int njobs;
    int critical;  // 1: on the model's stream (what a frame waits for): its waves ask for issue priority over side-stream work
    PrepJob job[CAP];
};
using PrepBatch = PrepBatchT<kMaxPrepJobs>;
using PrepBatchWide = PrepBatchT<kMaxPrepJobsWide>;

// transform_maps_px + pack_prev_kernel in one pass (same arithmetic; an invalid pixel's record is all NaN)
// (v_ok / n_ok: the source vertex / normal is valid, i.e. its x is not NaN)
// returns the global-frame vertex (NaN when the source is invalid)
__device__ __forceinline__ f3 transform_pack_store(int x, int y, int rows, int cols, bool v_ok, f3 vs, bool n_ok, f3 ns, m33 R, f3 t,
                                                   float* __restrict__ vdst, float* __restrict__ ndst, float* __restrict__ packed) {
    // (vdst / ndst: the planar copies, or null -- the chains gather from the packed records only)
    f3 vd = make_f3(qnan(), qnan(), qnan());
    if (v_ok) {
        vd = R * vs + t;
        if (vdst) {
            vdst[(size_t)(y + rows) * cols + x] = vd.y;
            vdst[(size_t)(y + 2 * rows) * cols + x] = vd.z;
        }
    }
    if (vdst) vdst[(size_t)y * cols + x] = vd.x;
    f3 nd = make_f3(qnan(), qnan(), qnan());
    if (n_ok) {
        nd = R * ns;
        if (ndst) {
            ndst[(size_t)(y + rows) * cols + x] = nd.y;
            ndst[(size_t)(y + 2 * rows) * cols + x] = nd.z;
        }
    }
    if (ndst) ndst[(size_t)y * cols + x] = nd.x;
    float2* o = reinterpret_cast<float2*>(packed + 6 * ((size_t)y * cols + x));
    o[0] = make_float2(vd.x, vd.y);
    o[1] = make_float2(vd.z, nd.x);
    o[2] = make_float2(nd.y, nd.z);
    return vd;
}
// what the threads of a job gather for a note its workgroup leaves behind (PrepJob::aabb)
struct PrepNote {
    float lo[3], hi[3];
};
__device__ __forceinline__ bool transform_pack_px(int x, int y, int rows, int cols, const float* __restrict__ vsrc,
                                                  const float* __restrict__ nsrc, m33 R, f3 t, float* __restrict__ vdst,
                                                  float* __restrict__ ndst, float* __restrict__ packed) {
    if (x >= cols || y >= rows) return false;
    f3 vs = make_f3(0.f, 0.f, 0.f), ns = vs;
    vs.x = vsrc[(size_t)y * cols + x];
    const bool v_ok = !(vs.x != vs.x);
    if (v_ok) {
        vs.y = vsrc[(size_t)(y + rows) * cols + x];
        vs.z = vsrc[(size_t)(y + 2 * rows) * cols + x];
    }
    ns.x = nsrc[(size_t)y * cols + x];
    const bool n_ok = !(ns.x != ns.x);
    if (n_ok) {
        ns.y = nsrc[(size_t)(y + rows) * cols + x];
        ns.z = nsrc[(size_t)(y + 2 * rows) * cols + x];
    }
    transform_pack_store(x, y, rows, cols, v_ok, vs, n_ok, ns, R, t, vdst, ndst, packed);
    return v_ok;  // (the vertex is valid)
}

// pixel (x, y) of job J
__device__ __forceinline__ void prep_job_px(const PrepJob& J, const void* src0, const void* src1, int x, int y, PrepNote& note) {
    const int cols = J.cols, rows = J.rows;
    switch (J.op) {
        case PREP_VMAP:
            create_vmap_px(x, y, (const float*)src0, cols, cols, rows, (float*)J.dst0, cols, J.f[0], J.f[1], J.f[2], J.f[3],
                           J.f[4]);
            if (J.zmin != nullptr && x < cols && y < rows) {  // (uniform test; the depth is in the cache)
                const float z = ((const float*)src0)[(size_t)y * cols + x];
                if (z != 0 && z < J.f[4]) note.lo[2] = fminf(note.lo[2], z);
            }
            break;
        case PREP_VMAP_NMAP:
            create_vmap_nmap_px(x, y, (const float*)src0, cols, rows, (float*)J.dst0, (float*)J.dst1, J.f[0], J.f[1], J.f[2], J.f[3],
                                J.f[4]);
            if (J.zmin != nullptr && x < cols && y < rows) {
                const float z = ((const float*)src0)[(size_t)y * cols + x];
                if (z != 0 && z < J.f[4]) note.lo[2] = fminf(note.lo[2], z);
            }
            break;
        case PREP_NMAP: create_nmap_px(x, y, rows, cols, (const float*)src0, cols, (float*)J.dst0, cols); break;
        case PREP_TRANSFORM_PACK: {
            m33 R;
#pragma unroll
            for (int k = 0; k < 9; ++k) R.m[k] = J.f[k];
            transform_pack_px(x, y, rows, cols, (const float*)src0, (const float*)src1, R, make_f3(J.f[9], J.f[10], J.f[11]),
                              (float*)J.dst0, (float*)J.dst1, (float*)J.dst2);
            break;
        }
        case PREP_COPY_MAPS:
            copy_maps_px(x, y, rows, cols, (const float4*)src0, (const float4*)src1, (float*)J.dst0, (float*)J.dst1, cols);
            break;
        case PREP_RESIZE_V: resize_map_px<false>(x, y, rows, cols, J.srows, (const float*)src0, J.scols, (float*)J.dst0, cols); break;
        case PREP_RESIZE_N: resize_map_px<true>(x, y, rows, cols, J.srows, (const float*)src0, J.scols, (float*)J.dst0, cols); break;
        case PREP_PYRDOWN_F:
            pyrdown_gauss_f_px(x, y, (const float*)src0, J.scols, J.scols, J.srows, (float*)J.dst0, cols, cols, rows);
            break;
        case PREP_PYRDOWN_U8:
            pyrdown_uchar_gauss_px(x, y, (const uint8_t*)src0, J.scols, J.scols, J.srows, (uint8_t*)J.dst0, cols, cols, rows);
            break;
        case PREP_V2D: vertices_to_depth_px(x, y, (const float4*)src0, cols, rows, (float*)J.dst0, cols, J.f[0]); break;
        case PREP_INTENSITY:
            image_to_intensity_px(x, y, (const uint8_t*)src0, J.scols, J.channels, cols, rows, (uint8_t*)J.dst0, cols);
            break;
        case PREP_DERIV:
            derivative_px(x, y, (const uint8_t*)src0, cols, cols, rows, (int16_t*)J.dst0, cols, (int16_t*)J.dst1, cols);
            break;
        case PREP_PROJECT:
            project_points_px(x, y, (const float*)src0, cols, cols, rows, (float*)J.dst0, J.f[0], J.f[1], J.f[2], J.f[3], (float4*)J.dst1);
            break;
        case PREP_RESIZE_TP: {
            if (x >= cols || y >= rows) break;
            m33 R;
#pragma unroll
            for (int k = 0; k < 9; ++k) R.m[k] = J.f[k];
            f3 vs = make_f3(0.f, 0.f, 0.f), ns = vs;
            const bool v_ok = resize_map_value<false>(x, y, J.srows, (const float*)src0, J.scols, vs);
            const bool n_ok = resize_map_value<true>(x, y, J.srows, (const float*)src1, J.scols, ns);
            transform_pack_store(x, y, rows, cols, v_ok, vs, n_ok, ns, R, make_f3(J.f[9], J.f[10], J.f[11]), (float*)J.dst0,
                                 (float*)J.dst1, (float*)J.dst2);
            break;
        }
        case PREP_TEX_TP: {
            if (x >= cols || y >= rows) break;
            m33 R;
#pragma unroll
            for (int k = 0; k < 9; ++k) R.m[k] = J.f[k];
            const float4 v = ((const float4*)src0)[(size_t)y * cols + x], n = ((const float4*)src1)[(size_t)y * cols + x];
            const bool ok = !(v.z == 0);  // copy_maps_px: an empty texel is an invalid vertex AND an invalid normal
            const f3 vs = ok ? make_f3(v.x, v.y, v.z) : make_f3(qnan(), qnan(), qnan());
            const f3 ns = ok ? make_f3(n.x, n.y, n.z) : make_f3(qnan(), qnan(), qnan());
            transform_pack_store(x, y, rows, cols, !(vs.x != vs.x), vs, !(ns.x != ns.x), ns, R, make_f3(J.f[9], J.f[10], J.f[11]),
                                 (float*)J.dst0, (float*)J.dst1, (float*)J.dst2);
            if (ok) {  // (extent.hpp: the pixel box and the camera-frame depth range of the valid vertices)
                note.lo[0] = fminf(note.lo[0], (float)x), note.lo[1] = fminf(note.lo[1], (float)y), note.lo[2] = fminf(note.lo[2], v.z);
                note.hi[0] = fmaxf(note.hi[0], (float)x), note.hi[1] = fmaxf(note.hi[1], (float)y), note.hi[2] = fmaxf(note.hi[2], v.z);
            }
            break;
        }
        case PREP_TEX_PROJECT: {
            if (x >= cols || y >= rows) break;
            const float z = vertex_depth_value(((const float4*)src0)[(size_t)y * cols + x].z, J.f[4]);
            ((float*)J.dst2)[(size_t)y * cols + x] = z;
            project_points_store(x, y, z, cols, (float*)J.dst0, J.f[0], J.f[1], J.f[2], J.f[3], (float4*)J.dst1);
            extent_note(J.ext, J.ext_gen, x, y, !(z != z) && (x == cols - 1 || y == rows - 1));  // (extent_of_level: level 0's last column / row)
            break;
        }
        case PREP_TEX_PYR_F: {
            if (x >= cols || y >= rows) break;
            const float4* tex = (const float4*)src0;
            const int scols = J.scols;
            const float cutoff = J.f[0];
            const float z = pyrdown_gauss_f_taps(
                x, y, scols, J.srows, [&](int yy, int xx) { return vertex_depth_value(tex[(size_t)yy * scols + xx].z, cutoff); });
            ((float*)J.dst0)[(size_t)y * cols + x] = z;
            extent_note(J.ext, J.ext_gen, x, y, !(z != z) && (x == cols - 1 || y == rows - 1));  // (level 1's last column / row)
            break;
        }
        case PREP_TEX_PYR_U8: {
            if (x >= cols || y >= rows) break;
            const uint8_t* img = (const uint8_t*)src0;
            const int ch = J.channels;
            const size_t stride = (size_t)J.scols * ch;
            if (ch == 4) {  // an RGBA8 prediction image: one load per tap
                const unsigned* tex = (const unsigned*)img;
                const int scols = J.scols;
                ((uint8_t*)J.dst0)[(size_t)y * cols + x] = pyrdown_uchar_gauss_taps(
                    x, y, scols, J.srows, [&](int yy, int xx) { return intensity_value_rgba(tex[(size_t)yy * scols + xx]); });
            } else {
                ((uint8_t*)J.dst0)[(size_t)y * cols + x] = pyrdown_uchar_gauss_taps(
                    x, y, J.scols, J.srows, [&](int yy, int xx) { return intensity_value(img + (size_t)yy * stride + (size_t)xx * ch); });
            }
            break;
        }
        case PREP_TEX_RESIZE: {
            if (x >= cols || y >= rows) break;
            const float4* vt = (const float4*)src0;
            const float4* nt = (const float4*)src1;
            const int scols = J.scols;
            float4 v[4], n[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t at = (size_t)(2 * y + (k >> 1)) * scols + 2 * x + (k & 1);
                v[k] = vt[at], n[k] = nt[at];
            }
            float vx[4], vy[4], vz[4], nx[4], ny[4], nz[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // copy_maps_px
                const bool ok = !(v[k].z == 0);
                vx[k] = ok ? v[k].x : qnan(), vy[k] = ok ? v[k].y : qnan(), vz[k] = ok ? v[k].z : qnan();
                nx[k] = ok ? n[k].x : qnan(), ny[k] = ok ? n[k].y : qnan(), nz[k] = ok ? n[k].z : qnan();
            }
            float* vd = (float*)J.dst0;
            float* nd = (float*)J.dst1;
            // resize_map_px<false> / <true>: x00 + x01 + x10 + x11 in that order
            if ((vx[0] != vx[0]) || (vx[1] != vx[1]) || (vx[2] != vx[2]) || (vx[3] != vx[3])) {
                vd[(size_t)y * cols + x] = qnan();
            } else {
                vd[(size_t)y * cols + x] = (vx[0] + vx[1] + vx[2] + vx[3]) / 4;
                vd[(size_t)(y + rows) * cols + x] = (vy[0] + vy[1] + vy[2] + vy[3]) / 4;
                vd[(size_t)(y + 2 * rows) * cols + x] = (vz[0] + vz[1] + vz[2] + vz[3]) / 4;
            }
            if ((nx[0] != nx[0]) || (nx[1] != nx[1]) || (nx[2] != nx[2]) || (nx[3] != nx[3])) {
                nd[(size_t)y * cols + x] = qnan();
            } else {
                f3 m;
                m.x = (nx[0] + nx[1] + nx[2] + nx[3]) / 4;
                m.y = (ny[0] + ny[1] + ny[2] + ny[3]) / 4;
                m.z = (nz[0] + nz[1] + nz[2] + nz[3]) / 4;
                m = normalized(m);
                nd[(size_t)y * cols + x] = m.x;
                nd[(size_t)(y + rows) * cols + x] = m.y;
                nd[(size_t)(y + 2 * rows) * cols + x] = m.z;
            }
            break;
        }
        case PREP_PYR_PROJECT: {
            if (x >= cols || y >= rows) break;
            const float z = pyrdown_gauss_f_value(x, y, (const float*)src0, J.scols, J.scols, J.srows);
            ((float*)J.dst2)[(size_t)y * cols + x] = z;
            project_points_store(x, y, z, cols, (float*)J.dst0, J.f[0], J.f[1], J.f[2], J.f[3], (float4*)J.dst1);
            extent_note(J.ext, J.ext_gen, x, y, !(z != z));
            break;
        }
        default: break;
    }
}

// workgroup `block` of the launch (64 x 4 threads)
template <int CAP>
__device__ __forceinline__ void prep_batch_body(const PrepBatchT<CAP>& b, int block) {
    if (b.critical) __builtin_amdgcn_s_setprio(3);
    int j = 0;
    for (int k = 1; k < b.njobs; ++k) j = block >= b.job[k].first_block ? k : j;  // wave uniform
    const PrepJob& J = b.job[j];
    const int local = block - J.first_block;
    const bool alt = J.sel != nullptr && (J.sel_total ? ((float)*J.sel / (float)J.sel_total < J.sel_ratio) : *J.sel != 0);  // wave uniform
    const void* src0 = alt ? J.alt0 : J.src0;
    const void* src1 = alt ? J.alt1 : J.src1;
    PrepNote note;
#pragma unroll
    for (int k = 0; k < 3; ++k) note.lo[k] = FLT_MAX, note.hi[k] = -FLT_MAX;
    if (J.rect_now != nullptr) {  // (uniform) an object model's job: the tiles of the hull only
        const ExtentBox now = box_of_ints(J.rect_now);
        ExtentBox hull = J.rect_prev ? extent_hull(now, box_of_ints(J.rect_prev)) : ExtentBox{0, 0, (J.cols << J.rect_level) - 1, (J.rows << J.rect_level) - 1};
        if (J.rect_store != nullptr && local == 0 && threadIdx.x == 0 && threadIdx.y == 0) box_to_ints(J.rect_store, now);
        if (!(hull.x1 < hull.x0 || hull.y1 < hull.y0)) {
            // a destination pixel of level l + 1 reads sources within +-2 of twice its coordinates (5 x 5 pyramid windows; 2 x 2 resize)
            for (int l = 0; l < J.rect_level; ++l) hull = ExtentBox{(hull.x0 - 4) >> 1, (hull.y0 - 4) >> 1, (hull.x1 + 4) >> 1, (hull.y1 + 4) >> 1};
            hull = box_clip(hull, J.cols, J.rows);
            const int th = kTileY * J.reps;
            const int tx0 = hull.x0 / kTileX, ty0 = hull.y0 / th, tw = hull.x1 / kTileX - tx0 + 1, ntiles = tw * (hull.y1 / th - ty0 + 1);
            for (int t = local; t < ntiles; t += J.rect_groups) {
                const int ty = t / tw, x = (tx0 + t - ty * tw) * kTileX + (int)threadIdx.x;
                for (int r = 0; r < J.reps; ++r) prep_job_px(J, src0, src1, x, ((ty0 + ty) * J.reps + r) * kTileY + (int)threadIdx.y, note);
            }
        }
    } else {
        const int by = local / J.gx, bx = local - by * J.gx;
        const int x = bx * kTileX + threadIdx.x;
        for (int r = 0; r < J.reps; ++r) prep_job_px(J, src0, src1, x, (by * J.reps + r) * kTileY + threadIdx.y, note);
    }
    if (J.zmin != nullptr) {  // (uniform) the workgroup's smallest valid depth -> the sensor slot of the extent words
        __shared__ float zlo[kTileY];
        float z = note.lo[2];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) z = fminf(z, __shfl_xor(z, d));
        if (threadIdx.x == 0) zlo[threadIdx.y] = z;
        __syncthreads();
        if (threadIdx.x == 0 && threadIdx.y == 0) {
            for (int w = 1; w < kTileY; ++w) z = fminf(z, zlo[w]);
            if (z < FLT_MAX) sensor_zmin_note(J.zmin, J.zmin_gen, z);
        }
    }
    if (J.aabb != nullptr) {  // (uniform) the workgroup's box of valid vertices -> the model's extent words
        __shared__ float box[kTileY][6];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                note.lo[k] = fminf(note.lo[k], __shfl_xor(note.lo[k], d));
                note.hi[k] = fmaxf(note.hi[k], __shfl_xor(note.hi[k], d));
            }
        if (threadIdx.x == 0)
            for (int k = 0; k < 3; ++k) box[threadIdx.y][k] = note.lo[k], box[threadIdx.y][3 + k] = note.hi[k];
        __syncthreads();
        if (threadIdx.x == 0 && threadIdx.y == 0) {
            for (int w = 1; w < kTileY; ++w)
                for (int k = 0; k < 3; ++k) note.lo[k] = fminf(note.lo[k], box[w][k]), note.hi[k] = fmaxf(note.hi[k], box[w][3 + k]);
            if (note.lo[0] <= note.hi[0]) aabb_note(J.aabb, J.ext_gen, note.lo, note.hi);
        }
    }
}
__global__ __launch_bounds__(256) void prep_batch_kernel(PrepBatch b) { prep_batch_body(b, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void prep_batch_wide_kernel(PrepBatchWide b) { prep_batch_body(b, (int)blockIdx.x); }

}  // namespace mmf
