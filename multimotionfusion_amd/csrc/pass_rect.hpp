// pass_rect.hpp -- the projection / fuse / clean / predict passes of OBJECT models restricted to where the model is.
//
// An object model is a few thousand surfels that cover a box of a hundred-odd pixels, yet every per-pixel pass of the
// reference covers the whole frame (a full-screen draw per pass and model, MultiMotionFusion.cpp:791-816, 863-875): the two
// index-map resolves write 52 B for each of 307 200 texels, the fuse pass tests every fourth pixel's mask, the clean pass
// walks count + 307 200 candidates, the prediction's resolve writes 38 B per pixel -- per model.  With seven object models
// on a GPU that is ~400 us of kernel time per frame for images that are zero almost everywhere.
//
// Here every pass that WRITES the key image (index_map, fuse_update_index, splat) notes the box of its writes, every resolve
// pass walks the hull of that box and of the box its images were last non-zero in (so what lies outside stays exactly what
// a full-frame pass would have written: zeros), and the two passes that walk the INPUT frame (fuse's data association, the
// new candidates of clean) walk the box of the model's id in the segmentation's id image.  Boxes live on the device
// (generation-tagged words raised by atomicMax, extent.hpp's encoding: nothing is ever reset, nothing is read back); a launch
// is a FIXED number of workgroups per model that stride over the box, and all object models of a frame share one launch per
// pass (gridDim.y = model).  The per-texel / per-pixel / per-candidate arithmetic is the very code of the full-frame kernels
// (surfel_kernels.hpp), so maps, images and their order keep their bits (tests/test_gpu_multimodel.py).
#pragma once
#include "extent.hpp"
#include "surfel_kernels.hpp"

namespace mmf {

constexpr int kRectGroups = 128;   // workgroups per model of a rect launch (an object's box of 200 x 200: one or two strides each)

// ---- the id image's boxes: one launch per frame for all ids (the segmentation's result is an input of processFrame) ----
// boxes[id][4] (id 0, the background, is not noted).  64 x 16 pixel tiles; per tile the ids present are few.
__global__ __launch_bounds__(256) void mask_boxes_kernel(const uint8_t* __restrict__ mask, int cols, int rows, unsigned long long* __restrict__ boxes,
                                                         unsigned gen) {
    __shared__ int lo_x[256], hi_x[256], lo_y[256], hi_y[256];
    const int t = threadIdx.x;
    lo_x[t] = 0x7FFF, hi_x[t] = -1, lo_y[t] = 0x7FFF, hi_y[t] = -1;
    __syncthreads();
    const int tiles_x = (cols + 63) / 64;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int x = bx * 64 + (t & 63);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = by * 16 + (t >> 6) * 4 + r;
        if (x < cols && y < rows) {
            const int id = mask[(size_t)y * cols + x];
            if (id != 0) {
                atomicMin(&lo_x[id], x), atomicMax(&hi_x[id], x);
                atomicMin(&lo_y[id], y), atomicMax(&hi_y[id], y);
            }
        }
    }
    __syncthreads();
    if (hi_x[t] >= lo_x[t]) {
        const unsigned long long g = (unsigned long long)gen << 32;
        unsigned long long* w = boxes + 4 * t;
        atomicMax(&w[0], g | (unsigned long long)(0xFFFF - lo_x[t]));
        atomicMax(&w[1], g | (unsigned long long)hi_x[t]);
        atomicMax(&w[2], g | (unsigned long long)(0xFFFF - lo_y[t]));
        atomicMax(&w[3], g | (unsigned long long)hi_y[t]);
    }
}

// ---- projections that note where they write ----
struct index_map_box_item {
    SurfelSoA s;
    int count;
    IndexArgs a;
    unsigned long long* keys;
    PassBoxes* boxes;
    unsigned kgen;
    unsigned grid;
};
__global__ __launch_bounds__(256) void index_map_box_batched_kernel(PassBatch<index_map_box_item> b) {
    const index_map_box_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    MMF_MODEL_STREAM_PRIORITY();
    const int id = (int)blockIdx.x * 256 + (int)threadIdx.x;
    int px = 0, py = 0;
    bool wrote = false;
    if (id < p.count) wrote = index_map_project_xy(p.a, id, p.s.pos[id], p.s.col[id].w, p.keys, px, py);
    box_note_wave(p.boxes->key[p.kgen & 1u], p.kgen, wrote, px, px, py, py);
}

struct fuse_update_index_box_item {
    SurfelSoA s;
    int count;
    SurfelSoA meas;
    int time;
    unsigned* winner;
    IndexArgs a;
    unsigned long long* keys;
    PassBoxes* boxes;
    unsigned kgen;
    unsigned grid;
};
__global__ __launch_bounds__(256) void fuse_update_index_box_batched_kernel(PassBatch<fuse_update_index_box_item> b) {
    const fuse_update_index_box_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    MMF_MODEL_STREAM_PRIORITY();
    const int k = (int)blockIdx.x * 256 + (int)threadIdx.x;
    int px = 0, py = 0;
    bool wrote = false;
    if (k < p.count) {
        const unsigned w = p.winner[k];
        float4 op, oc;
        if (w == kNoWinner) {
            op = p.s.pos[k];
            oc.w = p.s.col[k].w;
        } else {
            p.winner[k] = kNoWinner;
            fuse_update_one(p.s, k, w, p.meas, p.time, op, oc);
        }
        wrote = index_map_project_xy(p.a, k, op, oc.w, p.keys, px, py);
    }
    box_note_wave(p.boxes->key[p.kgen & 1u], p.kgen, wrote, px, px, py, py);
}

// ---- predictIndices' resolve over hull(what was written, what was non-zero) ----
struct index_resolve_rect_item {
    SurfelSoA s;
    IndexArgs a;
    unsigned long long* keys;
    unsigned* index;
    float4 *vertConf, *colorTime, *normRad;
    PassBoxes* boxes;
    unsigned kgen, igen;  // the projection launch whose keys are resolved; this resolve's number
    int prev_whole;       // != 0: the images may be non-zero anywhere (a full-frame pass wrote them last)
};
__global__ __launch_bounds__(256) void index_resolve_rect_batched_kernel(PassBatch<index_resolve_rect_item> b) {
    const index_resolve_rect_item& p = b.m[blockIdx.y];
    MMF_MODEL_STREAM_PRIORITY();
    const int cols = p.a.cols, rows = p.a.rows;
    const ExtentBox kb = box_clip(extent_load(p.boxes->key[p.kgen & 1u], p.kgen), cols, rows);
    const ExtentBox prev = p.prev_whole ? ExtentBox{0, 0, cols - 1, rows - 1} : box_of_ints(p.boxes->idx_nz[(p.igen + 1u) & 1u]);
    const ExtentBox r = extent_hull(kb, prev);
    if (blockIdx.x == 0 && threadIdx.x == 0) box_to_ints(p.boxes->idx_nz[p.igen & 1u], kb);
    if (r.x1 < r.x0 || r.y1 < r.y0) return;
    const int h = r.y1 - r.y0 + 1, total = (r.x1 - r.x0 + 1) * h;
    for (int t = (int)blockIdx.x * 256 + (int)threadIdx.x; t < total; t += (int)gridDim.x * 256) {
        const int cx = t / h;
        index_resolve_texel((r.x0 + cx) * rows + r.y0 + (t - cx * h), p.s, p.a, p.keys, p.index, p.vertConf, p.colorTime, p.normRad);
    }
}

// ---- fuse's data association over the box of the model's id ----
struct fuse_data_rect_item {
    const uint8_t* rgb;
    const float *depth_raw, *depth_fil;
    const uint8_t* mask;
    const unsigned* index;
    const float4 *vertConf, *normRad;
    FuseArgs a;
    SurfelSoA meas;
    unsigned *new_flags, *winner;
    const unsigned long long* mask_box;  // this id's four words
    unsigned mask_gen;
};
__global__ __launch_bounds__(256) void fuse_data_rect_batched_kernel(PassBatch<fuse_data_rect_item> b) {
    const fuse_data_rect_item& p = b.m[blockIdx.y];
    MMF_MODEL_STREAM_PRIORITY();
    const int cols = p.a.cols, rows = p.a.rows;
    const ExtentBox mb = box_clip(extent_load(p.mask_box, p.mask_gen), cols, rows);
    if (mb.x1 < mb.x0 || mb.y1 < mb.y0) return;
    // one thread per 2 x 2 block (fuse_data_kernel): the blocks that touch the box
    const int hr = (rows + 1) / 2;
    const int bx0 = mb.x0 / 2, by0 = mb.y0 / 2, bh = mb.y1 / 2 - by0 + 1, total = (mb.x1 / 2 - bx0 + 1) * bh;
    for (int t = (int)blockIdx.x * 256 + (int)threadIdx.x; t < total; t += (int)gridDim.x * 256) {
        const int c = t / bh;
        fuse_data_thread((bx0 + c) * hr + by0 + (t - c * bh), p.rgb, p.depth_raw, p.depth_fil, p.mask, p.index, p.vertConf, p.normRad, p.a, p.meas,
                         p.new_flags, p.winner);
    }
}

// ---- clean: the existing surfels, then the NEW candidates of the id's box in draw order (column-major), compacted ----
struct clean_rect_item {
    SurfelSoA s, meas;
    const unsigned* new_flags;
    CleanArgs a;
    const unsigned* index;
    const float4 *vertConf, *colorTime;
    const float* depth_in;
    const uint8_t* mask;
    unsigned* keep;
    float2* conf_time;
    unsigned* block_sums;
    SurfelSoA dst;
    int capacity;
    unsigned *total_out, *total_host;
    unsigned seq;
    const unsigned long long* mask_box;
    unsigned mask_gen;
};
// candidate c of the compacted list: surfel c, or the pixel of the box with column-major rank c - count; its index e in
// the full-frame numbering (clean_flag_one's: count + x * rows + y)
__device__ __forceinline__ int clean_rect_candidate(int c, int count, const ExtentBox& mb, int rows) {
    if (c < count) return c;
    const int h = mb.y1 - mb.y0 + 1, q = c - count, cx = q / h;
    return count + (mb.x0 + cx) * rows + mb.y0 + (q - cx * h);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MMF_CLEAN_WAVES))) void clean_flag_rect_batched_kernel(PassBatch<clean_rect_item> b) {
    const clean_rect_item& p = b.m[blockIdx.y];
    MMF_MODEL_STREAM_PRIORITY();
    const int cols = p.a.cols, rows = p.a.rows;
    const ExtentBox mb = box_clip(extent_load(p.mask_box, p.mask_gen), cols, rows);
    const int area = (mb.x1 < mb.x0 || mb.y1 < mb.y0) ? 0 : (mb.x1 - mb.x0 + 1) * (mb.y1 - mb.y0 + 1);
    const int n = p.a.count + area, nvb = (n + 255) / 256;
    for (int vb = (int)blockIdx.x; vb < nvb; vb += (int)gridDim.x) {  // (uniform)
        const int c = vb * 256 + (int)threadIdx.x;
        unsigned k = 0u;
        if (c < n) {
            k = clean_flag_one(clean_rect_candidate(c, p.a.count, mb, rows), p.s, p.meas, p.new_flags, p.a, p.index, p.vertConf, p.colorTime, p.depth_in,
                               p.mask, p.conf_time, c);
            p.keep[c] = k;
        }
        const int kept = __syncthreads_count((int)k);
        if (threadIdx.x == 0) p.block_sums[vb] = (unsigned)kept;
    }
}
__global__ __launch_bounds__(256) void clean_scatter_rect_batched_kernel(PassBatch<clean_rect_item> b) {
    const clean_rect_item& p = b.m[blockIdx.y];
    MMF_MODEL_STREAM_PRIORITY();
    __shared__ unsigned wave_part[4], wave_kept[4];
    const int cols = p.a.cols, rows = p.a.rows, count = p.a.count;
    const ExtentBox mb = box_clip(extent_load(p.mask_box, p.mask_gen), cols, rows);
    const int area = (mb.x1 < mb.x0 || mb.y1 < mb.y0) ? 0 : (mb.x1 - mb.x0 + 1) * (mb.y1 - mb.y0 + 1);
    const int n = count + area, nvb = (n + 255) / 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (nvb == 0 && blockIdx.x == 0 && threadIdx.x == 0) {  // nothing at all: the count is still published
        *p.total_out = 0u;
        if (p.total_host) {
            p.total_host[0] = 0u;
            __threadfence_system();
            __hip_atomic_store(&p.total_host[1], p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (int vb = (int)blockIdx.x; vb < nvb; vb += (int)gridDim.x) {  // (uniform)
        const int c = vb * 256 + (int)threadIdx.x;
        const bool live = c < n, old = c < count;
        const unsigned kp = live ? p.keep[c] : 0u;
        unsigned part = 0;
        for (int j = (int)threadIdx.x; j < vb; j += 256) part += p.block_sums[j];
        part = wave_sum_to_lane63(part);
        const unsigned long long ballot = __ballot(kp != 0u);
        __syncthreads();  // (the previous stride's readers of wave_part / wave_kept are done)
        if (lane == 63) wave_part[wave] = part;
        if (lane == 0) wave_kept[wave] = (unsigned)__popcll(ballot);
        __syncthreads();
        unsigned base = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3];
        for (int w = 0; w < wave; ++w) base += wave_kept[w];
        if (vb == nvb - 1 && threadIdx.x == 0) {
            const unsigned total = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3] + wave_kept[0] + wave_kept[1] + wave_kept[2] + wave_kept[3];
            *p.total_out = total;
            if (p.total_host) {
                p.total_host[0] = total;
                __threadfence_system();
                __hip_atomic_store(&p.total_host[1], p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (!kp) continue;
        const unsigned k = base + (unsigned)__popcll(ballot & ((1ull << lane) - 1ull));
        if (k >= (unsigned)p.capacity) continue;  // the reference's VBO is full: further primitives are dropped
        const float2 ct = p.conf_time[c];
        if (old) {
            const float4 sp = p.s.pos[c], sc = p.s.col[c];
            p.dst.pos[k] = make_float4(sp.x, sp.y, sp.z, ct.x);
            p.dst.col[k] = make_float4(sc.x, sc.y, sc.z, ct.y);
            p.dst.nrm[k] = p.s.nrm[c];
        } else {
            const int d = clean_rect_candidate(c, count, mb, rows) - count;
            const float4 mp = p.meas.pos[d], mc = p.meas.col[d];
            p.dst.pos[k] = make_float4(mp.x, mp.y, mp.z, ct.x);
            p.dst.col[k] = make_float4(mc.x, mc.y, mc.z, ct.y);
            p.dst.nrm[k] = p.meas.nrm[d];
        }
    }
}

// ---- combinedPredict: the rasterising pass notes its sprites' box, the resolve walks 16 x 16 tiles of the hull ----
struct splat_box_item {
    SurfelSoA s;
    int count;
    SplatArgs a;
    unsigned long long* keys;
    const unsigned* count_dev;
    PassBoxes* boxes;
    unsigned kgen;
    unsigned grid;
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void splat_box_batched_kernel(PassBatch<splat_box_item> b) {
    const splat_box_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    splat_kernel_body<false, true>(p.s, p.count, p.a, p.keys, p.count_dev, blockIdx.x, p.grid, p.boxes->key[p.kgen & 1u], p.kgen);
}

struct splat_resolve_rect_item {
    SurfelSoA s;
    SplatArgs a;
    unsigned long long* keys;
    uchar4* image;
    float4 *vertexConf, *normalRadius;
    unsigned short* time_out;
    unsigned* thumb;
    int gen;
    PassBoxes* boxes;
    unsigned kgen, sgen;
    int prev_whole;
};
__global__ __launch_bounds__(256) void splat_resolve_rect_batched_kernel(PassBatch<splat_resolve_rect_item> b) {
    const splat_resolve_rect_item& p = b.m[blockIdx.y];
    MMF_MODEL_STREAM_PRIORITY();
    const SplatArgs& a = p.a;
    const ExtentBox kb = box_clip(extent_load(p.boxes->key[p.kgen & 1u], p.kgen), a.cols, a.rows);
    const ExtentBox prev = p.prev_whole ? ExtentBox{0, 0, a.cols - 1, a.rows - 1} : box_of_ints(p.boxes->spl_nz[(p.sgen + 1u) & 1u]);
    const ExtentBox r = extent_hull(kb, prev);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        box_to_ints(p.boxes->spl_nz[p.sgen & 1u], kb);
        if (p.thumb) p.thumb[(p.gen + 1) & 1] = 0u;  // (thumbnail_count_px: the next generation's counter starts from zero)
    }
    if (r.x1 < r.x0 || r.y1 < r.y0) return;
    const int tiles_x = (a.cols + kSplatTile - 1) / kSplatTile;
    const int tx0 = r.x0 / kSplatTile, ty0 = r.y0 / kSplatTile, tw = r.x1 / kSplatTile - tx0 + 1, total = tw * (r.y1 / kSplatTile - ty0 + 1);
    for (int t = (int)blockIdx.x; t < total; t += (int)gridDim.x) {  // (uniform)
        const int ty = t / tw;
        int px, py;
        unsigned long long k;
        const bool in = splat_tile_key(p.keys, a.cols, a.rows, px, py, k, 0u, (unsigned)((ty0 + ty) * tiles_x + tx0 + (t - ty * tw)));
        if (in) {
            const int i = py * a.cols + px;
            const SplatTexel tex = splat_resolve_px(i, k, p.s, a);
            p.image[i] = tex.image;
            p.vertexConf[i] = tex.vertexConf, p.normalRadius[i] = tex.normalRadius;
            p.time_out[i] = tex.time;
            thumbnail_count_px(px, py, a.cols, a.rows, tex.image, p.thumb, p.gen, false);
        }
        __syncthreads();  // (splat_tile_key's LDS tile is rewritten by the next stride)
    }
}

}  // namespace mmf
