// track_kernels.hpp -- gfx950 kernels of the dense-tracking reductions:
//   icp_kernel          <- icpKernel + reduceSum            (Core/Cuda/reduce.cu:231-473)
//   rgb_residual_kernel <- residualKernel + reduceSum(int2) (Core/Cuda/reduce.cu:722-945)
//   rgb_step_kernel     <- rgbKernel + reduceSum            (Core/Cuda/reduce.cu:477-661)
//   so3_kernel          <- so3Kernel + reduceSum            (Core/Cuda/reduce.cu:947-1150)
// plus the single-lane bookkeeping kernels of the device-resident Gauss-Newton loop.
//
// All are HBM/latency bound (about 110 flop per 48 bytes for ICP), so the design goals are:
// coalesced 16-byte loads of the planar maps, all gathers of a pixel group issued together,
// one launch per reduction (grid_reduce.hpp) and no host synchronisation between iterations.
#pragma once
#include "device_math.hpp"
#include "grid_reduce.hpp"
#include "odom_state.hpp"

namespace mmf {

enum FinishMode { FINISH_RAW = 0, FINISH_GN = 1 };

// planar 3-plane map view: element (plane k, row y, col x) at base[(y + k*rows)*stride + x]
struct MapView {
    const float* base;
    int stride;  // in floats
};

// accumulate the 27 upper-triangular products of a 7-vector + residual^2 + inlier flag
// in the member order of JtJJtrSE3 (types.cuh:101-112, reduce.cu:331-365)
__device__ __forceinline__ void accumulate_se3(float (&sum)[29], const float (&row)[7], float found) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 7; ++j) {
            sum[k] = sum[k] + row[i] * row[j];
            ++k;
        }
    sum[27] = sum[27] + row[6] * row[6];
    sum[28] = sum[28] + found;
}

struct IcpArgs {
    MapView vmap_curr, nmap_curr, vmap_g_prev, nmap_g_prev;
    LevelIntr intr;
    float dist_thres, angle_thres;
    int cols, rows;
    float* err_map;  // optional
    int err_stride;
};

// One pixel of ICPReduction::search + getProducts (reduce.cu:257-368), given its current
// vertex / normal.  Branch-free up to the gather so a lane's four pixels issue their gathers
// back to back.
struct IcpPixel {
    f3 vcurr_g, vcurr_cp;
    int ux, uy;
    bool inside;
};

__device__ __forceinline__ IcpPixel icp_project(const OdomState* st, const IcpArgs& a, f3 vcurr) {
    const m33& Rcurr = *reinterpret_cast<const m33*>(st->Rcurr);
    const m33& Rprev_inv = *reinterpret_cast<const m33*>(st->Rprev_inv);
    const f3 tcurr = make_f3(st->tcurr[0], st->tcurr[1], st->tcurr[2]);
    const f3 tprev = make_f3(st->tprev[0], st->tprev[1], st->tprev[2]);
    IcpPixel p;
    p.vcurr_g = Rcurr * vcurr + tcurr;
    p.vcurr_cp = Rprev_inv * (p.vcurr_g - tprev);
    p.ux = float2int_rn(p.vcurr_cp.x * a.intr.fx / p.vcurr_cp.z + a.intr.cx);
    p.uy = float2int_rn(p.vcurr_cp.y * a.intr.fy / p.vcurr_cp.z + a.intr.cy);
    p.inside = !(p.ux < 0 || p.uy < 0 || p.ux >= a.cols || p.uy >= a.rows || p.vcurr_cp.z < 0);
    return p;
}

__device__ __forceinline__ void icp_row(const OdomState* st, const IcpArgs& a, const IcpPixel& p,
                                        f3 ncurr, f3 vprev_g, f3 nprev_g, float (&row)[7],
                                        float& found_f, float& err) {
    const m33& Rcurr = *reinterpret_cast<const m33*>(st->Rcurr);
    const m33& Rprev_inv = *reinterpret_cast<const m33*>(st->Rprev_inv);
    const f3 tprev = make_f3(st->tprev[0], st->tprev[1], st->tprev[2]);
    const f3 ncurr_g = Rcurr * ncurr;
    const float dist = norm(vprev_g - p.vcurr_g);
    const float sine = norm(cross(ncurr_g, nprev_g));
    err = p.inside ? (isfinite(dist) ? dist : 0.0f) : 0.0f;  // reduce.cu:275,299
    const bool found = p.inside && (sine < a.angle_thres && dist <= a.dist_thres &&
                                    !(ncurr.x != ncurr.x) && !(nprev_g.x != nprev_g.x));
#pragma unroll
    for (int k = 0; k < 7; ++k) row[k] = 0.f;
    if (found) {  // reduce.cu:320-329
        const f3 s_cp = Rprev_inv * (p.vcurr_g - tprev);
        const f3 d_cp = Rprev_inv * (vprev_g - tprev);
        const f3 n_cp = Rprev_inv * nprev_g;
        const f3 c = cross(s_cp, n_cp);
        row[0] = n_cp.x;
        row[1] = n_cp.y;
        row[2] = n_cp.z;
        row[3] = c.x;
        row[4] = c.y;
        row[5] = c.z;
        row[6] = dot(n_cp, s_cp - d_cp);
    }
    found_f = found ? 1.0f : 0.0f;
}

// PX pixels per lane per pass (PX = 4: one 16-byte load per plane; needs cols % 4 == 0 and
// 16-byte aligned rows, else PX = 1).
template <int PX, int MODE>
__global__ __launch_bounds__(kBlock) void icp_kernel(OdomState* __restrict__ st, IcpArgs a,
                                                     float* __restrict__ partials,
                                                     unsigned* __restrict__ ticket) {
    __shared__ GridReduceLds<float> lds;
    if (MODE == FINISH_GN && st->level_break) return;

    float sum[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) sum[k] = 0.f;

    const int N = a.cols * a.rows;
    const int rows = a.rows;
    for (int i0 = (blockIdx.x * kBlock + threadIdx.x) * PX; i0 < N; i0 += gridDim.x * kBlock * PX) {
        const int y = i0 / a.cols;
        const int x = i0 - y * a.cols;

        float vx[PX], vy[PX], vz[PX], nx[PX], ny[PX], nz[PX];
        if (PX == 4) {
            const float4 t0 = *reinterpret_cast<const float4*>(a.vmap_curr.base + (size_t)y * a.vmap_curr.stride + x);
            const float4 t1 = *reinterpret_cast<const float4*>(a.vmap_curr.base + (size_t)(y + rows) * a.vmap_curr.stride + x);
            const float4 t2 = *reinterpret_cast<const float4*>(a.vmap_curr.base + (size_t)(y + 2 * rows) * a.vmap_curr.stride + x);
            const float4 t3 = *reinterpret_cast<const float4*>(a.nmap_curr.base + (size_t)y * a.nmap_curr.stride + x);
            const float4 t4 = *reinterpret_cast<const float4*>(a.nmap_curr.base + (size_t)(y + rows) * a.nmap_curr.stride + x);
            const float4 t5 = *reinterpret_cast<const float4*>(a.nmap_curr.base + (size_t)(y + 2 * rows) * a.nmap_curr.stride + x);
            const float av[6][4] = {{t0.x, t0.y, t0.z, t0.w}, {t1.x, t1.y, t1.z, t1.w}, {t2.x, t2.y, t2.z, t2.w},
                                    {t3.x, t3.y, t3.z, t3.w}, {t4.x, t4.y, t4.z, t4.w}, {t5.x, t5.y, t5.z, t5.w}};
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                vx[p] = av[0][p];
                vy[p] = av[1][p];
                vz[p] = av[2][p];
                nx[p] = av[3][p];
                ny[p] = av[4][p];
                nz[p] = av[5][p];
            }
        } else {
            vx[0] = a.vmap_curr.base[(size_t)y * a.vmap_curr.stride + x];
            vy[0] = a.vmap_curr.base[(size_t)(y + rows) * a.vmap_curr.stride + x];
            vz[0] = a.vmap_curr.base[(size_t)(y + 2 * rows) * a.vmap_curr.stride + x];
            nx[0] = a.nmap_curr.base[(size_t)y * a.nmap_curr.stride + x];
            ny[0] = a.nmap_curr.base[(size_t)(y + rows) * a.nmap_curr.stride + x];
            nz[0] = a.nmap_curr.base[(size_t)(y + 2 * rows) * a.nmap_curr.stride + x];
        }

        IcpPixel px[PX];
        f3 vp[PX], np[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) px[p] = icp_project(st, a, make_f3(vx[p], vy[p], vz[p]));
        // all gathers of the group issued before any is consumed; pixels that project outside
        // read element 0 (a valid address) and are masked afterwards
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int ux = px[p].inside ? px[p].ux : 0, uy = px[p].inside ? px[p].uy : 0;
            const size_t ov = (size_t)uy * a.vmap_g_prev.stride + ux;
            const size_t on = (size_t)uy * a.nmap_g_prev.stride + ux;
            const size_t pv = (size_t)rows * a.vmap_g_prev.stride, pn = (size_t)rows * a.nmap_g_prev.stride;
            vp[p] = make_f3(a.vmap_g_prev.base[ov], a.vmap_g_prev.base[ov + pv], a.vmap_g_prev.base[ov + 2 * pv]);
            np[p] = make_f3(a.nmap_g_prev.base[on], a.nmap_g_prev.base[on + pn], a.nmap_g_prev.base[on + 2 * pn]);
        }
        float errs[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            float row[7], found;
            icp_row(st, a, px[p], make_f3(nx[p], ny[p], nz[p]), vp[p], np[p], row, found, errs[p]);
            accumulate_se3(sum, row, found);
        }
        if (a.err_map) {
            if (PX == 4)
                *reinterpret_cast<float4*>(a.err_map + (size_t)y * a.err_stride + x) =
                    make_float4(errs[0], errs[1], errs[2], errs[3]);
            else
                a.err_map[(size_t)y * a.err_stride + x] = errs[0];
        }
    }

    if (!grid_reduce<29>(sum, partials, ticket, lds)) return;
    if (threadIdx.x == 0) {
        const float* tot = lds.total;
        if (MODE == FINISH_RAW) {
            for (int k = 0; k < 29; ++k) st->out_f[k] = tot[k];
        } else {
            unpack_se3(tot, st->A_icp, st->b_icp);  // reduce.cu:458-472
            st->residual_icp[0] = tot[27];
            st->residual_icp[1] = tot[28];
            st->st.lastICPError = sqrtf(tot[27]) / tot[28];  // RGBDOdometry.cpp:412-413
            st->st.lastICPCount = tot[28];
            if (!st->rgb) solve_and_update(st, a.intr);
        }
    }
}

// ---- photometric correspondence pass ----------------------------------------------------
struct RgbResidualArgs {
    float min_scale, max_depth_delta;
    const int16_t *dIdx, *dIdy;
    int d_stride;  // in int16 elements
    const float *last_depth, *next_depth;
    int ld_stride, nd_stride;
    const uint8_t *last_image, *next_image;
    int li_stride, ni_stride;
    mmf_dataterm* corres;  // dense
    int cols, rows;
    float* err_map;
    int err_stride;
    LevelIntr intr;
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void rgb_residual_kernel(OdomState* __restrict__ st, RgbResidualArgs a,
                                                              int* __restrict__ partials,
                                                              unsigned* __restrict__ ticket) {
    __shared__ GridReduceLds<int> lds;
    if (MODE == FINISH_GN && st->level_break) return;
    int sum[2] = {0, 0};
    const int N = a.cols * a.rows, cols = a.cols, rows = a.rows;
    const float* K = st->krkinv;
    const float ktx = st->kt[0], kty = st->kt[1], ktz = st->kt[2];

    for (int k = blockIdx.x * kBlock + threadIdx.x; k < N; k += gridDim.x * kBlock) {
        const int i = k / cols, j0 = k - i * cols;
        mmf_dataterm c;
        c.zero_x = c.zero_y = c.one_x = c.one_y = 0;
        c.diff = 0.f;
        c.valid = 0;
        c.pad_[0] = c.pad_[1] = c.pad_[2] = 0;
        int vx = 0, vy = 0;
        if (j0 < cols - 5 && i < rows - 1) {  // reduce.cu:773
            bool valid = true;
            for (int u = max(i - 2, 0); u < min(i + 2, rows); u++)
                for (int v = max(j0 - 2, 0); v < min(j0 + 2, cols); v++)
                    valid = valid && (a.next_image[(size_t)u * a.ni_stride + v] > 0);
            if (valid) {
                const int valx = a.dIdx[(size_t)i * a.d_stride + j0], valy = a.dIdy[(size_t)i * a.d_stride + j0];
                const float mTwo = (float)((valx * valx) + (valy * valy));
                if (mTwo >= a.min_scale) {
                    const int y = i, x = j0;
                    const float d1 = a.next_depth[(size_t)y * a.nd_stride + x];
                    if (!(d1 != d1)) {
                        const float td1 = (float)(d1 * (K[6] * x + K[7] * y + K[8]) + ktz);
                        const int u0 = float2int_rn((d1 * (K[0] * x + K[1] * y + K[2]) + ktx) / td1);
                        const int v0 = float2int_rn((d1 * (K[3] * x + K[4] * y + K[5]) + kty) / td1);
                        if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
                            const float d0 = a.last_depth[(size_t)v0 * a.ld_stride + u0];
                            const uint8_t li = a.last_image[(size_t)v0 * a.li_stride + u0];
                            if (d0 > 0 && fabsf(td1 - d0) <= a.max_depth_delta && li != 0) {
                                c.zero_x = (int16_t)u0;
                                c.zero_y = (int16_t)v0;
                                c.one_x = (int16_t)x;
                                c.one_y = (int16_t)y;
                                c.diff = (float)a.next_image[(size_t)y * a.ni_stride + x] - (float)li;
                                c.valid = 1;
                                vx = 1;
                                vy = (int)(c.diff * c.diff);
                            }
                        }
                    }
                }
            }
        }
        if (a.err_map) a.err_map[(size_t)i * a.err_stride + j0] = c.valid ? 0.001f * vy : 0.0f;
        *reinterpret_cast<int4*>(&a.corres[k]) = *reinterpret_cast<const int4*>(&c);
        sum[0] += vx;
        sum[1] += vy;
    }

    if (!grid_reduce<2>(sum, partials, ticket, lds)) return;
    if (threadIdx.x == 0) {
        if (MODE == FINISH_RAW) {
            st->out_i[0] = lds.total[0];
            st->out_i[1] = lds.total[1];
        } else {
            residual_finish(st, lds.total[0], lds.total[1]);
        }
    }
}

// ---- photometric Jacobian reduction -------------------------------------------------------
struct RgbStepArgs {
    const mmf_dataterm* corres;
    const float* cloud;  // AoS float3, dense
    float fx, fy;
    const int16_t *dIdx, *dIdy;
    int d_stride;
    float sobel_scale;
    int cols, rows;
    LevelIntr intr;
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void rgb_step_kernel(OdomState* __restrict__ st, RgbStepArgs a,
                                                          float* __restrict__ partials,
                                                          unsigned* __restrict__ ticket) {
    __shared__ GridReduceLds<float> lds;
    if (MODE == FINISH_GN && st->level_break) return;
    float sum[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) sum[k] = 0.f;
    const int N = a.cols * a.rows;
    const float sigma = st->sigmaVal;

    for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
        const int4 raw = *reinterpret_cast<const int4*>(&a.corres[i]);
        mmf_dataterm c;
        *reinterpret_cast<int4*>(&c) = raw;
        float row[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) row[k] = 0.f;
        const bool found = c.valid != 0;
        if (found) {  // reduce.cu:504-535
            float w = sigma + fabsf(c.diff);
            w = w > FLT_EPSILON ? 1.0f / w : 1.0f;
            if (sigma == -1) w = 1;
            row[6] = -w * c.diff;
            const float* cp = a.cloud + (size_t)(c.zero_y * a.cols + c.zero_x) * 3;
            const float X = cp[0], Y = cp[1], Z = cp[2];
            const float invz = 1.0f / Z;  // (float)(1.0 / Z): double rounding is innocuous for division
            const float dI_dx = w * a.sobel_scale * a.dIdx[(size_t)c.one_y * a.d_stride + c.one_x];
            const float dI_dy = w * a.sobel_scale * a.dIdy[(size_t)c.one_y * a.d_stride + c.one_x];
            const float v0 = dI_dx * a.fx * invz;
            const float v1 = dI_dy * a.fy * invz;
            const float v2 = -(v0 * X + v1 * Y) * invz;
            row[0] = v0;
            row[1] = v1;
            row[2] = v2;
            row[3] = -Z * v1 + Y * v2;
            row[4] = Z * v0 - X * v2;
            row[5] = -Y * v0 + X * v1;
        }
        accumulate_se3(sum, row, found ? 1.0f : 0.0f);
    }

    if (!grid_reduce<29>(sum, partials, ticket, lds)) return;
    if (threadIdx.x == 0) {
        if (MODE == FINISH_RAW) {
            for (int k = 0; k < 29; ++k) st->out_f[k] = lds.total[k];
        } else {
            unpack_se3(lds.total, st->A_rgb, st->b_rgb);
            solve_and_update(st, a.intr);
        }
    }
}

// ---- SO3 pre-alignment ----------------------------------------------------------------------
struct So3Args {
    const uint8_t *last_image, *next_image;
    int l_stride, n_stride;
    int cols, rows;
    LevelIntr intr;
};

__device__ __forceinline__ void so3_gradient(const uint8_t* img, int stride, int x, int y, float& gx, float& gy) {
    const float actu = (float)img[(size_t)y * stride + x];  // reduce.cu:963-979
    float back = (float)img[(size_t)y * stride + x - 1];
    float fore = (float)img[(size_t)y * stride + x + 1];
    gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(size_t)(y - 1) * stride + x];
    fore = (float)img[(size_t)(y + 1) * stride + x];
    gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void so3_kernel(OdomState* __restrict__ st, So3Args a,
                                                     float* __restrict__ partials,
                                                     unsigned* __restrict__ ticket) {
    __shared__ GridReduceLds<float> lds;
    if (MODE == FINISH_GN && st->so3_done) return;
    float sum[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) sum[k] = 0.f;
    const int N = a.cols * a.rows, cols = a.cols, rows = a.rows;
    const m33& B = *reinterpret_cast<const m33*>(st->imageBasis);
    const m33& kinv = *reinterpret_cast<const m33*>(st->kinv);
    const float* krlr = st->krlr;

    for (int k = blockIdx.x * kBlock + threadIdx.x; k < N; k += gridDim.x * kBlock) {
        const int y = k / cols, x = k - y * cols;
        const f3 unwarped = make_f3((float)x, (float)y, 1.0f);
        const f3 warped = B * unwarped;
        const int wx = float2int_rn(warped.x / warped.z);
        const int wy = float2int_rn(warped.y / warped.z);
        const bool found = (wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 &&
                            x < cols - 1 && y >= 1 && y < rows - 1);
        float row[4] = {0.f, 0.f, 0.f, 0.f};
        if (found) {  // reduce.cu:1011-1046
            float gnx, gny, glx, gly;
            so3_gradient(a.next_image, a.n_stride, wx, wy, gnx, gny);
            so3_gradient(a.last_image, a.l_stride, x, y, glx, gly);
            const float gx = (gnx + glx) / 2.0f;
            const float gy = (gny + gly) / 2.0f;
            const f3 point = kinv * unwarped;
            const float z2 = point.z * point.z;
            const float A = krlr[0], Bc = krlr[1], C = krlr[2];
            const float D = krlr[3], E = krlr[4], F = krlr[5];
            const float G = krlr[6], H = krlr[7], I = krlr[8];
            f3 left;
            left.x = ((point.z * (D * gy + A * gx)) - (gy * G * y) - (gx * G * x)) / z2;
            left.y = ((point.z * (E * gy + Bc * gx)) - (gy * H * y) - (gx * H * x)) / z2;
            left.z = ((point.z * (F * gy + C * gx)) - (gy * I * y) - (gx * I * x)) / z2;
            const f3 jac = cross(left, point);
            row[0] = jac.x;
            row[1] = jac.y;
            row[2] = jac.z;
            row[3] = -((float)a.next_image[(size_t)wy * a.n_stride + wx] - (float)a.last_image[(size_t)y * a.l_stride + x]);
        }
        // member order of JtJJtrSO3 (types.cuh:154-162)
        sum[0] = sum[0] + row[0] * row[0];
        sum[1] = sum[1] + row[0] * row[1];
        sum[2] = sum[2] + row[0] * row[2];
        sum[3] = sum[3] + row[0] * row[3];
        sum[4] = sum[4] + row[1] * row[1];
        sum[5] = sum[5] + row[1] * row[2];
        sum[6] = sum[6] + row[1] * row[3];
        sum[7] = sum[7] + row[2] * row[2];
        sum[8] = sum[8] + row[2] * row[3];
        sum[9] = sum[9] + row[3] * row[3];
        sum[10] = sum[10] + (found ? 1.0f : 0.0f);
    }

    if (!grid_reduce<11>(sum, partials, ticket, lds)) return;
    if (threadIdx.x == 0) {
        if (MODE == FINISH_RAW) {
            for (int k = 0; k < 11; ++k) st->out_f[k] = lds.total[k];
        } else {
            so3_finish(st, lds.total, a.intr);
        }
    }
}

// ---- single-lane bookkeeping of the device-resident loop ----------------------------------
struct BeginArgs {
    float trans[3], rot[9];
    int rgb_only, icp, rgb, so3;
    float icp_weight;
    LevelIntr so3_intr;  // level 2
};

// RGBDOdometry.cpp:221-228, 237, 252-255, 316-328
__global__ void odom_begin_kernel(OdomState* st, BeginArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < 9; ++k) st->Rprev[k] = st->Rcurr[k] = a.rot[k];
    for (int k = 0; k < 3; ++k) st->tprev[k] = st->tcurr[k] = a.trans[k];
    inverse3f(st->Rprev, st->Rprev_inv);
    st->rgb_only = a.rgb_only;
    st->icp = a.icp;
    st->rgb = a.rgb;
    st->so3 = a.so3;
    st->icp_weight = a.icp_weight;
    for (int k = 0; k < 9; ++k) {
        const double e = (k % 4 == 0) ? 1.0 : 0.0;
        st->resultR[k] = e;
        st->lastResultR[k] = e;
        st->R_lr[k] = (float)e;
    }
    st->so3_lastError = FLT_MAX / 2;
    st->so3_lastCount = FLT_MAX / 2;
    st->so3_done = a.so3 ? 0 : 1;
    st->level_break = 0;
    st->st.iterations_run = 0;
    st->st.so3_iterations_run = 0;
    if (a.so3) so3_prepare(st, a.so3_intr);
}

// start of a pyramid level: RGBDOdometry.cpp:320-328 (first level only), :344, :348-358
__global__ void gn_level_begin_kernel(OdomState* st, int first_level, LevelIntr intr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (first_level) {
        for (int k = 0; k < 16; ++k) st->resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
        if (st->so3)
            for (int x = 0; x < 3; ++x)
                for (int y = 0; y < 3; ++y) st->resultRt[x * 4 + y] = st->resultR[x * 3 + y];
    }
    st->st.lastRGBError = FLT_MAX;
    st->level_break = 0;
    rgb_prepare(st, intr);
}

// RGBDOdometry.cpp:464-467, 475-476
__global__ void odom_end_kernel(OdomState* st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->rgb) {
        const float dx = st->tcurr[0] - st->tprev[0], dy = st->tcurr[1] - st->tprev[1], dz = st->tcurr[2] - st->tprev[2];
        if (sqrtf(dx * dx + dy * dy + dz * dz) > 0.3) {
            for (int k = 0; k < 9; ++k) st->Rcurr[k] = st->Rprev[k];
            for (int k = 0; k < 3; ++k) st->tcurr[k] = st->tprev[k];
        }
    }
    for (int k = 0; k < 3; ++k) st->trans_out[k] = st->tcurr[k];
    for (int k = 0; k < 9; ++k) st->rot_out[k] = st->Rcurr[k];
}

}  // namespace mmf
