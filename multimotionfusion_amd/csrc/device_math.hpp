// device_math.hpp -- float3 / mat33 helpers for the gfx950 tracking kernels.
//
// The operation ORDER mirrors the reference's operators (Core/Cuda/operators.cuh:56-91) and the
// translation unit is built with -ffp-contract=off, so a pixel's Jacobian row is bit-identical
// to the CPU oracle's (oracle/mmf_oracle.c).  `/` and sqrtf are IEEE correctly rounded under
// hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmf {

struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 make_f3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float norm(f3 a) { return sqrtf(dot(a, a)); }
// reference: rsqrtf (approximate on CUDA); here the correctly rounded 1/sqrt so the oracle and
// the kernel agree bit-for-bit.
__device__ __forceinline__ f3 normalized(f3 a) {
    const float rn = 1.0f / sqrtf(dot(a, a));
    return f3{a.x * rn, a.y * rn, a.z * rn};
}

// row-major 3x3 (types.cuh:61-73)
struct m33 {
    float m[9];
};
__device__ __forceinline__ f3 operator*(const m33& M, f3 a) {
    return f3{dot(f3{M.m[0], M.m[1], M.m[2]}, a), dot(f3{M.m[3], M.m[4], M.m[5]}, a),
              dot(f3{M.m[6], M.m[7], M.m[8]}, a)};
}

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fffffff); }  // cudafuncs.cu:131

// CUDA's __float2int_rn: nearest-even, NaN -> 0, saturating.
__device__ __forceinline__ int float2int_rn(float x) {
    if (x != x) return 0;
    x = fminf(fmaxf(x, -2147483648.0f), 2147483520.0f);
    return (int)rintf(x);
}

// ---- which XCD works on which blocks ------------------------------------------------------------
// The workgroups of a launch are dealt to the eight XCDs round-robin by their index, and every XCD has an L2 of its own.
// Neighbouring blocks of a pass over the IMAGE share cache lines -- the 3 x 3 window texels of fuse_data_kernel, the key
// image's lines under fuse_update_index_kernel's projection -- and dealt out round-robin the neighbours sit on eight different
// L2s, each of which fetches the shared lines again.  xcd_block maps a workgroup's index to the block it works on so that
// every XCD gets ONE contiguous eighth of the blocks: fuse_data_kernel 15.9 -> 13.7 us, fuse_update_index_kernel 11.7 -> 9.9
// (LABNOTES r5; the passes over the STORE in its own order -- clean_flag, clean_scatter -- gain nothing or lose a microsecond
// and keep the plain order).  A permutation of who does what: results are the same bits (MMF_XCD=0 / mmf_debug_set_xcd(0):
// the identity).
__host__ __device__ inline unsigned xcd_block_of(unsigned bx, unsigned nb) {  // (a permutation of [0, nb): tests/test_capi_exports.py)
    if (nb < 16u) return bx;
    const unsigned x = bx & 7u, slot = bx >> 3, q = nb >> 3, r = nb & 7u;
    return x * q + (x < r ? x : r) + slot;
}
__device__ int g_xcd_blocks = 1;
__device__ __forceinline__ unsigned xcd_block(unsigned bx, unsigned nb) { return g_xcd_blocks ? xcd_block_of(bx, nb) : bx; }

}  // namespace mmf
