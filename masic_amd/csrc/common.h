// common.h -- shared helpers for libmasic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include "masic_hip.h"

void masic_set_error(const char* fmt, ...);

#define MASIC_REQUIRE(cond, code, ...)        \
    do {                                      \
        if (!(cond)) {                        \
            masic_set_error(__VA_ARGS__);     \
            return (code);                    \
        }                                     \
    } while (0)

static inline int masic_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        masic_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MASIC_ERR_LAUNCH;
    }
    return MASIC_OK;
}

// Zero fill as a KERNEL node.  hipMemsetAsync becomes a memset node when the stream is being captured, and the first memset node of a
// graph captured by torch.cuda.graph runs unordered with the kernels around it from the second replay on when the graph is replayed on
// torch's default stream (ROCm 7.2 / torch 2.10: tools/memset_node_repro.py shows it with torch kernels only) --
// a kernel node keeps its place.  bytes must be a multiple of 4 and ptr 4-byte aligned.
__global__ static void masic_zero_kernel(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t masic_zero_async(void* ptr, size_t bytes, hipStream_t st, int site = 0) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    // experiment (tools/micro/memset_graph.hip): MASIC_ZERO_MEMSET=1|2 issues the fills as hipMemsetAsync (memset nodes
    // under capture); MASIC_ZERO_MEMSET_SITES = bit mask of the call sites that do (default all)
    static const bool as_memset = getenv("MASIC_ZERO_MEMSET") != nullptr && (getenv("MASIC_ZERO_MEMSET")[0] == '2');
    static const int sites = getenv("MASIC_ZERO_MEMSET_SITES") ? atoi(getenv("MASIC_ZERO_MEMSET_SITES")) : -1;
    if (as_memset && ((sites >> site) & 1)) return hipMemsetAsync(ptr, 0, bytes, st);
    size_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(masic_zero_kernel, dim3((unsigned)nb), dim3(256), 0, st, (unsigned*)ptr, n);
    return hipGetLastError();
}

// ---- plane copy: B x C planes of HW floats from a channel view of one NCHW buffer into a channel view of another, float4 per lane, four
// loads in flight per lane before the first store (copy_view / slice_copy / quantize: the concat and slice traffic of the training graphs ran
// at 2.5-3.7 TB/s on one-float-per-thread kernels that spent a 64-bit division per element on the index; this form: tools/bench_elementwise.py).
// OP: 0 round to nearest even, 1 + noise[linear input index], 3 copy.  gate (or null): multiply by gate[b][gate_c][p].
// Needs HW % 4 == 0 and 16-byte aligned bases (callers fall back to their scalar kernels otherwise).
template <int OP, bool GATE>
__global__ __launch_bounds__(256) static void masic_plane_copy_kernel(const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ gate,
                                                                     float* __restrict__ y, int C, int HW4, int src_ctot, int src_coff, int dst_ctot, int dst_coff,
                                                                     int gate_ctot, int gate_c) {
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const float4* xs = reinterpret_cast<const float4*>(x) + ((size_t)b * src_ctot + src_coff + c) * HW4;
    const float4* ns = OP == 1 ? reinterpret_cast<const float4*>(noise) + (size_t)plane * HW4 : nullptr;      // (noise: contiguous [B][C][HW])
    const float4* gs = GATE ? reinterpret_cast<const float4*>(gate) + ((size_t)b * gate_ctot + gate_c) * HW4 : nullptr;
    float4* yd = reinterpret_cast<float4*>(y) + ((size_t)b * dst_ctot + dst_coff + c) * HW4;
    const int step = gridDim.x * 256;
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < HW4; i0 += 4 * step) {
        float4 v[4], n[4], g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * step;
            if (i < HW4) {
                v[k] = xs[i];
                if (OP == 1) n[k] = ns[i];
                if (GATE) g[k] = gs[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * step;
            if (i < HW4) {
                float4 t = v[k];
                if (OP == 0) { t.x = rintf(t.x); t.y = rintf(t.y); t.z = rintf(t.z); t.w = rintf(t.w); }
                if (OP == 1) { t.x += n[k].x; t.y += n[k].y; t.z += n[k].z; t.w += n[k].w; }
                if (GATE) { t.x *= g[k].x; t.y *= g[k].y; t.z *= g[k].z; t.w *= g[k].w; }
                yd[i] = t;
            }
        }
    }
}
// true if launched; false: shape / alignment outside the vector form (the caller runs its scalar kernel)
static inline bool masic_plane_copy(const float* x, const float* noise, const float* gate, float* y, int B, int C, int HW, int src_ctot, int src_coff,
                                    int dst_ctot, int dst_coff, int gate_ctot, int gate_c, int op, hipStream_t st) {
    if (HW % 4 != 0 || (long)B * C > 65535 || B * C == 0 || (((size_t)x | (size_t)y | (size_t)noise | (size_t)gate) & 15)) return false;
    const int HW4 = HW / 4;
    int gx = (HW4 + 4 * 256 - 1) / (4 * 256);                 // one pass of four vectors per lane ...
    const int cap = (8192 + B * C - 1) / (B * C);             // ... unless that makes more than ~8k workgroups: then the lanes loop
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, B * C);
#define MASIC_PC(OPV, GV) hipLaunchKernelGGL((masic_plane_copy_kernel<OPV, GV>), grid, dim3(256), 0, st, x, noise, gate, y, C, HW4, src_ctot, src_coff, dst_ctot, dst_coff, gate_ctot, gate_c)
    if (gate) { if (op == 0) MASIC_PC(0, true); else if (op == 1) MASIC_PC(1, true); else MASIC_PC(3, true); }
    else { if (op == 0) MASIC_PC(0, false); else if (op == 1) MASIC_PC(1, false); else MASIC_PC(3, false); }
#undef MASIC_PC
    return true;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MASIC_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MASIC_ACT_LEAKY) return v >= 0.0f ? v : 0.01f * v;
    return v;
}
__device__ __forceinline__ float apply_inop(float v, int op) {
    if (op == MASIC_INOP_ABS) return fabsf(v);
    if (op == MASIC_INOP_ROUND) return rintf(v);
    return v;
}

// F.grid_sample's map of a normalised coordinate n in [-1, 1] to a pixel coordinate of an axis of `size` samples.
//   align_corners = true  (kornia 0.5.0's warp_perspective default, what MASIC runs):  (n + 1) * ((size - 1) / 2)
//   align_corners = false (kornia <= 0.4.1's default):                                 ((n + 1) * size - 1) / 2
// kornia is absent from this environment and the reference pins no result at this boundary ("parity
// unpinned"), so the convention is a run-time setting of the library, masic_set_warp_align_corners (default 1).
__device__ __forceinline__ float masic_grid_unnormalize(float n, int size, int align_corners) {
    return align_corners ? __fmul_rn(__fadd_rn(n, 1.0f), __fdiv_rn((float)(size - 1), 2.0f))
                         : __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(n, 1.0f), (float)size), 1.0f), 2.0f);
}
int masic_warp_align_corners_value();      // warp.hip
