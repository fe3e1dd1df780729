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
