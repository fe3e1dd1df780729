// common.h -- shared helpers for libmasic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
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
