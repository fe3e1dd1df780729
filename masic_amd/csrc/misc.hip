// misc.hip -- error state, version, small elementwise helpers and the RD-loss reductions.
//
// Reference: compressai/layers/layers.py:77 (`weight.data *= mask`), torch.cat at
// coremasic/mywork/MASIC.py:765,827 (copy_view), and the reductions of RateDistortionLoss
// (coremasic/mywork/newtrain_codec_real.py:79-83: sum(log(likelihoods)), MSE).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void masic_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int masic_version(void) { return MASIC_ABI_VERSION; }
extern "C" const char* masic_last_error(void) { return g_err; }

namespace {

constexpr int RED_BLOCKS = 1024;

__global__ __launch_bounds__(256) void mul_inplace_kernel(float* __restrict__ x, const float* __restrict__ m, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= m[i];
}

__global__ __launch_bounds__(256) void copy_view_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW,
                                                        int out_ctot, int out_coff, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bc = i / HW;
        const int p = (int)(i - bc * HW);
        const int b = (int)(bc / C), c = (int)(bc - (size_t)b * C);
        y[((size_t)b * out_ctot + out_coff + c) * HW + p] = x[i];
    }
}

__global__ __launch_bounds__(256) void lower_bound_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float bound, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = fmaxf(x[i], bound);
}

// bound_ops.py:40-42: the gradient passes where x >= bound or where it pushes x up (g < 0)
__global__ __launch_bounds__(256) void lower_bound_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              float* __restrict__ gx, float bound, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        gx[i] = (x[i] >= bound || g[i] < 0.0f) ? g[i] : 0.0f;
}

template <int OP>   // 0: sum log(a), 1: sum (a-b)^2
__global__ __launch_bounds__(256) void reduce_stage1(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                     double* __restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (OP == 0) acc += (double)logf(a[i]);
        else { const float d = a[i] - b[i]; acc += (double)(d * d); }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void reduce_stage2(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

int grid_for(size_t total, int cap) {
    size_t g = (total + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g == 0 ? 1 : g));
}

}  // namespace

extern "C" int masic_mul_inplace(float* x, const float* m, size_t n, void* stream) {
    MASIC_REQUIRE(x && m, MASIC_ERR_ARG, "mul_inplace: null pointer");
    hipLaunchKernelGGL(mul_inplace_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, m, n);
    return masic_launch_status("mul_inplace");
}

extern "C" int masic_lower_bound_fwd(const float* x, float* y, float bound, size_t n, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "lower_bound_fwd: null pointer");
    hipLaunchKernelGGL(lower_bound_fwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, bound, n);
    return masic_launch_status("lower_bound_fwd");
}

extern "C" int masic_lower_bound_bwd(const float* x, const float* g, float* gx, float bound, size_t n, void* stream) {
    MASIC_REQUIRE(x && g && gx, MASIC_ERR_ARG, "lower_bound_bwd: null pointer");
    hipLaunchKernelGGL(lower_bound_bwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, g, gx, bound, n);
    return masic_launch_status("lower_bound_bwd");
}

extern "C" int masic_copy_view(const float* x, float* y, int B, int C, int HW, int out_ctot, int out_coff, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "copy_view: null pointer");
    MASIC_REQUIRE(out_coff >= 0 && out_coff + C <= out_ctot, MASIC_ERR_SHAPE, "copy_view: output view out of range");
    const size_t total = (size_t)B * C * HW;
    hipLaunchKernelGGL(copy_view_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, C, HW,
                       out_ctot, out_coff, total);
    return masic_launch_status("copy_view");
}

extern "C" size_t masic_reduce_workspace_bytes(void) { return RED_BLOCKS * sizeof(double); }

extern "C" int masic_sum_log(const float* x, size_t n, double* out, void* workspace, void* stream) {
    MASIC_REQUIRE(x && out && workspace, MASIC_ERR_ARG, "sum_log: null pointer");
    const int g = grid_for(n, RED_BLOCKS);
    hipLaunchKernelGGL(reduce_stage1<0>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, n, (double*)workspace);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, g, out);
    return masic_launch_status("sum_log");
}

extern "C" int masic_sse(const float* a, const float* b, size_t n, double* out, void* workspace, void* stream) {
    MASIC_REQUIRE(a && b && out && workspace, MASIC_ERR_ARG, "sse: null pointer");
    const int g = grid_for(n, RED_BLOCKS);
    hipLaunchKernelGGL(reduce_stage1<1>, dim3(g), dim3(256), 0, (hipStream_t)stream, a, b, n, (double*)workspace);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, g, out);
    return masic_launch_status("sse");
}
