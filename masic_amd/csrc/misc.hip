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

// ---- the rate-distortion criterion (newtrain_codec_real.py:73-87) in two launches forward and one backward: the six reductions
// (two squared errors, up to four sums of log-likelihoods) used to be twelve launches plus ~20 one-element torch kernels of
// float64 scalar arithmetic, the backward six elementwise launches plus six multiplications by the incoming gradient.  Every sum
// is taken in exactly the order of reduce_stage1 / reduce_stage2 above (same grid per tensor, same strides): bit-identical values.
struct RdArgs {
    const float* a[6]; const float* b[6];      // tensors 0, 1: (x_hat, x) pairs of the squared errors; 2..: likelihood tensors (b unused)
    unsigned long long n[6];
    int g[6];                                   // stage-1 blocks of tensor t (grid_for(n[t], RED_BLOCKS))
    int nt;
};

__global__ __launch_bounds__(256) void rd_stage1(const RdArgs r, double* __restrict__ partial) {
    const int t = blockIdx.y;
    if ((int)blockIdx.x >= r.g[t]) return;
    __shared__ double red[256];
    const float* a = r.a[t];
    const float* b = r.b[t];
    const size_t n = r.n[t], stride = (size_t)r.g[t] * 256;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (t >= 2) acc += (double)logf(a[i]);
        else { const float d = a[i] - b[i]; acc += (double)(d * d); }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)t * RED_BLOCKS + blockIdx.x] = red[0];
}

struct RdOut { float* loss; double* mse1; double* mse2; double* bpp; double* per[4]; };

__global__ __launch_bounds__(256) void rd_finish(const RdArgs r, const double* __restrict__ partial, double cb, double cm, const RdOut o) {
    __shared__ double red[256];
    __shared__ double tot[6];
    for (int t = 0; t < r.nt; ++t) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < r.g[t]; i += 256) acc += partial[(size_t)t * RED_BLOCKS + i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) tot[t] = red[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (torch divides a device tensor by a host scalar as a product with the scalar's reciprocal; so does this)
        const double mse1 = tot[0] * (1.0 / (double)r.n[0]), mse2 = tot[1] * (1.0 / (double)r.n[1]);
        double sum = 0.0;
        for (int t = 2; t < r.nt; ++t) { sum = sum + tot[t]; *o.per[t - 2] = tot[t] * cb; }
        const double bpp = r.nt > 2 ? sum * cb : 0.0;
        *o.mse1 = mse1; *o.mse2 = mse2; *o.bpp = bpp;
        *o.loss = (float)(cm * (mse1 + mse2) + bpp);
    }
}

struct RdBwd { float* out[6]; float s[6]; };

// d loss / d x_hat = (s (x_hat - x)) g,  d loss / d lik = (s / lik) g;  g: the incoming gradient, a float32 device scalar
__global__ __launch_bounds__(256) void rd_bwd_kernel(const RdArgs r, const RdBwd w, const float* __restrict__ g) {
    const int t = blockIdx.y;
    const float gv = g[0], sc = w.s[t];
    const float* a = r.a[t];
    const float* b = r.b[t];
    float* y = w.out[t];
    const size_t n = r.n[t];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = (t >= 2 ? sc / a[i] : sc * (a[i] - b[i])) * gv;
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
    if (!masic_plane_copy(x, nullptr, nullptr, y, B, C, HW, C, 0, out_ctot, out_coff, 0, 0, 3, (hipStream_t)stream))
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

extern "C" size_t masic_rd_loss_workspace_bytes(void) { return (size_t)6 * RED_BLOCKS * sizeof(double); }

namespace {
int rd_args(RdArgs& r, const float* x1_hat, const float* x1, const float* x2_hat, const float* x2, size_t n_pic, const float* const* liks,
            const size_t* lik_n, int nliks) {
    r.a[0] = x1_hat; r.b[0] = x1; r.a[1] = x2_hat; r.b[1] = x2;
    r.n[0] = r.n[1] = n_pic;
    for (int k = 0; k < nliks; ++k) { r.a[2 + k] = liks[k]; r.b[2 + k] = nullptr; r.n[2 + k] = lik_n[k]; }
    r.nt = 2 + nliks;
    int gmax = 1;
    for (int t = 0; t < r.nt; ++t) { r.g[t] = grid_for(r.n[t], RED_BLOCKS); gmax = r.g[t] > gmax ? r.g[t] : gmax; }
    return gmax;
}
}  // namespace

// loss = cm (mse1 + mse2) + bpp (float32), mse_i = sum (x_i_hat - x_i)^2 / n_pic, bpp = cb sum_k sum log lik_k, per_k = cb sum log lik_k
// (float64 device scalars; newtrain_codec_real.py:73-87).  nliks <= 4 (0: the distortion-only criterion of the CQE stage).
extern "C" int masic_rd_loss(const float* x1_hat, const float* x1, const float* x2_hat, const float* x2, size_t n_pic,
                             const float* const* liks, const size_t* lik_n, int nliks, double cb, double cm,
                             float* loss, double* mse1, double* mse2, double* bpp, double* const* per, void* workspace, void* stream) {
    MASIC_REQUIRE(x1_hat && x1 && x2_hat && x2 && loss && mse1 && mse2 && bpp && workspace, MASIC_ERR_ARG, "rd_loss: null pointer");
    MASIC_REQUIRE(nliks >= 0 && nliks <= 4 && (nliks == 0 || (liks && lik_n && per)), MASIC_ERR_ARG, "rd_loss: 0..4 likelihood tensors");
    RdArgs r{};
    const int gmax = rd_args(r, x1_hat, x1, x2_hat, x2, n_pic, liks, lik_n, nliks);
    RdOut o{loss, mse1, mse2, bpp, {nullptr, nullptr, nullptr, nullptr}};
    for (int k = 0; k < nliks; ++k) {
        MASIC_REQUIRE(liks[k] && per[k], MASIC_ERR_ARG, "rd_loss: null likelihood tensor");
        o.per[k] = per[k];
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rd_stage1, dim3(gmax, r.nt), dim3(256), 0, st, r, (double*)workspace);
    hipLaunchKernelGGL(rd_finish, dim3(1), dim3(256), 0, st, r, (const double*)workspace, cb, cm, o);
    return masic_launch_status("rd_loss");
}

// gradients of masic_rd_loss's loss times the device scalar g: g_x1 = (s_pic (x1_hat - x1)) g, g_lik_k = (s_lik / lik_k) g
extern "C" int masic_rd_loss_bwd(const float* x1_hat, const float* x1, const float* x2_hat, const float* x2, size_t n_pic,
                                 const float* const* liks, const size_t* lik_n, int nliks, float s_pic, float s_lik, const float* g,
                                 float* g_x1, float* g_x2, float* const* g_liks, void* stream) {
    MASIC_REQUIRE(x1_hat && x1 && x2_hat && x2 && g && g_x1 && g_x2, MASIC_ERR_ARG, "rd_loss_bwd: null pointer");
    MASIC_REQUIRE(nliks >= 0 && nliks <= 4 && (nliks == 0 || (liks && lik_n && g_liks)), MASIC_ERR_ARG, "rd_loss_bwd: 0..4 likelihood tensors");
    RdArgs r{};
    rd_args(r, x1_hat, x1, x2_hat, x2, n_pic, liks, lik_n, nliks);
    RdBwd w{};
    w.out[0] = g_x1; w.out[1] = g_x2; w.s[0] = w.s[1] = s_pic;
    size_t nmax = n_pic;
    for (int k = 0; k < nliks; ++k) {
        MASIC_REQUIRE(liks[k] && g_liks[k], MASIC_ERR_ARG, "rd_loss_bwd: null likelihood tensor");
        w.out[2 + k] = g_liks[k]; w.s[2 + k] = s_lik;
        nmax = lik_n[k] > nmax ? lik_n[k] : nmax;
    }
    hipLaunchKernelGGL(rd_bwd_kernel, dim3(grid_for(nmax, 4096), r.nt), dim3(256), 0, (hipStream_t)stream, r, w, g);
    return masic_launch_status("rd_loss_bwd");
}

extern "C" int masic_sse(const float* a, const float* b, size_t n, double* out, void* workspace, void* stream) {
    MASIC_REQUIRE(a && b && out && workspace, MASIC_ERR_ARG, "sse: null pointer");
    const int g = grid_for(n, RED_BLOCKS);
    hipLaunchKernelGGL(reduce_stage1<1>, dim3(g), dim3(256), 0, (hipStream_t)stream, a, b, n, (double*)workspace);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, g, out);
    return masic_launch_status("sse");
}
