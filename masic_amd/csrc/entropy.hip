// entropy.hip -- entropy-model kernels of the MASIC hot path (gfx950): quantisation, the factorized
// EntropyBottleneck likelihood, the K-component Gaussian-mixture conditional likelihood.
//
// Reference: compressai/entropy_models/entropy_models.py
//   _quantize                      :98-125
//   EntropyBottleneck.forward      :384-411, _likelihood :372-382, _logits_cumulative :350-369, loss :345-348
//   GaussianMixtureConditional_gf  :808-858 (+ LowerBound compressai/ops/bound_ops.py:36-42)
// and the softmax over K of coremasic/mywork/MASIC.py:389-393.
//
// All three are HBM-bound elementwise kernels: one thread per latent element, every operand read
// once, coalesced along W (consecutive lanes = consecutive pixels of one channel), every result
// written once.  The GMM kernel reads (1 + 3K)*4 bytes and writes 8 bytes per latent element.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                       const float* __restrict__ gate, float* __restrict__ y,
                                                       int C, int HW, int out_ctot, int out_coff,
                                                       int gate_ctot, int gate_c, int mode, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bc = i / HW;
        const int p = (int)(i - bc * HW);
        const int b = (int)(bc / C), c = (int)(bc - (size_t)b * C);
        float v = x[i];
        v = (mode == 1) ? v + noise[i] : (mode == 3 ? v : rintf(v));
        if (gate) v *= gate[((size_t)b * gate_ctot + gate_c) * HW + p];
        y[((size_t)b * out_ctot + out_coff + c) * HW + p] = v;
    }
}

__global__ __launch_bounds__(256) void symbols_kernel(const float* __restrict__ x, const float* __restrict__ median,
                                                      int32_t* __restrict__ sym, int C, int HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const float m = median ? median[c] : 0.0f;
        sym[i] = (int32_t)rintf(x[i] - m);
    }
}

// ---------------------------------------------------------------------------- EntropyBottleneck
__device__ __forceinline__ float softplus_t(float x) {   // F.softplus, beta=1, threshold=20
    return x > 20.0f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_t(float x) { return 1.0f / (1.0f + expf(-x)); }

struct EbParams {
    float m0[3], m1[9], m2[9], m3[9], m4[3];
    float b0[3], b1[3], b2[3], b3[3], b4;
    float f0[3], f1[3], f2[3], f3[3];
};

__device__ __forceinline__ void eb_load(const float* __restrict__ t, EbParams& p) {
    // table row: matrices (3,9,9,9,3) | biases (3,3,3,3,1) | factors (3,3,3,3); softplus / tanh applied here
#pragma unroll
    for (int i = 0; i < 3; ++i) p.m0[i] = softplus_t(t[i]);
#pragma unroll
    for (int i = 0; i < 9; ++i) { p.m1[i] = softplus_t(t[3 + i]); p.m2[i] = softplus_t(t[12 + i]); p.m3[i] = softplus_t(t[21 + i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) p.m4[i] = softplus_t(t[30 + i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) { p.b0[i] = t[33 + i]; p.b1[i] = t[36 + i]; p.b2[i] = t[39 + i]; p.b3[i] = t[42 + i]; }
    p.b4 = t[45];
#pragma unroll
    for (int i = 0; i < 3; ++i) { p.f0[i] = tanhf(t[46 + i]); p.f1[i] = tanhf(t[49 + i]); p.f2[i] = tanhf(t[52 + i]); p.f3[i] = tanhf(t[55 + i]); }
}

__device__ __forceinline__ void eb_layer33(const float* m, const float* b, const float* f, float* v) {
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s = m[3 * i] * v[0];
        s = fmaf(m[3 * i + 1], v[1], s);
        s = fmaf(m[3 * i + 2], v[2], s);
        s += b[i];
        o[i] = s + f[i] * tanhf(s);
    }
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
}

__device__ __forceinline__ float eb_logits(const EbParams& p, float x) {
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float s = p.m0[i] * x + p.b0[i];
        v[i] = s + p.f0[i] * tanhf(s);
    }
    eb_layer33(p.m1, p.b1, p.f1, v);
    eb_layer33(p.m2, p.b2, p.f2, v);
    eb_layer33(p.m3, p.b3, p.f3, v);
    float s = p.m4[0] * v[0];
    s = fmaf(p.m4[1], v[1], s);
    s = fmaf(p.m4[2], v[2], s);
    return s + p.b4;
}

__global__ __launch_bounds__(256) void eb_fwd_kernel(const float* __restrict__ z, const float* __restrict__ params,
                                                     const float* __restrict__ medians, const float* __restrict__ noise,
                                                     float* __restrict__ z_hat, float* __restrict__ lik,
                                                     int B, int C, int HW, int training, float lik_bound) {
    const size_t total = (size_t)B * C * HW;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t bc = i / HW;
    const int p = (int)(i - bc * HW);
    const int b = (int)(bc / C), c = (int)(bc - (size_t)b * C);
    EbParams prm;
    eb_load(params + (size_t)c * MASIC_EB_PARAMS_PER_CHANNEL, prm);
    const float med = medians[c];
    const float v = z[i];
    float vq;
    if (training) vq = v + noise[(size_t)c * HW * B + (size_t)p * B + b];   // (C,1,H*W*B) draw layout
    else vq = rintf(v - med) + med;
    const float lower = eb_logits(prm, vq - 0.5f);
    const float upper = eb_logits(prm, vq + 0.5f);
    const float sum = lower + upper;
    const float sign = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
    float l = fabsf(sigmoid_t(sign * upper) - sigmoid_t(sign * lower));
    z_hat[i] = vq;
    lik[i] = fmaxf(l, lik_bound);
}

__global__ __launch_bounds__(256) void eb_auxloss_kernel(const float* __restrict__ params, const float* __restrict__ quantiles,
                                                         float* __restrict__ out, int C, float target) {
    __shared__ float red[256];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < C * 3; i += 256) {
        const int c = i / 3, q = i - 3 * c;
        EbParams prm;
        eb_load(params + (size_t)c * MASIC_EB_PARAMS_PER_CHANNEL, prm);
        const float t = q == 0 ? -target : (q == 1 ? 0.0f : target);
        acc += fabsf(eb_logits(prm, quantiles[i]) - t);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ---------------------------------------------------------------------------- Gaussian mixture
template <int K>
__global__ __launch_bounds__(256) void gmm_fwd_kernel(const float* __restrict__ y, const float* __restrict__ noise,
                                                      const float* __restrict__ sigma, const float* __restrict__ mu,
                                                      const float* __restrict__ wts, float* __restrict__ y_hat,
                                                      float* __restrict__ lik, float* __restrict__ wts_out,
                                                      int M, int HW, int training, int logits,
                                                      float scale_bound, float lik_bound, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t bm = i / HW;
    const int p = (int)(i - bm * HW);
    const int b = (int)(bm / M), m = (int)(bm - (size_t)b * M);
    const size_t base = ((size_t)b * K * M + m) * HW + p;    // component k at + k*M*HW
    const size_t kstride = (size_t)M * HW;
    float yv = y[i];
    yv = training == 1 ? yv + noise[i] : (training == 2 ? yv : rintf(yv));      // 2: y IS the quantised latent (likelihood only)
    float wk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wk[k] = wts[base + k * kstride];
    if (logits) {   // softmax over K (MASIC.py:389-393): exp(x - max) / sum
        float mx = wk[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, wk[k]);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) { wk[k] = expf(wk[k] - mx); s += wk[k]; }
#pragma unroll
        for (int k = 0; k < K; ++k) wk[k] = wk[k] / s;
        if (wts_out) {
#pragma unroll
            for (int k = 0; k < K; ++k) wts_out[base + k * kstride] = wk[k];
        }
    }
    const float cst = -0.70710678118654752440f;   // float(-(2**-0.5)), entropy_models.py:765
    float l = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = fmaxf(sigma[base + k * kstride], scale_bound);
        const float v = fabsf(yv - mu[base + k * kstride]);
        const float up = 0.5f * erfcf(cst * ((0.5f - v) / s));
        const float lo = 0.5f * erfcf(cst * ((-0.5f - v) / s));
        const float term = __fmul_rn(up - lo, wk[k]);
        l = (k == 0) ? term : __fadd_rn(l, term);
    }
    if (y_hat != nullptr) y_hat[i] = yv;
    lik[i] = fmaxf(l, lik_bound);
}

__global__ __launch_bounds__(256) void softmax_k_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        int M, int K, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over B*M*HW
    if (i >= total) return;
    const size_t bm = i / HW;
    const int p = (int)(i - bm * HW);
    const int b = (int)(bm / M), m = (int)(bm - (size_t)b * M);
    const size_t base = ((size_t)b * K * M + m) * HW + p, ks = (size_t)M * HW;
    float mx = x[base];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[base + k * ks]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += expf(x[base + k * ks] - mx);
    for (int k = 0; k < K; ++k) y[base + k * ks] = expf(x[base + k * ks] - mx) / s;
}

int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g == 0 ? 1 : g));
}

}  // namespace

extern "C" int masic_quantize_fwd(const float* x, const float* noise, const float* gate, float* y,
                                  int B, int C, int H, int W, int out_ctot, int out_coff,
                                  int gate_ctot, int gate_c, int mode, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "quantize_fwd: null pointer");
    MASIC_REQUIRE(mode == 0 || mode == 1 || mode == 3, MASIC_ERR_ARG, "quantize_fwd: mode %d (symbols: use masic_symbols_fwd)", mode);
    MASIC_REQUIRE(mode != 1 || noise, MASIC_ERR_ARG, "quantize_fwd: noise mode without a noise tensor");
    MASIC_REQUIRE(out_coff >= 0 && out_coff + C <= out_ctot, MASIC_ERR_SHAPE, "quantize_fwd: output view out of range");
    const size_t total = (size_t)B * C * H * W;
    if (!masic_plane_copy(x, noise, gate, y, B, C, H * W, C, 0, out_ctot, out_coff, gate_ctot, gate_c, mode, (hipStream_t)stream))
        hipLaunchKernelGGL(quantize_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, noise, gate, y,
                           C, H * W, out_ctot, out_coff, gate_ctot, gate_c, mode, total);
    return masic_launch_status("quantize_fwd");
}

extern "C" int masic_symbols_fwd(const float* x, const float* median, int32_t* sym,
                                 int B, int C, int H, int W, void* stream) {
    MASIC_REQUIRE(x && sym, MASIC_ERR_ARG, "symbols_fwd: null pointer");
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(symbols_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, median, sym, C, H * W, total);
    return masic_launch_status("symbols_fwd");
}

extern "C" int masic_entropy_bottleneck_fwd(const float* z, const float* params, const float* medians,
                                            const float* noise, float* z_hat, float* lik,
                                            int B, int C, int H, int W, int training, float lik_bound, void* stream) {
    MASIC_REQUIRE(z && params && medians && z_hat && lik, MASIC_ERR_ARG, "entropy_bottleneck_fwd: null pointer");
    MASIC_REQUIRE(!training || noise, MASIC_ERR_ARG, "entropy_bottleneck_fwd: training without noise");
    const size_t total = (size_t)B * C * H * W;
    hipLaunchKernelGGL(eb_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       z, params, medians, noise, z_hat, lik, B, C, H * W, training, lik_bound);
    return masic_launch_status("entropy_bottleneck_fwd");
}

extern "C" int masic_entropy_bottleneck_auxloss(const float* params, const float* quantiles, float* out,
                                                int C, double tail_mass, void* stream) {
    MASIC_REQUIRE(params && quantiles && out, MASIC_ERR_ARG, "entropy_bottleneck_auxloss: null pointer");
    const float target = (float)__builtin_log(2.0 / tail_mass - 1.0);   // entropy_models.py:295
    hipLaunchKernelGGL(eb_auxloss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, params, quantiles, out, C, target);
    return masic_launch_status("entropy_bottleneck_auxloss");
}

extern "C" int masic_gmm_likelihood_fwd(const float* y, const float* noise, const float* sigma, const float* mu,
                                        const float* wts, float* y_hat, float* lik, float* wts_out,
                                        int B, int M, int K, int H, int W, int training, int weights_are_logits,
                                        float scale_bound, float lik_bound, void* stream) {
    MASIC_REQUIRE(y && sigma && mu && wts && lik && (y_hat || training == 2), MASIC_ERR_ARG, "gmm_likelihood_fwd: null pointer");
    MASIC_REQUIRE(training >= 0 && training <= 2, MASIC_ERR_ARG, "gmm_likelihood_fwd: training = %d", training);
    MASIC_REQUIRE(training != 1 || noise, MASIC_ERR_ARG, "gmm_likelihood_fwd: training without noise");
    MASIC_REQUIRE(K >= 1 && K <= 8, MASIC_ERR_UNSUPPORTED, "gmm_likelihood_fwd: K=%d", K);
    const size_t total = (size_t)B * M * H * W;
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define GMM_CASE(KK)                                                                                                  \
    case KK:                                                                                                          \
        hipLaunchKernelGGL(gmm_fwd_kernel<KK>, grid, blk, 0, st, y, noise, sigma, mu, wts, y_hat, lik, wts_out, M,    \
                           H * W, training, weights_are_logits, scale_bound, lik_bound, total);                       \
        break;
    switch (K) {
        GMM_CASE(1) GMM_CASE(2) GMM_CASE(3) GMM_CASE(4) GMM_CASE(5) GMM_CASE(6) GMM_CASE(7) GMM_CASE(8)
    }
#undef GMM_CASE
    return masic_launch_status("gmm_likelihood_fwd");
}

extern "C" int masic_softmax_k_fwd(const float* x, float* y, int B, int M, int K, int HW, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "softmax_k_fwd: null pointer");
    const size_t total = (size_t)B * M * HW;
    hipLaunchKernelGGL(softmax_k_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, M, K, HW, total);
    return masic_launch_status("softmax_k_fwd");
}
