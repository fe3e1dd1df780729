// backward.hip -- backward kernels of the MASIC hot path that are not contractions (gfx950).
//
// What torch autograd derives for the reference's graph, written out per op:
//   * activation / abs / gate / softmax-over-K backward, per-channel sums (bias, beta gradients)
//   * GDN elementwise pieces (the three 128x128 contractions of its backward reuse the conv / wgrad kernels)
//   * GaussianMixtureConditional_gf backward incl. both LowerBound rules (compressai/ops/bound_ops.py:40-42) and the
//     softmax over K (coremasic/mywork/MASIC.py:389-393)
//   * EntropyBottleneck backward (entropy_models.py:350-411): inputs and all 58 per-channel density parameters
//   * perspective-warp backward w.r.t. the source image (scatter-add of the 4 bilinear taps)
// All are HBM-bound elementwise kernels; the two with cross-element sums (channel sums, EB parameter gradients)
// reduce inside a workgroup and write one result per channel: deterministic, no atomics. The warp backward
// scatters with float atomics (order-dependent in the last bits, like torch's grid_sample backward on GPUs).
#include "common.h"

namespace {

int grid_for(size_t total, int cap = 8192) {
    size_t g = (total + 255) / 256;
    return (int)(g > (size_t)cap ? cap : (g == 0 ? 1 : g));
}

// ---------------------------------------------------------------------------- elementwise
enum { EW_ACT_BWD = 0, EW_ABS_BWD = 1, EW_SQUARE = 2, EW_ABS = 3, EW_AXPY = 4, EW_RECIP_SCALE = 5, EW_DIFF_SCALE = 6,
       EW_MUL = 7, EW_REPARAM = 8, EW_REPARAM_BWD = 9, EW_ADD = 10 };

__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                 size_t n, int op, float s0, float s1) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float r;
        switch (op) {
            case EW_ACT_BWD: {      // a = grad wrt activation output, b = activation output, s0 = act code
                const float yv = b[i];
                const int act = (int)s0;
                r = act == MASIC_ACT_RELU ? (yv > 0.0f ? a[i] : 0.0f) : (act == MASIC_ACT_LEAKY ? (yv > 0.0f ? a[i] : 0.01f * a[i]) : a[i]);
                break;
            }
            case EW_ABS_BWD: { const float xv = b[i]; r = xv > 0.0f ? a[i] : (xv < 0.0f ? -a[i] : 0.0f); break; }
            case EW_SQUARE: r = a[i] * a[i]; break;
            case EW_ABS: r = fabsf(a[i]); break;
            case EW_AXPY: r = s0 * a[i] + (b ? s1 * b[i] : 0.0f); break;
            case EW_RECIP_SCALE: r = s0 / a[i]; break;                     // d (c * sum log x) / dx = c / x
            case EW_DIFF_SCALE: r = s0 * (a[i] - b[i]); break;             // d sse: 2 c (a - b)
            case EW_MUL: r = a[i] * b[i]; break;
            case EW_REPARAM: { const float v = fmaxf(a[i], s0); r = v * v - s1; break; }           // parametrizers.py:61-64
            case EW_REPARAM_BWD: {  // a = grad wrt reparametrised value, b = stored parameter, s0 = bound
                const float p = b[i], g = a[i] * 2.0f * fmaxf(p, s0);
                r = (p >= s0 || g < 0.0f) ? g : 0.0f;
                break;
            }
            default: r = a[i] + b[i]; break;
        }
        y[i] = r;
    }
}

// per-channel sum over batch and pixels: out[c] = sum_{b,p} x[b,c,p]   (bias / beta gradients).
// Two deterministic stages: grid (C, NSPLIT) partial sums in float64 over strided (b, pixel-chunk) slices, then one
// block per channel adds the NSPLIT partials in a fixed order.
constexpr int CS_SPLIT = 64;

__global__ __launch_bounds__(256) void channel_sum_stage1(const float* __restrict__ x, double* __restrict__ partial,
                                                          int B, int C, int HW, int ctot, int coff, float* __restrict__ out_direct) {
    __shared__ double red[256];
    const int c = blockIdx.x, sp = blockIdx.y;
    // chunks of 4096 pixels (16-byte loads: 4 per thread), dealt round-robin over (b, chunk) pairs to the CS_SPLIT blocks of this
    // channel; a thread sums at most a few hundred values in float32, the cross-thread tree is float64 (deterministic order)
    const int chunks = (HW + 4095) / 4096;
    const bool vec = (HW & 3) == 0;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int w = sp; w < B * chunks; w += (int)gridDim.y) {
        const int b = w / chunks, ck = w - b * chunks;
        const float* p = x + ((size_t)b * ctot + coff + c) * HW;
        const int lo = ck * 4096, hi = lo + 4096 < HW ? lo + 4096 : HW;
        if (vec) {
            for (int i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(p + i);
                a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
            }
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += 256) a0 += p[i];
        }
    }
    red[threadIdx.x] = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (out_direct != nullptr) out_direct[c] = (float)red[0];      // one block per channel (gridDim.y == 1): no second stage
        else partial[(size_t)c * CS_SPLIT + sp] = red[0];
    }
}

__global__ __launch_bounds__(64) void channel_sum_stage2(const double* __restrict__ partial, float* __restrict__ out, int C) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double acc = 0.0;
    for (int i = 0; i < CS_SPLIT; ++i) acc += partial[(size_t)c * CS_SPLIT + i];
    out[c] = (float)acc;
}

// slice of a wider buffer -> contiguous (the backward of copy_view / torch.cat)
__global__ __launch_bounds__(256) void slice_copy_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW,
                                                         int ctot, int coff, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t bc = i / HW;
        const int p = (int)(i - bc * HW);
        const int b = (int)(bc / C), c = (int)(bc - (size_t)b * C);
        y[i] = x[((size_t)b * ctot + coff + c) * HW + p];
    }
}

// gate product backward: gx = g * gate[b,gc,p];  ggate[b,p] = sum_c g[b,c,p] * x[b,c,p]
// Block = 64 consecutive pixels x 16 channel groups (1024 threads): a thread walks every 16th channel of its pixel (coalesced 256-byte
// rows), the 16 partial sums of a pixel meet in LDS in a fixed order.  (One thread per pixel walking all channels -- 32 workgroups
// and a 384-step dependent chain at 8 x 32 x 32 latents -- took 137 us per launch.)
__global__ __launch_bounds__(1024) void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        const float* __restrict__ gate, float* __restrict__ gx,
                                                        float* __restrict__ ggate, int B, int C, int HW, int gate_ctot, int gate_c) {
    __shared__ float red[16][64];
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + px;               // over B*HW
    const bool ok = i < (size_t)B * HW;
    const int b = ok ? (int)(i / HW) : 0, p = ok ? (int)(i - (size_t)b * HW) : 0;
    const float gv = ok ? gate[((size_t)b * gate_ctot + gate_c) * HW + p] : 0.0f;
    float acc = 0.0f;
    if (ok) {
        for (int c = cg; c < C; c += 16) {
            const size_t k = ((size_t)b * C + c) * HW + p;
            const float gg = g[k];
            gx[k] = gg * gv;
            acc = fmaf(gg, x[k], acc);
        }
    }
    red[cg][px] = acc;
    __syncthreads();
    if (cg == 0 && ok) {
        float s = red[0][px];
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k][px];
        ggate[((size_t)b * gate_ctot + gate_c) * HW + p] = s;
    }
}

// softmax over K on the (B,K,M,HW) view, backward: gx_k = y_k (g_k - sum_j g_j y_j)
__global__ __launch_bounds__(256) void softmax_k_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                            float* __restrict__ gx, int M, int K, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t bm = i / HW;
    const int p = (int)(i - bm * HW);
    const int b = (int)(bm / M), m = (int)(bm - (size_t)b * M);
    const size_t base = ((size_t)b * K * M + m) * HW + p, ks = (size_t)M * HW;
    float dot = 0.0f;
    for (int k = 0; k < K; ++k) dot = fmaf(g[base + k * ks], y[base + k * ks], dot);
    for (int k = 0; k < K; ++k) gx[base + k * ks] = y[base + k * ks] * (g[base + k * ks] - dot);
}

// ---------------------------------------------------------------------------- GDN pieces
// given x, n = beta^ + gamma^ x^2 and g = dL/dy:  s = g * n^(-1/2) [inverse: g * n^(1/2)],
// t = dL/dn = -1/2 g x n^(-3/2)   [inverse: +1/2 g x n^(-1/2)]
__global__ __launch_bounds__(256) void gdn_bwd_pre_kernel(const float* __restrict__ x, const float* __restrict__ nrm,
                                                          const float* __restrict__ g, float* __restrict__ s,
                                                          float* __restrict__ t, size_t n, int inverse) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float nv = nrm[i], r = sqrtf(nv), gv = g[i], xv = x[i];
        if (inverse) { s[i] = gv * r; t[i] = 0.5f * gv * xv / r; }
        else { s[i] = gv / r; t[i] = -0.5f * gv * xv / (nv * r); }
    }
}
// dx = s + 2 x u
__global__ __launch_bounds__(256) void gdn_bwd_post_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                           const float* __restrict__ u, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dx[i] = fmaf(2.0f * x[i], u[i], s[i]);
}

// ---------------------------------------------------------------------------- (inverse) GDN backward, C <= 4, one pass
// encoder2.pre_gdn / decoder2.after_gdn (MASIC.py:560, :575: GDN(3) on full-resolution pictures).  The generic path above runs
// two 3-channel 1x1 convolutions, two elementwise passes and a 1x1 weight gradient (nine launches over 2 M pixels, ~200 us at
// 8 x 512 x 512); here a thread does the whole per-pixel algebra in registers and keeps the C + C^2 parameter sums, a block
// writes one partial per sum, and the finishing block adds the partials in float64 in a fixed order and applies the
// reparametrisation rules (parametrizers.py:61-64, bound_ops.py:40-42).  Float32 arithmetic in every precision mode.
template <int C>
__global__ __launch_bounds__(256) void gdn_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ beta, const float* __restrict__ gamma,
                                                            float* __restrict__ gx, float* __restrict__ part, int B, int HW, int inverse,
                                                            float b_bound, float g_bound, float ped) {
    float gam[C][C], bet[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const float bv = fmaxf(beta[i], b_bound);
        bet[i] = bv * bv - ped;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float gv = fmaxf(gamma[i * C + j], g_bound);
            gam[i][j] = gv * gv - ped;
        }
    }
    float sb[C], sg[C][C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        sb[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < C; ++j) sg[i][j] = 0.0f;
    }
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / HW, p = i - b * HW;
        const size_t base = b * C * (size_t)HW + p;
        float xv[C], gv[C], x2[C], sv[C], tv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { xv[c] = x[base + (size_t)c * HW]; gv[c] = g[base + (size_t)c * HW]; x2[c] = xv[c] * xv[c]; }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float n = bet[c];
#pragma unroll
            for (int j = 0; j < C; ++j) n = fmaf(gam[c][j], x2[j], n);
            const float r = sqrtf(n);
            if (inverse) { sv[c] = gv[c] * r; tv[c] = 0.5f * gv[c] * xv[c] / r; }
            else { sv[c] = gv[c] / r; tv[c] = -0.5f * gv[c] * xv[c] / (n * r); }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float u = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) u = fmaf(gam[c][j], tv[c], u);
            gx[base + (size_t)j * HW] = fmaf(2.0f * xv[j], u, sv[j]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            sb[c] += tv[c];
#pragma unroll
            for (int j = 0; j < C; ++j) sg[c][j] = fmaf(tv[c], x2[j], sg[c][j]);
        }
    }
    __shared__ float red[4][C + C * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < C + C * C; ++k) {
        float v = k < C ? sb[k < C ? k : 0] : sg[(k >= C ? k - C : 0) / C][(k >= C ? k - C : 0) % C];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < C + C * C)
        part[(size_t)blockIdx.x * (C + C * C) + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one block: sums [nblocks][C + C*C] partials per parameter in float64 (fixed order: wave k % 4 takes sum k, lanes stride over the
// blocks, butterfly) and applies the reparametrisation backward
__global__ __launch_bounds__(256) void gdn_bwd_small_finish(const float* __restrict__ part, int nblocks, int n, const float* __restrict__ beta,
                                                            const float* __restrict__ gamma, float* __restrict__ g_beta,
                                                            float* __restrict__ g_gamma, int C, float b_bound, float g_bound) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < n; k += 4) {
        double s = 0.0;
        for (int i0 = lane; i0 < nblocks; i0 += 512) {      // eight loads in flight per lane (a dependent load per add is a latency each)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = i0 + 64 * j < nblocks ? part[(size_t)(i0 + 64 * j) * n + k] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)v[j];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const float d = (float)s;
            const float pv = k < C ? beta[k] : gamma[k - C], bound = k < C ? b_bound : g_bound;
            const float gg = d * 2.0f * fmaxf(pv, bound);
            const float r = (pv >= bound || gg < 0.0f) ? gg : 0.0f;
            if (k < C) g_beta[k] = r; else g_gamma[k - C] = r;
        }
    }
}

// ---------------------------------------------------------------------------- Gaussian mixture backward
__device__ __forceinline__ float norm_pdf(float t) { return 0.3989422804014327f * expf(-0.5f * t * t); }

template <int K>
__global__ __launch_bounds__(256) void gmm_bwd_kernel(const float* __restrict__ y_hat, const float* __restrict__ sigma,
                                                      const float* __restrict__ mu, const float* __restrict__ wts,
                                                      const float* __restrict__ g_lik, const float* __restrict__ g_yhat,
                                                      float* __restrict__ g_y, float* __restrict__ g_sigma,
                                                      float* __restrict__ g_mu, float* __restrict__ g_w,
                                                      int M, int HW, int logits, float scale_bound, float lik_bound, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t bm = i / HW;
    const int p = (int)(i - bm * HW);
    const int b = (int)(bm / M), m = (int)(bm - (size_t)b * M);
    const size_t base = ((size_t)b * K * M + m) * HW + p, ks = (size_t)M * HW;
    const float yv = y_hat[i];
    float wk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wk[k] = wts[base + k * ks];
    if (logits) {
        float mx = wk[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, wk[k]);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) { wk[k] = expf(wk[k] - mx); s += wk[k]; }
#pragma unroll
        for (int k = 0; k < K; ++k) wk[k] = wk[k] / s;
    }
    const float cst = -0.70710678118654752440f;
    float lik = 0.0f, dcdf[K], dsig[K], dv[K], sgn[K];
    bool sig_ok[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float sraw = sigma[base + k * ks];
        const float s = fmaxf(sraw, scale_bound);
        const float d = yv - mu[base + k * ks];
        const float v = fabsf(d);
        const float ta = (0.5f - v) / s, tb = (-0.5f - v) / s;
        const float up = 0.5f * erfcf(cst * ta), lo = 0.5f * erfcf(cst * tb);
        const float pa = norm_pdf(ta), pb = norm_pdf(tb);
        dcdf[k] = up - lo;                                  // dL/dw_k
        dsig[k] = wk[k] * (tb * pb - ta * pa) / s;          // dL/ds_k
        dv[k] = -wk[k] * (pa - pb) / s;                     // dL/dv_k
        sgn[k] = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        sig_ok[k] = sraw >= scale_bound;
        lik += dcdf[k] * wk[k];
    }
    // LowerBound(lik, 1e-9): pass if lik >= bound or the gradient is negative
    float g = g_lik[i];
    g = (lik >= lik_bound || g < 0.0f) ? g : 0.0f;
    float gy = 0.0f, gw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float gs = g * dsig[k];
        g_sigma[base + k * ks] = (sig_ok[k] || gs < 0.0f) ? gs : 0.0f;      // LowerBound(sigma, 0.11)
        const float gv = g * dv[k] * sgn[k];
        g_mu[base + k * ks] = -gv;
        gy += gv;
        gw[k] = g * dcdf[k];
    }
    if (logits) {       // through the softmax over K
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) dot = fmaf(gw[k], wk[k], dot);
#pragma unroll
        for (int k = 0; k < K; ++k) gw[k] = wk[k] * (gw[k] - dot);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) g_w[base + k * ks] = gw[k];
    g_y[i] = gy + (g_yhat ? g_yhat[i] : 0.0f);               // y_hat = y + noise: identity
}

// ---------------------------------------------------------------------------- EntropyBottleneck backward
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

struct EbRaw { float m[33], b[13], f[12]; };     // raw parameters of one channel (table row)

// forward of the cumulative-logits net at x, keeping what the backward needs; returns logits
struct EbTape { float pre[4][3], th[4][3], in[5][3]; };   // pre-activation s, tanh(s), layer inputs

__device__ __forceinline__ float eb_forward_tape(const float* M /*softplus*/, const float* Bv, const float* F /*tanh(f)*/,
                                                 float x, EbTape& tp) {
    float v[3];
    tp.in[0][0] = x;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float s = M[i] * x + Bv[i];
        tp.pre[0][i] = s; tp.th[0][i] = tanhf(s);
        v[i] = s + F[i] * tp.th[0][i];
    }
#pragma unroll
    for (int l = 1; l < 4; ++l) {
        float o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tp.in[l][i] = v[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float* mr = M + 3 + 9 * (l - 1) + 3 * i;
            float s = mr[0] * v[0];
            s = fmaf(mr[1], v[1], s);
            s = fmaf(mr[2], v[2], s);
            s += Bv[3 * l + i];
            tp.pre[l][i] = s; tp.th[l][i] = tanhf(s);
            o[i] = s + F[3 * l + i] * tp.th[l][i];
        }
        v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) tp.in[4][i] = v[i];
    float s = M[30] * v[0];
    s = fmaf(M[31], v[1], s);
    s = fmaf(M[32], v[2], s);
    return s + Bv[12];
}

// backward of the net: accumulates d/d(softplus(M)), d/db, d/d(tanh f) into gM,gB,gF scaled by `gout`; returns d/dx
__device__ __forceinline__ float eb_backward_tape(const float* M, const float* F, const EbTape& tp, float gout,
                                                  float* gM, float* gB, float* gF) {
    float gv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { gM[30 + j] += gout * tp.in[4][j]; gv[j] = gout * M[30 + j]; }
    gB[12] += gout;
#pragma unroll
    for (int l = 3; l >= 1; --l) {
        float gin[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float th = tp.th[l][i];
            gF[3 * l + i] += gv[i] * th;
            const float gs = gv[i] * (1.0f + F[3 * l + i] * (1.0f - th * th));
            gB[3 * l + i] += gs;
            const float* mr = M + 3 + 9 * (l - 1) + 3 * i;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                gM[3 + 9 * (l - 1) + 3 * i + j] += gs * tp.in[l][j];
                gin[j] = fmaf(gs, mr[j], gin[j]);
            }
        }
        gv[0] = gin[0]; gv[1] = gin[1]; gv[2] = gin[2];
    }
    float gx = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float th = tp.th[0][i];
        gF[i] += gv[i] * th;
        const float gs = gv[i] * (1.0f + F[i] * (1.0f - th * th));
        gB[i] += gs;
        gM[i] += gs * tp.in[0][0];
        gx = fmaf(gs, M[i], gx);
    }
    return gx;
}

// one workgroup per channel: every element of the channel, parameter gradients reduced in LDS (deterministic)
__global__ __launch_bounds__(256) void eb_bwd_kernel(const float* __restrict__ z_hat, const float* __restrict__ params,
                                                     const float* __restrict__ g_lik, const float* __restrict__ g_zhat,
                                                     float* __restrict__ g_z, float* __restrict__ g_params,
                                                     int B, int C, int HW, float lik_bound) {
    const int c = blockIdx.x;
    const float* row = params + (size_t)c * MASIC_EB_PARAMS_PER_CHANNEL;
    float M[33], Bv[13], F[12];
#pragma unroll
    for (int i = 0; i < 33; ++i) M[i] = softplusf_(row[i]);
#pragma unroll
    for (int i = 0; i < 13; ++i) Bv[i] = row[33 + i];
#pragma unroll
    for (int i = 0; i < 12; ++i) F[i] = tanhf(row[46 + i]);
    float gM[33], gB[13], gF[12];
#pragma unroll
    for (int i = 0; i < 33; ++i) gM[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 13; ++i) gB[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 12; ++i) gF[i] = 0.0f;

    const int per = B * HW;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int b = e / HW, p = e - b * HW;
        const size_t idx = ((size_t)b * C + c) * HW + p;
        const float v = z_hat[idx];
        EbTape tl, tu;
        const float lower = eb_forward_tape(M, Bv, F, v - 0.5f, tl);
        const float upper = eb_forward_tape(M, Bv, F, v + 0.5f, tu);
        const float sum = lower + upper;
        const float sign = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
        const float su = sigmoidf_(sign * upper), sl = sigmoidf_(sign * lower);
        const float diff = su - sl, lik = fabsf(diff);
        float g = g_lik[idx];
        g = (lik >= lik_bound || g < 0.0f) ? g : 0.0f;                       // LowerBound(lik, 1e-9)
        const float gd = g * (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f));
        const float gu = gd * su * (1.0f - su) * sign;
        const float gl = -gd * sl * (1.0f - sl) * sign;
        float gx = eb_backward_tape(M, F, tu, gu, gM, gB, gF);
        gx += eb_backward_tape(M, F, tl, gl, gM, gB, gF);
        g_z[idx] = gx + (g_zhat ? g_zhat[idx] : 0.0f);                       // z_hat = z + noise: identity
    }
    // chain to the raw parameters: softplus' = sigmoid(raw), tanh' = 1 - tanh^2
#pragma unroll
    for (int i = 0; i < 33; ++i) gM[i] *= sigmoidf_(row[i]);
#pragma unroll
    for (int i = 0; i < 12; ++i) gF[i] *= (1.0f - F[i] * F[i]);
    __shared__ float red[256];
    auto reduce_store = [&](float val, int slot) {
        red[threadIdx.x] = val;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) g_params[(size_t)c * MASIC_EB_PARAMS_PER_CHANNEL + slot] = red[0];
        __syncthreads();
    };
#pragma unroll
    for (int i = 0; i < 33; ++i) reduce_store(gM[i], i);
#pragma unroll
    for (int i = 0; i < 13; ++i) reduce_store(gB[i], 33 + i);
#pragma unroll
    for (int i = 0; i < 12; ++i) reduce_store(gF[i], 46 + i);
}

// aux loss backward: d/d quantiles of sum |logits(q) - target| (all density parameters detached)
__global__ __launch_bounds__(256) void eb_auxloss_bwd_kernel(const float* __restrict__ params, const float* __restrict__ quantiles,
                                                             float* __restrict__ g_q, int C, float target, float gout) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * 3) return;
    const int c = i / 3, q = i - 3 * c;
    const float* row = params + (size_t)c * MASIC_EB_PARAMS_PER_CHANNEL;
    float M[33], Bv[13], F[12];
    for (int k = 0; k < 33; ++k) M[k] = softplusf_(row[k]);
    for (int k = 0; k < 13; ++k) Bv[k] = row[33 + k];
    for (int k = 0; k < 12; ++k) F[k] = tanhf(row[46 + k]);
    EbTape tp;
    const float lg = eb_forward_tape(M, Bv, F, quantiles[i], tp);
    const float t = q == 0 ? -target : (q == 1 ? 0.0f : target);
    const float d = lg - t;
    const float gs = gout * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
    float gM[33] = {0}, gB[13] = {0}, gF[12] = {0};
    g_q[i] = eb_backward_tape(M, F, tp, gs, gM, gB, gF);
}

// The auxiliary step of a training iteration (newtrain_codec_real.py:143-145: aux_loss = sum of EntropyBottleneck.loss(), backward) for
// up to four bottlenecks in two launches: every (channel, quantile) element evaluates logits(q) with its tape once -- |logits - target|
// for the loss, sign(logits - target) back through the tape for d/d quantiles -- then one block adds the elements per bottleneck in a
// fixed order.  Replaces, per bottleneck, a one-block forward kernel, a backward kernel and the torch glue around them (11 launches
// for HSIC's two bottlenecks).
struct EbAuxArgs { const float* params[4]; const float* quantiles[4]; float* g_q[4]; int C[4]; float target[4]; int n; };

__global__ __launch_bounds__(256) void eb_aux_fused_kernel(const EbAuxArgs a, float* __restrict__ absd, int pitch) {
    const int e = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.C[e] * 3) return;
    const int c = i / 3, q = i - 3 * c;
    const float* row = a.params[e] + (size_t)c * MASIC_EB_PARAMS_PER_CHANNEL;
    float M[33], Bv[13], F[12];
    for (int k = 0; k < 33; ++k) M[k] = softplusf_(row[k]);
    for (int k = 0; k < 13; ++k) Bv[k] = row[33 + k];
    for (int k = 0; k < 12; ++k) F[k] = tanhf(row[46 + k]);
    EbTape tp;
    const float lg = eb_forward_tape(M, Bv, F, a.quantiles[e][i], tp);
    const float t = q == 0 ? -a.target[e] : (q == 1 ? 0.0f : a.target[e]);
    const float d = lg - t;
    const float gs = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    float gM[33] = {0}, gB[13] = {0}, gF[12] = {0};
    a.g_q[e][i] = eb_backward_tape(M, F, tp, gs, gM, gB, gF);
    absd[(size_t)e * pitch + i] = fabsf(d);
}

// loss[0] = sum over bottlenecks (in index order) of the per-bottleneck sums loss[1 + e]
__global__ __launch_bounds__(256) void eb_aux_sum_kernel(const EbAuxArgs a, const float* __restrict__ absd, int pitch, float* __restrict__ loss) {
    __shared__ float red[256];
    float total = 0.0f;
    for (int e = 0; e < a.n; ++e) {
        float acc = 0.0f;
        for (int i = threadIdx.x; i < a.C[e] * 3; i += 256) acc += absd[(size_t)e * pitch + i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) { loss[1 + e] = red[0]; total = e == 0 ? red[0] : total + red[0]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = total;
}

// ---------------------------------------------------------------------------- warp backward (w.r.t. the source)
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ g_dst, const float* __restrict__ minv,
                                                       float* __restrict__ g_src, int C, int Hs, int Ws, int Hd, int Wd, int align_corners) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hd * Wd) return;
    const int oy = pix / Wd, ox = pix - oy * Wd;
    const float* m = minv + b * 9;
    const float gx = __fmul_rn(__fsub_rn(__fdiv_rn((float)ox, (float)(Wd - 1)), 0.5f), 2.0f);
    const float gy = __fmul_rn(__fsub_rn(__fdiv_rn((float)oy, (float)(Hd - 1)), 0.5f), 2.0f);
    const float X = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[0]), __fmul_rn(gy, m[1])), m[2]);
    const float Y = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[3]), __fmul_rn(gy, m[4])), m[5]);
    const float Z = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[6]), __fmul_rn(gy, m[7])), m[8]);
    const float scale = fabsf(Z) > 1e-8f ? __fdiv_rn(1.0f, __fadd_rn(Z, 1e-8f)) : 1.0f;
    const float nx = __fmul_rn(X, scale), ny = __fmul_rn(Y, scale);
    const float fx = masic_grid_unnormalize(nx, Ws, align_corners);
    const float fy = masic_grid_unnormalize(ny, Hs, align_corners);
    if (!(fx == fx) || !(fy == fy)) return;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f, ex = 1.0f - wx, ey = 1.0f - wy;
    const float lim = 1.0e9f;
    const int x0 = (int)fminf(fmaxf(x0f, -lim), lim), y0 = (int)fminf(fmaxf(y0f, -lim), lim);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Ws, vx1 = x1 >= 0 && x1 < Ws, vy0 = y0 >= 0 && y0 < Hs, vy1 = y1 >= 0 && y1 < Hs;
    const size_t splane = (size_t)Hs * Ws, dplane = (size_t)Hd * Wd;
    for (int c = 0; c < C; ++c) {
        const float g = g_dst[((size_t)b * C + c) * dplane + pix];
        float* s = g_src + ((size_t)b * C + c) * splane;
        if (vx0 && vy0) atomicAdd(s + (size_t)y0 * Ws + x0, g * ex * ey);
        if (vx1 && vy0) atomicAdd(s + (size_t)y0 * Ws + x1, g * wx * ey);
        if (vx0 && vy1) atomicAdd(s + (size_t)y1 * Ws + x0, g * ex * wy);
        if (vx1 && vy1) atomicAdd(s + (size_t)y1 * Ws + x1, g * wx * wy);
    }
}

// ---- the same adjoint as a GATHER: one thread per SOURCE pixel q collects g[p] w(p, q) from every destination pixel p whose bilinear
// footprint contains q.  The scatter form above spends four float atomics per element -- 268 M of them for a 32-channel feature map
// at 8 x 512 x 512, at the chip's atomic rate of ~1.3 TB/s of added bytes: 1.1 ms per launch, two per CQE training step.  Here:
//   * the destination pixels that can reach q are those whose sampling point s(p) lies in (qx - 1, qx + 1) x (qy - 1, qy + 1); s is a
//     projective map, so they lie inside the image of that square under s^-1 -- a convex quadrilateral as long as the homogeneous
//     coordinate keeps its sign on the four corners; its bounding box (padded by 0.05 pixels for rounding) is the candidate set;
//   * every candidate's sampling point and weights are evaluated with exactly the forward kernel's float operations, and a candidate
//     counts iff floor(s(p)) puts q among its four taps: the weight pairs (p, q) are the forward's, bit for bit -- an exact adjoint,
//     summed in a fixed order (no atomics: results are reproducible, which the scatter form's are not);
//   * up to WG_MAX matches are kept in registers and applied channel by channel: g is read ~4 x through the caches, g_src written once
//     (no zero fill before it).
// A pixel whose corners straddle the horizon of the homography, whose box holds more than WG_BOX candidates or which finds more than
// WG_MAX matches (magnification beyond ~2 x: (2 s)^2 destination pixels reach a source pixel at s destination pixels per source pixel)
// raises *flag: the caller then runs zero fill + scatter over the whole tensor (the two launches
// that follow test the flag on the device, no host round trip).
constexpr int WG_MAX = 16, WG_BOX = 196;
template <int N>
__device__ __forceinline__ void warp_gather_channels(const float* __restrict__ gb, float* __restrict__ sb, const int (&mp)[WG_MAX], const float (&mwx)[WG_MAX],
                                                     const float (&mwy)[WG_MAX], int C, size_t dplane, size_t splane) {
    int c = 0;
    for (; c + 2 <= C; c += 2) {
        const float* g0 = gb + (size_t)c * dplane;
        const float* g1 = g0 + dplane;
        float v0[N], v1[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { v0[k] = g0[mp[k]]; v1[k] = g1[mp[k]]; }
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int k = 0; k < N; ++k) { a0 += v0[k] * mwx[k] * mwy[k]; a1 += v1[k] * mwx[k] * mwy[k]; }
        sb[(size_t)c * splane] = a0;
        sb[(size_t)(c + 1) * splane] = a1;
    }
    for (; c < C; ++c) {
        const float* g0 = gb + (size_t)c * dplane;
        float a0 = 0.0f;
#pragma unroll
        for (int k = 0; k < N; ++k) a0 += g0[mp[k]] * mwx[k] * mwy[k];
        sb[(size_t)c * splane] = a0;
    }
}
__global__ __launch_bounds__(256) void warp_bwd_gather_kernel(const float* __restrict__ g_dst, const float* __restrict__ minv, float* __restrict__ g_src,
                                                              int* __restrict__ flag, int C, int Hs, int Ws, int Hd, int Wd, int align_corners) {
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Hs * Ws) return;
    const int qy = q / Ws, qx = q - qy * Ws;
    const float* m = minv + b * 9;
    // s^-1 on normalised coordinates = inverse(minv) (float64 cofactors; only used for the candidate box)
    const double a0 = m[0], a1 = m[1], a2 = m[2], a3 = m[3], a4 = m[4], a5 = m[5], a6 = m[6], a7 = m[7], a8 = m[8];
    const double c0 = a4 * a8 - a5 * a7, c1 = a2 * a7 - a1 * a8, c2 = a1 * a5 - a2 * a4;
    const double c3 = a5 * a6 - a3 * a8, c4 = a0 * a8 - a2 * a6, c5 = a2 * a3 - a0 * a5;
    const double c6 = a3 * a7 - a4 * a6, c7 = a1 * a6 - a0 * a7, c8 = a0 * a4 - a1 * a3;
    double lo_x = 1e30, hi_x = -1e30, lo_y = 1e30, hi_y = -1e30;
    int sign = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double sx = qx + ((k & 1) ? 1.0 : -1.0), sy = qy + ((k & 2) ? 1.0 : -1.0);
        const double nx = align_corners ? 2.0 * sx / (Ws - 1) - 1.0 : (2.0 * sx + 1.0) / Ws - 1.0;
        const double ny = align_corners ? 2.0 * sy / (Hs - 1) - 1.0 : (2.0 * sy + 1.0) / Hs - 1.0;
        const double X = c0 * nx + c1 * ny + c2, Y = c3 * nx + c4 * ny + c5, Z = c6 * nx + c7 * ny + c8;
        const int sg = Z > 0.0 ? 1 : -1;
        if (k == 0) sign = sg;
        if (sg != sign || !(fabs(Z) > 1e-12)) bad = true;
        const double px = (X / Z * 0.5 + 0.5) * (Wd - 1), py = (Y / Z * 0.5 + 0.5) * (Hd - 1);
        lo_x = fmin(lo_x, px); hi_x = fmax(hi_x, px); lo_y = fmin(lo_y, py); hi_y = fmax(hi_y, py);
    }
    if (!(lo_x == lo_x && hi_x == hi_x && lo_y == lo_y && hi_y == hi_y)) bad = true;
    int px0 = 0, px1 = -1, py0 = 0, py1 = -1;
    if (!bad) {
        // (the box of the quadrilateral already holds every p with s(p) in the open square; the pad only has to cover the difference between
        // this float64 inverse and the forward's float32 evaluation of s: ~1e-4 pixels at these picture sizes)
        const double pad = 0.05;
        const double fx0 = fmax(lo_x - pad, -1.0), fx1 = fmin(hi_x + pad, (double)Wd), fy0 = fmax(lo_y - pad, -1.0), fy1 = fmin(hi_y + pad, (double)Hd);
        px0 = (int)ceil(fx0); px1 = (int)floor(fx1); py0 = (int)ceil(fy0); py1 = (int)floor(fy1);
        px0 = px0 < 0 ? 0 : px0; py0 = py0 < 0 ? 0 : py0;
        px1 = px1 > Wd - 1 ? Wd - 1 : px1; py1 = py1 > Hd - 1 ? Hd - 1 : py1;
        if (px1 >= px0 && py1 >= py0 && (long)(px1 - px0 + 1) * (py1 - py0 + 1) > WG_BOX) bad = true;
    }
    int mp[WG_MAX];
    float mwx[WG_MAX], mwy[WG_MAX];
#pragma unroll
    for (int k = 0; k < WG_MAX; ++k) { mp[k] = 0; mwx[k] = 0.0f; mwy[k] = 0.0f; }
    int nm = 0;
    if (!bad) {
        for (int oy = py0; oy <= py1; ++oy) {
            const float gy = __fmul_rn(__fsub_rn(__fdiv_rn((float)oy, (float)(Hd - 1)), 0.5f), 2.0f);
            for (int ox = px0; ox <= px1; ++ox) {
                const float gx = __fmul_rn(__fsub_rn(__fdiv_rn((float)ox, (float)(Wd - 1)), 0.5f), 2.0f);
                const float X = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[0]), __fmul_rn(gy, m[1])), m[2]);
                const float Y = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[3]), __fmul_rn(gy, m[4])), m[5]);
                const float Z = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[6]), __fmul_rn(gy, m[7])), m[8]);
                const float scale = fabsf(Z) > 1e-8f ? __fdiv_rn(1.0f, __fadd_rn(Z, 1e-8f)) : 1.0f;
                const float fx = masic_grid_unnormalize(__fmul_rn(X, scale), Ws, align_corners);
                const float fy = masic_grid_unnormalize(__fmul_rn(Y, scale), Hs, align_corners);
                if (!(fx == fx) || !(fy == fy)) continue;
                const float x0f = floorf(fx), y0f = floorf(fy);
                const float lim = 1.0e9f;
                const int x0 = (int)fminf(fmaxf(x0f, -lim), lim), y0 = (int)fminf(fmaxf(y0f, -lim), lim);
                if ((qx != x0 && qx != x0 + 1) || (qy != y0 && qy != y0 + 1)) continue;
                const float wx = fx - x0f, wy = fy - y0f;
                if (nm == WG_MAX) { bad = true; break; }
                const int pi = oy * Wd + ox;
                const float vx = qx == x0 ? 1.0f - wx : wx, vy = qy == y0 ? 1.0f - wy : wy;
#pragma unroll
                for (int k = 0; k < WG_MAX; ++k)         // (static indices: the three lists stay in registers)
                    if (k == nm) { mp[k] = pi; mwx[k] = vx; mwy[k] = vy; }
                ++nm;
            }
            if (bad) break;
        }
    }
    if (bad) {
        *flag = 1;               // (benign race: every writer stores 1)
        return;
    }
    const size_t splane = (size_t)Hs * Ws, dplane = (size_t)Hd * Wd;
    const float* gb = g_dst + (size_t)b * C * dplane;
    float* sb = g_src + (size_t)b * C * splane + q;
    // the channel loop without branches: unused slots hold pixel 0 with weight 0, and the slot count is the wave's largest match count
    // rounded up to 4 / 6 / 9 / 16 (near-identity homographies: 4) -- every load of a channel is in flight before the first use
    int nmax = nm;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    if (nmax <= 4) warp_gather_channels<4>(gb, sb, mp, mwx, mwy, C, dplane, splane);
    else if (nmax <= 6) warp_gather_channels<6>(gb, sb, mp, mwx, mwy, C, dplane, splane);
    else if (nmax <= 9) warp_gather_channels<9>(gb, sb, mp, mwx, mwy, C, dplane, splane);
    else warp_gather_channels<WG_MAX>(gb, sb, mp, mwx, mwy, C, dplane, splane);
}
__global__ __launch_bounds__(256) void zero_if_kernel(float* __restrict__ p, size_t n, const int* __restrict__ flag) {
    if (*flag == 0) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.0f;
}
__global__ __launch_bounds__(256) void warp_bwd_if_kernel(const float* __restrict__ g_dst, const float* __restrict__ minv, float* __restrict__ g_src,
                                                          int C, int Hs, int Ws, int Hd, int Wd, int align_corners, const int* __restrict__ flag) {
    if (*flag == 0) return;
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hd * Wd) return;
    const int oy = pix / Wd, ox = pix - oy * Wd;
    const float* m = minv + b * 9;
    const float gx = __fmul_rn(__fsub_rn(__fdiv_rn((float)ox, (float)(Wd - 1)), 0.5f), 2.0f);
    const float gy = __fmul_rn(__fsub_rn(__fdiv_rn((float)oy, (float)(Hd - 1)), 0.5f), 2.0f);
    const float X = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[0]), __fmul_rn(gy, m[1])), m[2]);
    const float Y = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[3]), __fmul_rn(gy, m[4])), m[5]);
    const float Z = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[6]), __fmul_rn(gy, m[7])), m[8]);
    const float scale = fabsf(Z) > 1e-8f ? __fdiv_rn(1.0f, __fadd_rn(Z, 1e-8f)) : 1.0f;
    const float fx = masic_grid_unnormalize(__fmul_rn(X, scale), Ws, align_corners);
    const float fy = masic_grid_unnormalize(__fmul_rn(Y, scale), Hs, align_corners);
    if (!(fx == fx) || !(fy == fy)) return;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f, ex = 1.0f - wx, ey = 1.0f - wy;
    const float lim = 1.0e9f;
    const int x0 = (int)fminf(fmaxf(x0f, -lim), lim), y0 = (int)fminf(fmaxf(y0f, -lim), lim);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Ws, vx1 = x1 >= 0 && x1 < Ws, vy0 = y0 >= 0 && y0 < Hs, vy1 = y1 >= 0 && y1 < Hs;
    const size_t splane = (size_t)Hs * Ws, dplane = (size_t)Hd * Wd;
    for (int c = 0; c < C; ++c) {
        const float g = g_dst[((size_t)b * C + c) * dplane + pix];
        float* s = g_src + ((size_t)b * C + c) * splane;
        if (vx0 && vy0) atomicAdd(s + (size_t)y0 * Ws + x0, g * ex * ey);
        if (vx1 && vy0) atomicAdd(s + (size_t)y0 * Ws + x1, g * wx * ey);
        if (vx0 && vy1) atomicAdd(s + (size_t)y1 * Ws + x0, g * ex * wy);
        if (vx1 && vy1) atomicAdd(s + (size_t)y1 * Ws + x1, g * wx * wy);
    }
}

}  // namespace

extern "C" int masic_elementwise(const float* a, const float* b, float* y, size_t n, int op, float s0, float s1, void* stream) {
    MASIC_REQUIRE(a && y, MASIC_ERR_ARG, "elementwise: null pointer");
    MASIC_REQUIRE(op >= 0 && op <= EW_ADD, MASIC_ERR_ARG, "elementwise: op %d", op);
    const bool binary = op == EW_ACT_BWD || op == EW_ABS_BWD || op == EW_DIFF_SCALE || op == EW_MUL || op == EW_REPARAM_BWD || op == EW_ADD;
    MASIC_REQUIRE(!binary || b != nullptr, MASIC_ERR_ARG, "elementwise: op %d needs a second operand", op);
    hipLaunchKernelGGL(ew_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, op, s0, s1);
    return masic_launch_status("elementwise");
}

extern "C" size_t masic_channel_sum_workspace_bytes(int C) { return (size_t)C * CS_SPLIT * sizeof(double); }

extern "C" int masic_channel_sum(const float* x, float* out, void* workspace, int B, int C, int HW, int ctot, int coff, void* stream) {
    MASIC_REQUIRE(x && out && workspace, MASIC_ERR_ARG, "channel_sum: null pointer");
    MASIC_REQUIRE(coff >= 0 && coff + C <= ctot, MASIC_ERR_SHAPE, "channel_sum: view out of range");
    // latent-resolution tensors of many channels (the 1x1 layers of the entropy-parameter stacks: 8 x 32 x 32 pixels, <= 1152 channels):
    // one block per channel already fills the chip and a second launch would cost more than the sum itself
    if ((long)B * HW <= 16384 && C >= 128) {
        hipLaunchKernelGGL(channel_sum_stage1, dim3(C, 1), dim3(256), 0, (hipStream_t)stream, x, (double*)workspace, B, C, HW, ctot, coff, out);
        return masic_launch_status("channel_sum");
    }
    hipLaunchKernelGGL(channel_sum_stage1, dim3(C, CS_SPLIT), dim3(256), 0, (hipStream_t)stream, x, (double*)workspace, B, C, HW, ctot, coff, (float*)nullptr);
    hipLaunchKernelGGL(channel_sum_stage2, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, out, C);
    return masic_launch_status("channel_sum");
}

extern "C" int masic_slice_copy(const float* x, float* y, int B, int C, int HW, int ctot, int coff, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "slice_copy: null pointer");
    MASIC_REQUIRE(coff >= 0 && coff + C <= ctot, MASIC_ERR_SHAPE, "slice_copy: view out of range");
    const size_t total = (size_t)B * C * HW;
    if (!masic_plane_copy(x, nullptr, nullptr, y, B, C, HW, ctot, coff, C, 0, 0, 0, 3, (hipStream_t)stream))
        hipLaunchKernelGGL(slice_copy_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, HW, ctot, coff, total);
    return masic_launch_status("slice_copy");
}

extern "C" int masic_gate_bwd(const float* g, const float* x, const float* gate, float* gx, float* ggate,
                              int B, int C, int HW, int gate_ctot, int gate_c, void* stream) {
    MASIC_REQUIRE(g && x && gate && gx && ggate, MASIC_ERR_ARG, "gate_bwd: null pointer");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)(((size_t)B * HW + 63) / 64)), dim3(1024), 0, (hipStream_t)stream,
                       g, x, gate, gx, ggate, B, C, HW, gate_ctot, gate_c);
    return masic_launch_status("gate_bwd");
}

// Backward of the [C][58] parameter table of an EntropyBottleneck (torch.cat of 14 per-channel tensors, entropy_models.py:289-300): the
// table gradient split into the 14 parameter gradients with ONE launch, each a contiguous [C][w] block of `flat` (the caller hands
// views of it to autograd) -- torch's CatBackward + AccumulateGrad make 14 strided views and then 14 copies, per bottleneck and step.
namespace {
struct EbSplitArgs { int widths[16]; int nparts; };
__global__ void eb_table_split_kernel(const float* __restrict__ g, float* __restrict__ flat, int C, int ncol, const EbSplitArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * ncol) return;
    const int c = i / ncol, col = i - c * ncol;
    int start = 0, t = 0;
    while (t + 1 < a.nparts && col >= start + a.widths[t]) { start += a.widths[t]; ++t; }
    flat[(size_t)C * start + (size_t)c * a.widths[t] + (col - start)] = g[i];
}
}  // namespace
extern "C" int masic_eb_table_split(const float* g_table, float* flat, int C, const int* widths, int nparts, void* stream) {
    MASIC_REQUIRE(g_table && flat && widths && C > 0 && nparts > 0 && nparts <= 16, MASIC_ERR_ARG, "eb_table_split: bad argument");
    EbSplitArgs a{};
    int ncol = 0;
    for (int t = 0; t < nparts; ++t) {
        MASIC_REQUIRE(widths[t] > 0, MASIC_ERR_ARG, "eb_table_split: non-positive width");
        a.widths[t] = widths[t];
        ncol += widths[t];
    }
    a.nparts = nparts;
    hipLaunchKernelGGL(eb_table_split_kernel, dim3((unsigned)((C * ncol + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_table, flat, C, ncol, a);
    return masic_launch_status("eb_table_split");
}

extern "C" int masic_softmax_k_bwd(const float* g, const float* y, float* gx, int B, int M, int K, int HW, void* stream) {
    MASIC_REQUIRE(g && y && gx, MASIC_ERR_ARG, "softmax_k_bwd: null pointer");
    const size_t total = (size_t)B * M * HW;
    hipLaunchKernelGGL(softmax_k_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, y, gx, M, K, HW, total);
    return masic_launch_status("softmax_k_bwd");
}

extern "C" int masic_gdn_bwd_pre(const float* x, const float* nrm, const float* g, float* s, float* t, size_t n, int inverse, void* stream) {
    MASIC_REQUIRE(x && nrm && g && s && t, MASIC_ERR_ARG, "gdn_bwd_pre: null pointer");
    hipLaunchKernelGGL(gdn_bwd_pre_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, nrm, g, s, t, n, inverse);
    return masic_launch_status("gdn_bwd_pre");
}

extern "C" int masic_gdn_bwd_post(const float* x, const float* s, const float* u, float* dx, size_t n, void* stream) {
    MASIC_REQUIRE(x && s && u && dx, MASIC_ERR_ARG, "gdn_bwd_post: null pointer");
    hipLaunchKernelGGL(gdn_bwd_post_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, s, u, dx, n);
    return masic_launch_status("gdn_bwd_post");
}

extern "C" size_t masic_gdn_bwd_small_workspace_bytes() { return (size_t)2048 * 20 * sizeof(float); }

// x, g: float32 NCHW [B][C][H][W], C <= 4; beta [C], gamma [C][C]: the STORED tensors; gx: dL/dx; g_beta, g_gamma: gradients with
// respect to the stored tensors; workspace: masic_gdn_bwd_small_workspace_bytes() of device memory
extern "C" int masic_gdn_bwd_small(const float* x, const float* g, const float* beta, const float* gamma, float* gx, float* g_beta,
                                   float* g_gamma, void* workspace, int B, int C, int H, int W, int inverse, double beta_min, void* stream) {
    MASIC_REQUIRE(x && g && beta && gamma && gx && g_beta && g_gamma && workspace, MASIC_ERR_ARG, "gdn_bwd_small: null pointer");
    MASIC_REQUIRE(B > 0 && H > 0 && W > 0, MASIC_ERR_SHAPE, "gdn_bwd_small: non-positive dimension");
    MASIC_REQUIRE(C >= 1 && C <= 4, MASIC_ERR_UNSUPPORTED, "gdn_bwd_small: C=%d (1..4)", C);
    const double ped = 0x1p-36;
    const float b_bound = (float)__builtin_sqrt(beta_min + ped), g_bound = (float)__builtin_sqrt(ped);
    const size_t total = (size_t)B * H * W;
    const int nb = grid_for(total, 2048);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
#define GDNS_CASE(CC)                                                                                                              \
    case CC:                                                                                                                       \
        hipLaunchKernelGGL(gdn_bwd_small_kernel<CC>, dim3(nb), dim3(256), 0, st, x, g, beta, gamma, gx, part, B, H * W, inverse,   \
                           b_bound, g_bound, (float)ped);                                                                          \
        break;
    switch (C) { GDNS_CASE(1) GDNS_CASE(2) GDNS_CASE(3) GDNS_CASE(4) }
#undef GDNS_CASE
    hipLaunchKernelGGL(gdn_bwd_small_finish, dim3(1), dim3(256), 0, st, part, nb, C + C * C, beta, gamma, g_beta, g_gamma, C, b_bound, g_bound);
    return masic_launch_status("gdn_bwd_small");
}

extern "C" int masic_gmm_likelihood_bwd(const float* y_hat, const float* sigma, const float* mu, const float* wts,
                                        const float* g_lik, const float* g_yhat, float* g_y, float* g_sigma, float* g_mu,
                                        float* g_w, int B, int M, int K, int H, int W, int weights_are_logits,
                                        float scale_bound, float lik_bound, void* stream) {
    MASIC_REQUIRE(y_hat && sigma && mu && wts && g_lik && g_y && g_sigma && g_mu && g_w, MASIC_ERR_ARG, "gmm_likelihood_bwd: null pointer");
    MASIC_REQUIRE(K >= 1 && K <= 8, MASIC_ERR_UNSUPPORTED, "gmm_likelihood_bwd: K=%d", K);
    const size_t total = (size_t)B * M * H * W;
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define GMMB_CASE(KK)                                                                                                       \
    case KK:                                                                                                                \
        hipLaunchKernelGGL(gmm_bwd_kernel<KK>, grid, blk, 0, st, y_hat, sigma, mu, wts, g_lik, g_yhat, g_y, g_sigma, g_mu,  \
                           g_w, M, H * W, weights_are_logits, scale_bound, lik_bound, total);                               \
        break;
    switch (K) { GMMB_CASE(1) GMMB_CASE(2) GMMB_CASE(3) GMMB_CASE(4) GMMB_CASE(5) GMMB_CASE(6) GMMB_CASE(7) GMMB_CASE(8) }
#undef GMMB_CASE
    return masic_launch_status("gmm_likelihood_bwd");
}

extern "C" int masic_entropy_bottleneck_bwd(const float* z_hat, const float* params, const float* g_lik, const float* g_zhat,
                                            float* g_z, float* g_params, int B, int C, int H, int W, float lik_bound, void* stream) {
    MASIC_REQUIRE(z_hat && params && g_lik && g_z && g_params, MASIC_ERR_ARG, "entropy_bottleneck_bwd: null pointer");
    hipLaunchKernelGGL(eb_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, z_hat, params, g_lik, g_zhat, g_z, g_params,
                       B, C, H * W, lik_bound);
    return masic_launch_status("entropy_bottleneck_bwd");
}

extern "C" int masic_entropy_bottleneck_auxloss_bwd(const float* params, const float* quantiles, float* g_quantiles,
                                                    int C, double tail_mass, float gout, void* stream) {
    MASIC_REQUIRE(params && quantiles && g_quantiles, MASIC_ERR_ARG, "entropy_bottleneck_auxloss_bwd: null pointer");
    const float target = (float)__builtin_log(2.0 / tail_mass - 1.0);
    hipLaunchKernelGGL(eb_auxloss_bwd_kernel, dim3(ceil_div(C * 3, 256)), dim3(256), 0, (hipStream_t)stream, params, quantiles,
                       g_quantiles, C, target, gout);
    return masic_launch_status("entropy_bottleneck_auxloss_bwd");
}

// n <= 4 bottlenecks: params[e] [C[e]][58] (masic_eb_param_table), quantiles[e] / g_quantiles[e] [C[e]][3]; loss: 1 + n floats -- the
// total, then each bottleneck's sum; workspace: 4 * 3 * max C floats
extern "C" int masic_entropy_bottleneck_aux_step(const float* const* params, const float* const* quantiles, float* const* g_quantiles,
                                                 const int* C, const double* tail_mass, int n, float* loss, float* workspace, int workspace_floats,
                                                 void* stream) {
    MASIC_REQUIRE(params && quantiles && g_quantiles && C && tail_mass && loss && workspace, MASIC_ERR_ARG, "entropy_bottleneck_aux_step: null pointer");
    MASIC_REQUIRE(n >= 1 && n <= 4, MASIC_ERR_UNSUPPORTED, "entropy_bottleneck_aux_step: %d bottlenecks (1..4)", n);
    EbAuxArgs a{};
    a.n = n;
    int cmax = 0;
    for (int e = 0; e < n; ++e) {
        MASIC_REQUIRE(params[e] && quantiles[e] && g_quantiles[e] && C[e] > 0, MASIC_ERR_ARG, "entropy_bottleneck_aux_step: bottleneck %d", e);
        a.params[e] = params[e]; a.quantiles[e] = quantiles[e]; a.g_q[e] = g_quantiles[e]; a.C[e] = C[e];
        a.target[e] = (float)__builtin_log(2.0 / tail_mass[e] - 1.0);
        cmax = C[e] > cmax ? C[e] : cmax;
    }
    const int pitch = 3 * cmax;
    MASIC_REQUIRE(workspace_floats >= n * pitch, MASIC_ERR_ARG, "entropy_bottleneck_aux_step: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eb_aux_fused_kernel, dim3(ceil_div(pitch, 256), n), dim3(256), 0, st, a, workspace, pitch);
    hipLaunchKernelGGL(eb_aux_sum_kernel, dim3(1), dim3(256), 0, st, a, (const float*)workspace, pitch, loss);
    return masic_launch_status("entropy_bottleneck_aux_step");
}

// The gather form (see warp_bwd_gather_kernel): g_src needs NO zero fill; flag: one int32 on the device, zero on entry (set when some
// pixel's footprint was outside the gather form's bounds -- the scatter form has then redone the whole tensor, still inside this call).
extern "C" int masic_warp_perspective_bwd_gather(const float* g_dst, const float* minv_norm, float* g_src, int* flag,
                                                 int B, int C, int Hs, int Ws, int Hd, int Wd, void* stream) {
    MASIC_REQUIRE(g_dst && minv_norm && g_src && flag, MASIC_ERR_ARG, "warp_perspective_bwd_gather: null pointer");
    MASIC_REQUIRE(B > 0 && C > 0 && Hs > 1 && Ws > 1 && Hd > 1 && Wd > 1, MASIC_ERR_SHAPE, "warp_perspective_bwd_gather: sizes");
    hipStream_t st = (hipStream_t)stream;
    const int ac = masic_warp_align_corners_value();
    hipLaunchKernelGGL(warp_bwd_gather_kernel, dim3(ceil_div(Hs * Ws, 256), B), dim3(256), 0, st, g_dst, minv_norm, g_src, flag, C, Hs, Ws, Hd, Wd, ac);
    const size_t n = (size_t)B * C * Hs * Ws;
    hipLaunchKernelGGL(zero_if_kernel, dim3(grid_for(n)), dim3(256), 0, st, g_src, n, (const int*)flag);
    hipLaunchKernelGGL(warp_bwd_if_kernel, dim3(ceil_div(Hd * Wd, 256), B), dim3(256), 0, st, g_dst, minv_norm, g_src, C, Hs, Ws, Hd, Wd, ac, (const int*)flag);
    return masic_launch_status("warp_perspective_bwd_gather");
}

extern "C" int masic_warp_perspective_bwd(const float* g_dst, const float* minv_norm, float* g_src,
                                          int B, int C, int Hs, int Ws, int Hd, int Wd, void* stream) {
    MASIC_REQUIRE(g_dst && minv_norm && g_src, MASIC_ERR_ARG, "warp_perspective_bwd: null pointer");
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(ceil_div(Hd * Wd, 256), B), dim3(256), 0, (hipStream_t)stream, g_dst, minv_norm,
                       g_src, C, Hs, Ws, Hd, Wd, masic_warp_align_corners_value());
    return masic_launch_status("warp_perspective_bwd");
}
