// codec.hip -- per-symbol coding tables of the y latents (reference MASIC.py:986-1044, :1262-1296).
//
// For every (latent pixel, non-zero channel) the reference evaluates the K-component Gaussian-mixture PMF over the symbol
// alphabet 0 .. 2*minmax (value - mean shifted by minmax, |.| applied before the two standardized cumulatives, sigma
// lower-bounded, softmax-ed mixture weights), clips it to [2^-16, 1], renormalises to 2^16 and rounds -- once per symbol,
// through .cpu().numpy().  Here one wavefront (64 lanes) builds one table on the device: lanes own contiguous runs of the
// alphabet, wave reductions give the normaliser, the count total and the mode, a wave scan the interval starts.
// Difference to the reference, container-level: the rounded counts are forced to total exactly 2^16 (the surplus goes to
// the mode) because the rANS coder needs a power-of-two total, where `range_coder` takes any total.
// The same kernel serves both directions: the decoder fetches whole tables (u16 starts) for the current wavefront of
// pixels, the encoder only the (start, freq) of the symbols it holds.  Encoder and decoder run the SAME instruction
// sequence on the same parameters, so their tables agree bit for bit.
#include "common.h"

namespace {

constexpr int CDF_MAX_L = 1024;

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int K>
__global__ __launch_bounds__(64) void gmm_cdf_kernel(const float* __restrict__ sigma, const float* __restrict__ mu,
                                                     const float* __restrict__ logits, int M, int HW,
                                                     const int* __restrict__ pix, const int* __restrict__ chan, int nch,
                                                     int minmax, float scale_bound, const float* __restrict__ y_hat,
                                                     unsigned short* __restrict__ starts, int* __restrict__ start_freq,
                                                     int* __restrict__ err) {
    __shared__ float pm[CDF_MAX_L];
    __shared__ unsigned fr[CDF_MAX_L];
    const int r = blockIdx.x, lane = threadIdx.x;
    const int p = pix[r / nch], m = chan[r % nch];
    if (p < 0) return;                                     // padding entry of a fixed-size wavefront list (graph replay)
    const int L = 2 * minmax + 1;
    const size_t base = (size_t)m * HW + p, ks = (size_t)M * HW;
    float wk[K], sk[K], mk[K];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wk[k] = logits[base + k * ks];
        mx = fmaxf(mx, wk[k]);
        sk[k] = fmaxf(sigma[base + k * ks], scale_bound);
        mk[k] = mu[base + k * ks] + (float)minmax;
    }
    float ws = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) { wk[k] = expf(wk[k] - mx); ws += wk[k]; }
#pragma unroll
    for (int k = 0; k < K; ++k) wk[k] = wk[k] / ws;

    const int chunk = (L + 63) / 64, s0 = lane * chunk, s1 = min(L, s0 + chunk);
    const float cst = -0.70710678118654752440f;
    float part = 0.0f;
    for (int s = s0; s < s1; ++s) {
        float l = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float v = fabsf((float)s - mk[k]);
            const float up = 0.5f * erfcf(cst * ((0.5f - v) / sk[k]));
            const float lo = 0.5f * erfcf(cst * ((-0.5f - v) / sk[k]));
            const float term = __fmul_rn(up - lo, wk[k]);
            l = (k == 0) ? term : __fadd_rn(l, term);
        }
        l = fminf(fmaxf(l, 1.0f / 65536.0f), 1.0f);
        pm[s] = l;
        part += l;
    }
    const float S = wave_sum(part);
    unsigned tot = 0, best = 0;
    int besti = 0x7fffffff;
    for (int s = s0; s < s1; ++s) {
        const float q = rintf(pm[s] / S * 65536.0f);
        const unsigned f = q < 1.0f ? 1u : (unsigned)q;
        fr[s] = f;
        tot += f;
        if (f > best) { best = f; besti = s; }
    }
    const unsigned T = wave_sum(tot);
    // mode: largest count, first index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const int fixed = (int)best + (65536 - (int)T);
    if (fixed < 1) {
        if (lane == 0) atomicOr(err, 1);
        return;
    }
    __syncthreads();
    if (lane == 0) fr[besti] = (unsigned)fixed;
    __syncthreads();
    unsigned run = 0;
    for (int s = s0; s < s1; ++s) run += fr[s];
    unsigned incl = run;                                   // inclusive scan over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    unsigned st = incl - run;
    int sym = -1;
    if (y_hat != nullptr) {
        sym = (int)y_hat[base] + minmax;
        if ((sym < 0 || sym >= L) && lane == 0) atomicOr(err, 2);
    }
    for (int s = s0; s < s1; ++s) {
        if (starts != nullptr) starts[(size_t)r * L + s] = (unsigned short)st;
        if (s == sym) { start_freq[2 * r] = (int)st; start_freq[2 * r + 1] = (int)fr[s]; }
        st += fr[s];
    }
}

}  // namespace

extern "C" int masic_gmm_cdf_rows(const float* sigma, const float* mu, const float* logits, int M, int K, int HW,
                                  const int32_t* pix, int npix, const int32_t* chan, int nch, int minmax, float scale_bound,
                                  const float* y_hat, uint16_t* starts, int32_t* start_freq, int32_t* err_flag, void* stream) {
    MASIC_REQUIRE(sigma && mu && logits && pix && chan && err_flag, MASIC_ERR_ARG, "gmm_cdf_rows: null pointer");
    MASIC_REQUIRE((y_hat != nullptr) == (start_freq != nullptr), MASIC_ERR_ARG, "gmm_cdf_rows: y_hat and start_freq go together");
    MASIC_REQUIRE(starts || start_freq, MASIC_ERR_ARG, "gmm_cdf_rows: no output requested");
    MASIC_REQUIRE(minmax >= 1 && 2 * minmax + 1 <= CDF_MAX_L, MASIC_ERR_UNSUPPORTED, "gmm_cdf_rows: minmax=%d (alphabet of at most %d symbols)", minmax, CDF_MAX_L);
    MASIC_REQUIRE(K >= 1 && K <= 8, MASIC_ERR_UNSUPPORTED, "gmm_cdf_rows: K=%d", K);
    if (npix <= 0 || nch <= 0) return MASIC_OK;
    const dim3 grid((unsigned)((size_t)npix * nch)), blk(64);
    hipStream_t st = (hipStream_t)stream;
#define CDF_CASE(KK)                                                                                                        \
    case KK:                                                                                                                \
        hipLaunchKernelGGL(gmm_cdf_kernel<KK>, grid, blk, 0, st, sigma, mu, logits, M, HW, (const int*)pix, (const int*)chan, nch, \
                           minmax, scale_bound, y_hat, (unsigned short*)starts, (int*)start_freq, (int*)err_flag);          \
        break;
    switch (K) { CDF_CASE(1) CDF_CASE(2) CDF_CASE(3) CDF_CASE(4) CDF_CASE(5) CDF_CASE(6) CDF_CASE(7) CDF_CASE(8) }
#undef CDF_CASE
    return masic_launch_status("gmm_cdf_rows");
}
