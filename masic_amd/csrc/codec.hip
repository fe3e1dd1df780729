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
                                                     int* __restrict__ err, const int* __restrict__ step, int npix) {
    __shared__ float pm[CDF_MAX_L];
    __shared__ unsigned fr[CDF_MAX_L];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (step != nullptr) pix += (size_t)(*step) * npix;    // the wavefront list of coding step *step (device-side loop of the decoder)
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

int cdf_launch(const float* sigma, const float* mu, const float* logits, int M, int K, int HW, const int32_t* pix, const int32_t* step, int npix,
               const int32_t* chan, int nch, int minmax, float scale_bound, const float* y_hat, uint16_t* starts, int32_t* start_freq,
               int32_t* err_flag, void* stream) {
    const dim3 grid((unsigned)((size_t)npix * nch)), blk(64);
    hipStream_t st = (hipStream_t)stream;
#define CDF_CASE(KK)                                                                                                        \
    case KK:                                                                                                                \
        hipLaunchKernelGGL(gmm_cdf_kernel<KK>, grid, blk, 0, st, sigma, mu, logits, M, HW, (const int*)pix, (const int*)chan, nch, \
                           minmax, scale_bound, y_hat, (unsigned short*)starts, (int*)start_freq, (int*)err_flag, (const int*)step, npix); \
        break;
    switch (K) { CDF_CASE(1) CDF_CASE(2) CDF_CASE(3) CDF_CASE(4) CDF_CASE(5) CDF_CASE(6) CDF_CASE(7) CDF_CASE(8) }
#undef CDF_CASE
    return masic_launch_status("gmm_cdf_rows");
}

// ---- the decoder's symbol search on the device: one wavefront (64 lanes) per channel stream.
// The rANS state of a stream is one serial chain, but the streams are independent: with one stream per latent channel
// (masic_rans_encode_channels) the <= h symbols a channel has on the current coding wavefront are decoded by one wave while the other
// channels' waves do theirs.  The table rows do not depend on the coder state, so a wave first loads ALL its rows (lane l keeps entry l
// of every row; alphabets up to 64 entries in one register per row, larger ones in strides of 64), and the serial part is registers
// only: cum = x & 0xffff, one ballot finds the interval, x = freq * (x >> 16) + cum - start, and a 32-bit word is pulled in when x
// drops below 2^31 (the next words of the stream are prefetched two ahead).  The decoded values go straight into the latent
// (y_hat[chan][pix] = s - minmax), so the next coding step's context convolution sees them: nothing crosses PCIe inside the loop.
// *step is incremented by the last wave of the launch to finish (an arrival counter): every wave reads it at its start only, and the
// next reader is the next launch.
constexpr int DEC_MAX_ROWS = 64;      // rows (pixels) per coding step a wave keeps in registers: a wavefront of the 5x5 mask has <= h of them
__global__ __launch_bounds__(64) void rans_decode_step_kernel(const unsigned* __restrict__ words, const unsigned* __restrict__ word_off,
                                                              unsigned long long* __restrict__ state, unsigned* __restrict__ pos,
                                                              const unsigned short* __restrict__ starts, const int* __restrict__ pix_all,
                                                              int* __restrict__ step, int npix, const int* __restrict__ chan, int nch, int L,
                                                              int minmax, float* __restrict__ y_hat, int HW, const unsigned* __restrict__ word_cnt,
                                                              int* __restrict__ err, int* __restrict__ done, unsigned short* __restrict__ y16) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int st = *step;
    const int* pix = pix_all + (size_t)st * npix;
    const int nL = (L + 63) >> 6;                          // table entries per lane
    if (nL == 1 && npix <= DEC_MAX_ROWS) {
        unsigned ent[DEC_MAX_ROWS];                        // entry `lane` of row i (0x10000 past the end: never <= cum)
        int pr[DEC_MAX_ROWS];
#pragma unroll
        for (int i = 0; i < DEC_MAX_ROWS; ++i) {
            pr[i] = i < npix ? pix[i] : -1;
            ent[i] = (pr[i] >= 0 && lane < L) ? (unsigned)starts[((size_t)i * nch + c) * L + lane] : 0x10000u;
        }
        unsigned long long x = state[c];
        unsigned pp = pos[c];
        const unsigned* w = words + word_off[c];
        const unsigned wn = word_cnt[c];
        unsigned w0 = pp < wn ? w[pp] : 0u, w1 = pp + 1 < wn ? w[pp + 1] : 0u;
        const int m = chan[c];
        bool bad = false;
#pragma unroll
        for (int i = 0; i < DEC_MAX_ROWS; ++i) {
            if (pr[i] < 0) continue;
            const unsigned cum = (unsigned)(x & 0xffffu);
            const unsigned long long le = __ballot(ent[i] <= cum);          // entries 0 .. s are <= cum (strictly increasing starts)
            const int s = 63 - __builtin_clzll(le | 1ull);
            const unsigned sst = __shfl(ent[i], s, 64);
            const unsigned nxt = s + 1 < L ? __shfl(ent[i], s + 1, 64) : 0x10000u;
            x = (unsigned long long)(nxt - sst) * (x >> 16) + cum - sst;
            if (x < (1ull << 31)) {
                bad = bad || pp >= wn;
                x = (x << 32) | w0;
                ++pp;
                w0 = w1;
                w1 = pp + 1 < wn ? w[pp + 1] : 0u;
            }
            if (lane == 0) {
                y_hat[(size_t)m * HW + pr[i]] = (float)(s - minmax);
                if (y16 != nullptr) { const __bf16 bv = (__bf16)(float)(s - minmax); y16[((size_t)(m >> 4) * HW + pr[i]) * 16 + (m & 15)] = __builtin_bit_cast(unsigned short, bv); }
            }
        }
        if (lane == 0) {
            state[c] = x;
            pos[c] = pp;
            if (bad) atomicOr(err, 4);
        }
    } else {
        // general form (alphabets above 64 entries or very tall latents): rows re-read per symbol, entries in strides of 64
        unsigned long long x = state[c];
        unsigned pp = pos[c];
        const unsigned* w = words + word_off[c];
        const unsigned wn = word_cnt[c];
        const int m = chan[c];
        bool bad = false;
        for (int i = 0; i < npix; ++i) {
            const int p = pix[i];
            if (p < 0) continue;
            const unsigned short* row = starts + ((size_t)i * nch + c) * L;
            const unsigned cum = (unsigned)(x & 0xffffu);
            int s = -1;
            for (int k = 0; k < nL; ++k) {
                const int e = k * 64 + lane;
                const unsigned v = e < L ? (unsigned)row[e] : 0x10000u;
                const unsigned long long le = __ballot(v <= cum);
                if (le != 0) s = k * 64 + 63 - __builtin_clzll(le);
            }
            s = s < 0 ? 0 : s;
            const unsigned sst = row[s], nxt = s + 1 < L ? (unsigned)row[s + 1] : 0x10000u;
            x = (unsigned long long)(nxt - sst) * (x >> 16) + cum - sst;
            if (x < (1ull << 31)) {
                bad = bad || pp >= wn;
                x = (x << 32) | (pp < wn ? w[pp] : 0u);
                ++pp;
            }
            if (lane == 0) {
                y_hat[(size_t)m * HW + p] = (float)(s - minmax);
                if (y16 != nullptr) { const __bf16 bv = (__bf16)(float)(s - minmax); y16[((size_t)(m >> 4) * HW + p) * 16 + (m & 15)] = __builtin_bit_cast(unsigned short, bv); }
            }
        }
        if (lane == 0) {
            state[c] = x;
            pos[c] = pp;
            if (bad) atomicOr(err, 4);
        }
    }
    // the last wave to finish moves the loop on: the next launch (the next step's table kernel) reads the new *step
    __threadfence();
    if (lane == 0) {
        const int prev = atomicAdd(done, 1);
        if (prev == nch - 1) {
            *done = 0;
            *step = st + 1;
        }
    }
}

}  // namespace

// Device side of the decoder's loop (one coding step): masic_gmm_cdf_rows_at builds the tables of step *step, masic_rans_decode_step decodes
// its symbols into the latent and increments *step.  words: all channel streams back to back as little-endian 32-bit words; word_off[c] /
// word_cnt[c]: first word and number of words of channel c's stream; state[c] / pos[c]: coder state (initialised from the stream's first two
// words) and next word index (2) -- caller-initialised, kept between steps; pix_all: [nsteps][npix] pixel lists (-1 = padding);
// done: caller-zeroed int.  err_flag bit 2: a stream ended early.  y_f16k (nullable): the latent also as F16K bf16 [ceil16(M) / 16][HW][16]
// (exact for |value| <= 256) -- the input layout of the context model's kernels.
extern "C" int masic_rans_decode_step(const uint32_t* words, const uint32_t* word_off, const uint32_t* word_cnt, uint64_t* state, uint32_t* pos,
                                      const uint16_t* starts, const int32_t* pix_all, int32_t* step, int npix, const int32_t* chan, int nch,
                                      int L, int minmax, float* y_hat, void* y_f16k, int HW, int32_t* err_flag, int32_t* done, void* stream) {
    MASIC_REQUIRE(words && word_off && word_cnt && state && pos && starts && pix_all && step && chan && y_hat && err_flag && done, MASIC_ERR_ARG,
                  "rans_decode_step: null pointer");
    MASIC_REQUIRE(npix >= 1 && nch >= 1 && L >= 1 && L <= CDF_MAX_L && L == 2 * minmax + 1, MASIC_ERR_SHAPE, "rans_decode_step: bad shape");
    hipLaunchKernelGGL(rans_decode_step_kernel, dim3(nch), dim3(64), 0, (hipStream_t)stream, (const unsigned*)words, (const unsigned*)word_off,
                       (unsigned long long*)state, (unsigned*)pos, (const unsigned short*)starts, (const int*)pix_all, (int*)step, npix,
                       (const int*)chan, nch, L, minmax, y_hat, HW, (const unsigned*)word_cnt, (int*)err_flag, (int*)done, (unsigned short*)y_f16k);
    return masic_launch_status("rans_decode_step");
}

extern "C" int masic_gmm_cdf_rows_at(const float* sigma, const float* mu, const float* logits, int M, int K, int HW,
                                     const int32_t* pix_all, const int32_t* step, int npix, const int32_t* chan, int nch, int minmax, float scale_bound,
                                     uint16_t* starts, int32_t* err_flag, void* stream);

extern "C" int masic_gmm_cdf_rows(const float* sigma, const float* mu, const float* logits, int M, int K, int HW,
                                  const int32_t* pix, int npix, const int32_t* chan, int nch, int minmax, float scale_bound,
                                  const float* y_hat, uint16_t* starts, int32_t* start_freq, int32_t* err_flag, void* stream) {
    MASIC_REQUIRE(sigma && mu && logits && pix && chan && err_flag, MASIC_ERR_ARG, "gmm_cdf_rows: null pointer");
    MASIC_REQUIRE((y_hat != nullptr) == (start_freq != nullptr), MASIC_ERR_ARG, "gmm_cdf_rows: y_hat and start_freq go together");
    MASIC_REQUIRE(starts || start_freq, MASIC_ERR_ARG, "gmm_cdf_rows: no output requested");
    MASIC_REQUIRE(minmax >= 1 && 2 * minmax + 1 <= CDF_MAX_L, MASIC_ERR_UNSUPPORTED, "gmm_cdf_rows: minmax=%d (alphabet of at most %d symbols)", minmax, CDF_MAX_L);
    MASIC_REQUIRE(K >= 1 && K <= 8, MASIC_ERR_UNSUPPORTED, "gmm_cdf_rows: K=%d", K);
    if (npix <= 0 || nch <= 0) return MASIC_OK;
    return cdf_launch(sigma, mu, logits, M, K, HW, pix, nullptr, npix, chan, nch, minmax, scale_bound, y_hat, starts, start_freq, err_flag, stream);
}

// the tables of coding step *step (device int): pix_all is [nsteps][npix]
extern "C" int masic_gmm_cdf_rows_at(const float* sigma, const float* mu, const float* logits, int M, int K, int HW,
                                     const int32_t* pix_all, const int32_t* step, int npix, const int32_t* chan, int nch, int minmax, float scale_bound,
                                     uint16_t* starts, int32_t* err_flag, void* stream) {
    MASIC_REQUIRE(sigma && mu && logits && pix_all && step && chan && starts && err_flag, MASIC_ERR_ARG, "gmm_cdf_rows_at: null pointer");
    MASIC_REQUIRE(minmax >= 1 && 2 * minmax + 1 <= CDF_MAX_L, MASIC_ERR_UNSUPPORTED, "gmm_cdf_rows_at: minmax=%d (alphabet of at most %d symbols)", minmax, CDF_MAX_L);
    MASIC_REQUIRE(K >= 1 && K <= 8 && npix >= 1 && nch >= 1, MASIC_ERR_UNSUPPORTED, "gmm_cdf_rows_at: K=%d npix=%d nch=%d", K, npix, nch);
    return cdf_launch(sigma, mu, logits, M, K, HW, pix_all, step, npix, chan, nch, minmax, scale_bound, nullptr, starts, nullptr, err_flag, stream);
}
