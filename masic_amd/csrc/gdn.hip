// gdn.hip -- fused one-pass GDN / inverse GDN for gfx950.
//
// Reference: compressai/layers/gdn.py:77-92 (norm = conv1x1(x^2, gamma^, beta^); y = x*rsqrt(norm),
// or x*sqrt(norm) when inverse) with the NonNegativeParametrizer of
// compressai/ops/parametrizers.py:47-64 folded in (stored beta/gamma -> max(.,bound)^2 - 2^-36).
//
// C = 128 (the only width HSIC uses for its 15 large GDNs): per workgroup, gamma^ lives in
// registers as the A operand of v_mfma_f32_32x32x2_f32 (wave w owns output channels 32w..32w+31,
// 64 VGPRs for K = 128), a [128 ch][64 px] tile of x is read coalesced from NCHW into LDS once,
// squared on the fly as the B operand, and the epilogue re-reads x from LDS, applies
// rsqrt/sqrt and stores coalesced -- x is read from HBM once and y written once
// (algorithmic traffic 2*4*C*H*W bytes).  Workgroups are persistent over pixel tiles so the
// 64 KB of gamma is fetched once per workgroup, not once per tile.
//
// Any other C (3 for pre_gdn/after_gdn, small test configs): a VALU kernel, one pixel per thread.
#include "common.h"

namespace {

constexpr int PT = 64;   // pixels per tile

__global__ __launch_bounds__(256) void gdn_mfma_c128(const float* __restrict__ x, const float* __restrict__ beta,
                                                     const float* __restrict__ gamma, float* __restrict__ y,
                                                     int HW, int tiles_per_image, int ntiles, int inverse,
                                                     float beta_bound, float gamma_bound, float pedestal) {
    constexpr int C = 128;
    __shared__ __attribute__((aligned(16))) float xt[C * PT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    // A operand: gamma^[i = 32w + j][k = 2kk + h], reparametrised on load
    float areg[C / 2];
#pragma unroll
    for (int kk = 0; kk < C / 2; ++kk) {
        const float g = fmaxf(gamma[(size_t)(32 * w + j) * C + 2 * kk + h], gamma_bound);
        areg[kk] = __fsub_rn(__fmul_rn(g, g), pedestal);
    }
    float bcoef[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float bv = fmaxf(beta[32 * w + (e & 3) + 8 * (e >> 2) + 4 * h], beta_bound);
        bcoef[e] = __fsub_rn(__fmul_rn(bv, bv), pedestal);
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_image;
        const int p0 = (tile - b * tiles_per_image) * PT;
        const float* xb = x + (size_t)b * C * HW + p0;
        float* yb = y + (size_t)b * C * HW + p0;
        // [128][64] floats = 2048 float4, 8 per thread, 256-byte rows
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256;
            const int row = idx >> 4, q = idx & 15;
            reinterpret_cast<float4*>(xt)[idx] = *(reinterpret_cast<const float4*>(xb + (size_t)row * HW) + q);
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            const float* col = xt + sub * 32 + j;
#pragma unroll
            for (int kk = 0; kk < C / 2; ++kk) {
                const float v = col[(2 * kk + h) * PT];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[kk], __fmul_rn(v, v), acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ch = 32 * w + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float xv = col[ch * PT];
                const float n = acc[e] + bcoef[e];
                const float s = sqrtf(n);
                yb[(size_t)ch * HW + sub * 32 + j] = inverse ? xv * s : xv * (1.0f / s);
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// bf16x3 variant (used when the forward runs with bf16 operands): gamma^ and x^2 are each split into bf16 hi + lo and
// n = ghi*xhi + ghi*xlo + glo*xhi on v_mfma_f32_32x32x16_bf16 (relative error ~2^-16 of a term, float32 accumulate) at
// 16/3 of the f32 matrix rate, which leaves the kernel HBM-bound.  No LDS round trip for x: with B = x^2 the fragment
// of lane (pixel j, half h) is 8 channels of ONE pixel, i.e. 8 coalesced NCHW loads, and the k <-> channel assignment
// of each MFMA step is chosen so that the 64 channels a lane loads are exactly the 64 output channels it owns in the
// accumulator layout -- x is read from HBM once, reused from registers in the epilogue, and y is written once.
// gamma^ fragments (hi, lo) sit in LDS in fragment order, built once per (persistent) workgroup.
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

__global__ __launch_bounds__(256, 2) void gdn_bf16x3_c128(const float* __restrict__ x, const float* __restrict__ beta,
                                                       const float* __restrict__ gamma, float* __restrict__ y,
                                                       unsigned short* __restrict__ y16,
                                                       int HW, int strips_per_image, int nstrips, int inverse,
                                                       float beta_bound, float gamma_bound, float pedestal) {
    constexpr int C = 128;
    __shared__ __attribute__((aligned(16))) uint4 Ahi[32 * 64];      // [(m*8+s)][lane] 8 bf16
    __shared__ __attribute__((aligned(16))) uint4 Alo[32 * 64];
    __shared__ float bet[C];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    // channel held by (step s, half hh, element c): 32-block q = s>>1, half-block t = s&1
    auto chan = [](int s, int hh, int c) { return 32 * (s >> 1) + 16 * (s & 1) + 8 * (c >> 2) + 4 * hh + (c & 3); };

    for (int idx = tid; idx < 32 * 64; idx += 256) {
        const int ms = idx >> 6, l = idx & 63;
        const int m = ms >> 3, s = ms & 7, r = l & 31, hh = l >> 5;
        __bf16 vh[8], vl[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float gr = fmaxf(gamma[(size_t)(32 * m + r) * C + chan(s, hh, c)], gamma_bound);
            split_bf16(__fsub_rn(__fmul_rn(gr, gr), pedestal), vh[c], vl[c]);
        }
        gbf16x8 ph, pl;
#pragma unroll
        for (int c = 0; c < 8; ++c) { ph[c] = vh[c]; pl[c] = vl[c]; }
        Ahi[idx] = __builtin_bit_cast(uint4, ph);
        Alo[idx] = __builtin_bit_cast(uint4, pl);
    }
    if (tid < C) {
        const float bv = fmaxf(beta[tid], beta_bound);
        bet[tid] = __fsub_rn(__fmul_rn(bv, bv), pedestal);
    }
    __syncthreads();

    const int gwave = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    for (int strip = gwave; strip < nstrips; strip += nwaves) {
        const int b = __builtin_amdgcn_readfirstlane(strip / strips_per_image);
        const int sp0 = __builtin_amdgcn_readfirstlane((strip - b * strips_per_image) * 32);
        // wave-uniform base + ONE 32-bit per-lane index (half h owns channels +4h); the channel part of every address is
        // a scalar offset, so the 64 loads / stores need no per-channel address registers
        const float* xw = x + (size_t)b * C * HW + sp0;
        float* yw = y + (size_t)b * C * HW + sp0;
        const unsigned vidx = (unsigned)(4 * h) * (unsigned)HW + (unsigned)j;
        float xv[8][8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) xv[s][c] = (xw + (size_t)chan(s, 0, c) * HW)[vidx];
        f32x16 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            gbf16x8 bh, bl;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                __bf16 hi, lo;
                split_bf16(__fmul_rn(xv[s][c], xv[s][c]), hi, lo);
                bh[c] = hi; bl[c] = lo;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const gbf16x8 ah = __builtin_bit_cast(gbf16x8, Ahi[(m * 8 + s) * 64 + lane]);
                const gbf16x8 al = __builtin_bit_cast(gbf16x8, Alo[(m * 8 + s) * 64 + lane]);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the 64 fragment reads from being hoisted to the top (register blow-up)
        }
        // epilogue: accumulator register e of block m <-> the value this lane loaded as xv[2m + (e>>3)][4*((e>>2)&1) + (e&3)]
        if (y16 != nullptr) {
            // F16K bf16 [B][8][HW][16] for the next convolution (conv_f16k.hip): 4 consecutive channels = one 8-byte store
            unsigned short* yr = y16 + ((size_t)b * 8 * HW + sp0 + j) * 16 + 4 * h;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * q + i;
                        const float xval = xv[2 * m + (e >> 3)][4 * ((e >> 2) & 1) + (e & 3)];
                        const float sq = sqrtf(acc[m][e] + bet[32 * m + 8 * q + i + 4 * h]);
                        o[i] = inverse ? xval * sq : xval * (1.0f / sq);
                    }
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    bf2 p0, p1;
                    p0[0] = (__bf16)o[0]; p0[1] = (__bf16)o[1]; p1[0] = (__bf16)o[2]; p1[1] = (__bf16)o[3];
                    uint2 st;
                    st.x = __builtin_bit_cast(unsigned, p0);
                    st.y = __builtin_bit_cast(unsigned, p1);
                    // channels 32m + 8q + 4h ..+3: record 2m + (q >> 1), offset 8 (q & 1) + 4h
                    *reinterpret_cast<uint2*>(yr + (size_t)(2 * m + (q >> 1)) * HW * 16 + 8 * (q & 1)) = st;
                }
            continue;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ch0 = 32 * m + (e & 3) + 8 * (e >> 2);          // + 4h per lane half
                const float xval = xv[2 * m + (e >> 3)][4 * ((e >> 2) & 1) + (e & 3)];
                const float n = acc[m][e] + bet[ch0 + 4 * h];
                const float sq = sqrtf(n);
                (yw + (size_t)ch0 * HW)[vidx] = inverse ? xval * sq : xval * (1.0f / sq);
            }
    }
}

// gamma^ (bf16 hi, lo) in MFMA A-fragment order and beta^ for the GDN epilogue of conv_f16k.hip:
//   img[((m*8 + s)*2 + hl)*64 + lane] = 8 bf16: gamma^[32m + (lane & 31)][chan(s, lane >> 5, c)], c = 0..7;  then beta^[128] floats
__global__ void gdn_pack_f16k_kernel(const float* __restrict__ beta, const float* __restrict__ gamma, uint4* __restrict__ img,
                                     float beta_bound, float gamma_bound, float pedestal) {
    constexpr int C = 128;
    auto chan = [](int s, int hh, int c) { return 32 * (s >> 1) + 16 * (s & 1) + 8 * (c >> 2) + 4 * hh + (c & 3); };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 32 * 64) {
        const int ms = idx >> 6, l = idx & 63;
        const int m = ms >> 3, s = ms & 7, r = l & 31, hh = l >> 5;
        gbf16x8 ph, pl;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float gr = fmaxf(gamma[(size_t)(32 * m + r) * C + chan(s, hh, c)], gamma_bound);
            __bf16 hi, lo;
            split_bf16(__fsub_rn(__fmul_rn(gr, gr), pedestal), hi, lo);
            ph[c] = hi; pl[c] = lo;
        }
        img[(ms * 2 + 0) * 64 + l] = __builtin_bit_cast(uint4, ph);
        img[(ms * 2 + 1) * 64 + l] = __builtin_bit_cast(uint4, pl);
    }
    if (idx < C) {
        const float bv = fmaxf(beta[idx], beta_bound);
        reinterpret_cast<float*>(img + 4 * 8 * 2 * 64)[idx] = __fsub_rn(__fmul_rn(bv, bv), pedestal);
    }
}

// generic: 64 threads per block, one pixel per thread, x^2 column in LDS
__global__ __launch_bounds__(64) void gdn_generic(const float* __restrict__ x, const float* __restrict__ beta,
                                                  const float* __restrict__ gamma, float* __restrict__ y,
                                                  int C, int HW, int inverse,
                                                  float beta_bound, float gamma_bound, float pedestal, int simplified) {
    extern __shared__ float sq[];   // [C][64]
    const int b = blockIdx.y;
    const int p = blockIdx.x * 64 + threadIdx.x;
    const bool live = p < HW;
    const float* xb = x + (size_t)b * C * HW + p;
    float* yb = y + (size_t)b * C * HW + p;
    for (int c = 0; c < C; ++c) {
        const float v = live ? xb[(size_t)c * HW] : 0.0f;
        sq[c * 64 + threadIdx.x] = simplified ? fabsf(v) : __fmul_rn(v, v);
    }
    // each thread only reads back its own column: no barrier needed
    for (int i = 0; i < C; ++i) {
        const float bv = fmaxf(beta[i], beta_bound);
        float n = __fsub_rn(__fmul_rn(bv, bv), pedestal);
        float s = 0.0f;
        for (int k = 0; k < C; ++k) {
            const float g = fmaxf(gamma[(size_t)i * C + k], gamma_bound);
            s = fmaf(__fsub_rn(__fmul_rn(g, g), pedestal), sq[k * 64 + threadIdx.x], s);
        }
        n += s;
        if (live) {
            const float xv = xb[(size_t)i * HW];
            const float r = simplified ? n : sqrtf(n);
            yb[(size_t)i * HW] = inverse ? xv * r : xv * (1.0f / r);
        }
    }
}

}  // namespace

extern "C" int masic_gdn_fwd_ex(const float* x, const float* beta, const float* gamma, float* y,
                                int B, int C, int H, int W, int inverse, double beta_min, int prec, void* stream);

extern "C" int masic_gdn_fwd(const float* x, const float* beta, const float* gamma, float* y,
                             int B, int C, int H, int W, int inverse, double beta_min, void* stream) {
    return masic_gdn_fwd_ex(x, beta, gamma, y, B, C, H, W, inverse, beta_min, MASIC_PREC_F32, stream);
}

extern "C" int masic_gdn_fwd_ex(const float* x, const float* beta, const float* gamma, float* y,
                                int B, int C, int H, int W, int inverse, double beta_min, int prec, void* stream) {
    MASIC_REQUIRE(x && beta && gamma && y, MASIC_ERR_ARG, "gdn_fwd: null pointer");
    MASIC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, MASIC_ERR_SHAPE, "gdn_fwd: non-positive dimension");
    MASIC_REQUIRE(C <= 512, MASIC_ERR_UNSUPPORTED, "gdn_fwd: C=%d > 512", C);
    // parametrizers.py:47-56: pedestal = (2^-18)^2, bound = sqrt(minimum + pedestal), held as float32 buffers
    const double ped = 0x1p-36;
    const float pedestal = (float)ped;
    const float beta_bound = (float)__builtin_sqrt(beta_min + ped);
    const float gamma_bound = (float)__builtin_sqrt(ped);
    const int HW = H * W;
    hipStream_t st = (hipStream_t)stream;
    if (C == 128 && HW % 32 == 0 && prec == MASIC_PREC_BF16) {
        const int spi = HW / 32, nstrips = spi * B;
        const int grid = nstrips < 4 * 512 ? (nstrips + 3) / 4 : 512;
        hipLaunchKernelGGL(gdn_bf16x3_c128, dim3(grid), dim3(256), 0, st, x, beta, gamma, y, (unsigned short*)nullptr, HW, spi, nstrips,
                           inverse, beta_bound, gamma_bound, pedestal);
    } else if (C == 128 && HW % PT == 0) {
        const int tpi = HW / PT, ntiles = tpi * B;
        const int grid = ntiles < 1024 ? ntiles : 1024;
        hipLaunchKernelGGL(gdn_mfma_c128, dim3(grid), dim3(256), 0, st, x, beta, gamma, y, HW, tpi, ntiles, inverse,
                           beta_bound, gamma_bound, pedestal);
    } else {
        hipLaunchKernelGGL(gdn_generic, dim3(ceil_div(HW, 64), B), dim3(64), (size_t)C * 64 * sizeof(float), st,
                           x, beta, gamma, y, C, HW, inverse, beta_bound, gamma_bound, pedestal, 0);
    }
    return masic_launch_status("gdn_fwd");
}

extern "C" int masic_gdn1_fwd(const float* x, const float* beta, const float* gamma, float* y,
                              int B, int C, int H, int W, int inverse, double beta_min, void* stream) {
    MASIC_REQUIRE(x && beta && gamma && y, MASIC_ERR_ARG, "gdn1_fwd: null pointer");
    MASIC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (size_t)C * 64 * sizeof(float) <= 160 * 1024, MASIC_ERR_SHAPE, "gdn1_fwd: bad dimension (C <= 640)");
    const double ped = 0x1p-36;
    const int HW = H * W;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gdn_generic, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(gdn_generic, dim3(ceil_div(HW, 64), B), dim3(64), (size_t)C * 64 * sizeof(float), (hipStream_t)stream,
                       x, beta, gamma, y, C, HW, inverse, (float)__builtin_sqrt(beta_min + ped), (float)__builtin_sqrt(ped), (float)ped, 1);
    return masic_launch_status("gdn1_fwd");
}

// GDN of a 128-channel float32 NCHW tensor with the result written as F16K bf16 [B][8][H*W][16] -- the input layout of
// masic_conv_f16k_fwd; the bf16-operand forward's analysis / synthesis chains (MASIC.py:521-531, :544-554) go
// conv -> (float32) -> GDN -> (F16K) -> conv.
extern "C" int masic_gdn_fwd_f16k(const float* x, const float* beta, const float* gamma, void* y_f16k,
                                  int B, int C, int H, int W, int inverse, double beta_min, void* stream) {
    MASIC_REQUIRE(x && beta && gamma && y_f16k, MASIC_ERR_ARG, "gdn_fwd_f16k: null pointer");
    MASIC_REQUIRE(B > 0 && H > 0 && W > 0, MASIC_ERR_SHAPE, "gdn_fwd_f16k: non-positive dimension");
    MASIC_REQUIRE(C == 128 && (H * W) % 32 == 0, MASIC_ERR_UNSUPPORTED, "gdn_fwd_f16k: needs C = 128 and H*W %% 32 == 0");
    const double ped = 0x1p-36;
    const int HW = H * W, spi = HW / 32, nstrips = spi * B;
    const int grid = nstrips < 4 * 512 ? (nstrips + 3) / 4 : 512;
    hipLaunchKernelGGL(gdn_bf16x3_c128, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, beta, gamma, (float*)nullptr,
                       (unsigned short*)y_f16k, HW, spi, nstrips, inverse, (float)__builtin_sqrt(beta_min + ped),
                       (float)__builtin_sqrt(ped), (float)ped);
    return masic_launch_status("gdn_fwd_f16k");
}

// Parameters of a 128-channel GDN re-laid out for the fused epilogue of masic_conv_f16k_gdn_fwd: 65536 + 512 bytes.
extern "C" size_t masic_gdn_f16k_packed_bytes(void) { return 65536 + 512; }
extern "C" int masic_gdn_pack_f16k(const float* beta, const float* gamma, void* packed, int C, double beta_min, void* stream) {
    MASIC_REQUIRE(beta && gamma && packed, MASIC_ERR_ARG, "gdn_pack_f16k: null pointer");
    MASIC_REQUIRE(C == 128, MASIC_ERR_UNSUPPORTED, "gdn_pack_f16k: needs C = 128");
    const double ped = 0x1p-36;
    hipLaunchKernelGGL(gdn_pack_f16k_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, beta, gamma, (uint4*)packed,
                       (float)__builtin_sqrt(beta_min + ped), (float)__builtin_sqrt(ped), (float)ped);
    return masic_launch_status("gdn_pack_f16k");
}
