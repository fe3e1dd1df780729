// gdn_bwd.hip -- GDN / inverse-GDN backward for C = 128 in ONE kernel (gfx950), the training step's bf16-operand mode.
//
// compressai/layers/gdn.py:77-92 under autograd:  n = beta^ + gamma^ x^2,  y = x n^(-1/2)   (inverse: x n^(1/2)).  With g = dL/dy:
//     s = g n^(-1/2)               t = dL/dn = -1/2 g x n^(-3/2)       (inverse: s = g n^(1/2), t = +1/2 g x n^(-1/2))
//     u = gamma^T t                dx = s + 2 x u
//     d gamma^[i][j] = sum_px t_i x_j^2           d beta^[i] = sum_px t_i
// and the NonNegativeParametrizer rule (parametrizers.py:61-64, bound_ops.py:40-42) from gamma^/beta^ to the stored tensors.
// The unfused float32 path (autograd.py: seven kernels, ~20 passes over the activation) stays the parity path; this kernel
// reads x and g once and writes dx once.
//
// Everything is local to a pixel, so a wave owns 32 pixels x all 128 channels in MFMA accumulator layout (lane = pixel,
// 16 channels per 32-channel block) loaded straight from NCHW (128-byte segments per channel row):
//   n  = gamma^ x^2   : v_mfma_f32_32x32x16_bf16, B operand = x^2 from the lane's own registers (k-step s covers the channels
//                       32(s>>1) + 16(s&1) + 8(c>>2) + 4h + (c&3) -- the 8 registers [s>>1][8(s&1)+c]), A = gamma^ fragments in LDS
//   u  = gamma^T t    : the same with a transposed fragment image and B = t
//   d gamma^ += t x^2^T: k = pixel.  The 4 waves of a workgroup write their t and x^2 (bf16) into [channel][128 px] LDS tiles
//                       (pitch 272 B: a ds_read_b128 of 16 channels x 8 pixels is conflict-free); wave w then contracts
//                       channels 32w..32w+31 of t against all 128 of x^2 -- 4 persistent accumulators; d beta^ is a fifth
//                       with B = ones.
// Workgroups are persistent (one per CU, 135 KiB of LDS); their partial d gamma^ / d beta^ go to a workspace and a second
// kernel adds them in a fixed order and applies the parametrizer rule.
//
// F16K operands (masic_gdn_bwd_fused_ex; the fused transforms of the training step keep their activations in that layout,
// masic_amd/autograd.py: AnalysisFn / SynthesisFn): the accumulator layout of a wave IS the tile layout of conv_f16k's epilogue, so
//   x, g  may come as F16K bf16 (8-byte loads of 4 channels, no conversion pass in front of this kernel),
//   dx    may ALSO be written as F16K bf16 (what the input-gradient convolution of the producing layer reads; whole 32-byte
//         records through v_permlane32_swap) next to -- or instead of -- the float32 NCHW tensor the weight gradient reads,
//   and the per-channel sums of dx (the bias gradient of the convolution in front of the GDN) ride along: per-lane float32
//   sums over the workgroup's tiles, reduced across lanes and waves once at the end, one more row of the partial slot.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));

constexpr int GB_NBLK = 256;                    // persistent workgroups (= partial slots)
constexpr int GB_IMG = 32768;                   // one fragment image: [m 4][s 8][lane 64] x 16 B
constexpr int GB_TP = 272;                      // LDS tile pitch in bytes: 128 px x 2 B + 16
constexpr int GB_TILE = 128 * GB_TP;            // 34816
constexpr int GB_LDS = 2 * GB_IMG + 2 * GB_TILE + 512;

__device__ __forceinline__ int gb_chan(int s, int h, int c) { return 32 * (s >> 1) + 16 * (s & 1) + 8 * (c >> 2) + 4 * h + (c & 3); }

// img[0]: A[m = i][k = j] = gamma^[i][j] (n = gamma^ x^2);  img[1]: A[m = j][k = i] = gamma^[i][j] (u = gamma^T t);  then beta^[128]
__global__ __launch_bounds__(256) void gdn_bwd_pack_kernel(const float* __restrict__ beta, const float* __restrict__ gamma,
                                                           uint4* __restrict__ img, float beta_bound, float gamma_bound, float pedestal) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 2 * 2048) {
        const int which = idx >> 11, f = idx & 2047, l = f & 63, ms = f >> 6, m = ms >> 3, s = ms & 7, r = l & 31, hh = l >> 5;
        gbf16x8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int row = 32 * m + r, ch = gb_chan(s, hh, c);
            const float gr = fmaxf(which ? gamma[(size_t)ch * 128 + row] : gamma[(size_t)row * 128 + ch], gamma_bound);
            v[c] = (__bf16)__fsub_rn(__fmul_rn(gr, gr), pedestal);
        }
        img[idx] = __builtin_bit_cast(uint4, v);
    }
    if (idx < 128) {
        const float bv = fmaxf(beta[idx], beta_bound);
        reinterpret_cast<float*>(img + 2 * 2048)[idx] = __fsub_rn(__fmul_rn(bv, bv), pedestal);
    }
}

constexpr int GB_SLOT = 128 * 128 + 256;        // floats per partial slot: d gamma^ | d beta^ | channel sums of dx

struct GdnBwdArgs {
    const float* x; const float* g; float* gx;
    const unsigned short* x16;   // X16: x as F16K [B][8][HW][16] bf16 (then x is unused)
    const unsigned short* g16;   // G16: g as F16K
    unsigned short* gx16;        // dx as F16K too, or null (gx may then be null)
    unsigned short* gxb;         // dx as bf16 NCHW (what the bf16-input 5x5 stride-2 weight-gradient kernel reads), or null
    const uint4* img;            // two fragment images + beta^
    float* part;                 // [GB_NBLK][GB_SLOT]
    long long npix;              // B * HW
    int HW, ntiles, inverse, want_sum;
    int dbg;                     // timing-only ablations (MASIC_GDNB_DBG): 1 no stores, 2 no third contraction, 4 no loads, 8 no first two contractions
};

__device__ __forceinline__ unsigned gb_pack2bf(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// lane (j, h) of a wave holds channels 32m + 8q + 4h + i (registers [m][4q + i]) of its pixel: 4 consecutive bf16 of record
// 2m + (q >> 1), elements 8(q & 1) + 4h ...  rec: address of the pixel in record 0 (+ 4h elements); rs: elements between records
__device__ __forceinline__ void gb_load_f16k(f32x16 (&v)[4], const unsigned short* rec, size_t rs, bool ok) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint2 r = make_uint2(0u, 0u);
            if (ok) r = *reinterpret_cast<const uint2*>(rec + (size_t)(2 * m + (q >> 1)) * rs + 8 * (q & 1));
            v[m][4 * q + 0] = __builtin_bit_cast(float, r.x << 16);
            v[m][4 * q + 1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
            v[m][4 * q + 2] = __builtin_bit_cast(float, r.y << 16);
            v[m][4 * q + 3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
        }
}

template <bool X16, bool G16>
__global__ __launch_bounds__(256, 1) void gdn_bwd_c128(const GdnBwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* tl = lds + 2 * GB_IMG;            // t tile   [128 ch][272 B]
    unsigned char* xl = tl + GB_TILE;                // x^2 tile
    const float* bet = reinterpret_cast<const float*>(xl + GB_TILE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    for (int i = tid; i < (2 * GB_IMG + 512) / 16; i += 256) {
        const int dst = i < 2 * GB_IMG / 16 ? i * 16 : 2 * GB_IMG + 2 * GB_TILE + (i - 2 * GB_IMG / 16) * 16;
        *reinterpret_cast<uint4*>(lds + dst) = a.img[i];
    }
    __syncthreads();

    f32x16 dg[4], db, dsum[4];                       // dsum: this lane's share of the per-channel sums of dx
#pragma unroll
    for (int e = 0; e < 16; ++e) { dg[0][e] = 0.0f; dg[1][e] = 0.0f; dg[2][e] = 0.0f; dg[3][e] = 0.0f; db[e] = 0.0f; }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) dsum[m][e] = 0.0f;
    gbf16x8 ones;
#pragma unroll
    for (int c = 0; c < 8; ++c) ones[c] = (__bf16)1.0f;

    const unsigned char* gi0 = lds + lane * 16;
    const unsigned char* gi1 = lds + GB_IMG + lane * 16;
    const unsigned hw = (unsigned)a.HW;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long long p = (long long)tile * 128 + w * 32 + j;
        const bool ok = p < a.npix;
        const long long b = ok ? p / a.HW : 0;
        const size_t pix = (size_t)(ok ? p - b * a.HW : 0);
        const size_t base = (size_t)b * 128 * hw + pix + (size_t)(4 * h) * hw;
        const size_t base16 = ((size_t)b * 8 * hw + pix) * 16 + 4 * h;       // F16K: record 0 of the pixel, + 4h elements
        const size_t rs16 = (size_t)hw * 16;
        f32x16 xv[4], gv[4];
        if constexpr (X16) gb_load_f16k(xv, a.x16 + base16, rs16, ok && !(a.dbg & 4));
        if constexpr (G16) gb_load_f16k(gv, a.g16 + base16, rs16, ok && !(a.dbg & 4));
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned off = (unsigned)(32 * m + (e & 3) + 8 * (e >> 2)) * hw;
                if constexpr (!X16) xv[m][e] = ok ? a.x[base + off] : 0.0f;
                if constexpr (!G16) gv[m][e] = ok ? a.g[base + off] : 0.0f;
            }

        // ---- n = beta^ + gamma^ x^2; x^2 (bf16) also goes to its LDS tile
        f32x16 nv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 be = *reinterpret_cast<const float4*>(bet + m * 32 + 8 * q + 4 * h);
                nv[m][4 * q] = be.x; nv[m][4 * q + 1] = be.y; nv[m][4 * q + 2] = be.z; nv[m][4 * q + 3] = be.w;
            }
        __syncthreads();                                           // the previous tile's d gamma^ reads are done
        unsigned char* xw = xl + (4 * h) * GB_TP + (w * 32 + j) * 2;
        unsigned char* tw = tl + (4 * h) * GB_TP + (w * 32 + j) * 2;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            gbf16x8 bq;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float xe = xv[s >> 1][8 * (s & 1) + c];
                bq[c] = (__bf16)__fmul_rn(xe, xe);
                *reinterpret_cast<__bf16*>(xw + (32 * (s >> 1) + 16 * (s & 1) + 8 * (c >> 2) + (c & 3)) * GB_TP) = bq[c];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const gbf16x8 ga = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(gi0 + (m * 8 + s) * 1024));
                if (!(a.dbg & 8)) nv[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bq, nv[m], 0, 0, 0);
            }
        }
        // ---- s (kept in nv) and t (kept in gv)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float r = __builtin_amdgcn_rsqf(nv[m][e]), gg = gv[m][e], gxr = gg * xv[m][e] * r;
                if (a.inverse) { nv[m][e] = gg * __builtin_amdgcn_sqrtf(nv[m][e]); gv[m][e] = 0.5f * gxr; }
                else { nv[m][e] = gg * r; gv[m][e] = -0.5f * gxr * r * r; }
            }
        // ---- u = gamma^T t; t (bf16) also goes to its LDS tile
        f32x16 uv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) uv[m][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            gbf16x8 bq;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                bq[c] = (__bf16)gv[s >> 1][8 * (s & 1) + c];
                *reinterpret_cast<__bf16*>(tw + (32 * (s >> 1) + 16 * (s & 1) + 8 * (c >> 2) + (c & 3)) * GB_TP) = bq[c];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const gbf16x8 ga = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(gi1 + (m * 8 + s) * 1024));
                if (!(a.dbg & 8)) uv[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bq, uv[m], 0, 0, 0);
            }
        }
        // ---- dx = s + 2 x u (kept in uv); rows past the end contribute zeros (x = g = 0 there)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                uv[m][e] = fmaf(2.0f * xv[m][e], uv[m][e], nv[m][e]);
                dsum[m][e] += uv[m][e];
            }
        if (ok && a.gx != nullptr && !(a.dbg & 1)) {
            float* op = a.gx + base;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) op[(unsigned)(32 * m + (e & 3) + 8 * (e >> 2)) * hw] = uv[m][e];
        }
        if (ok && a.gxb != nullptr && !(a.dbg & 1)) {
            unsigned short* ob = a.gxb + base;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const __bf16 bv = (__bf16)uv[m][e];
                    ob[(unsigned)(32 * m + (e & 3) + 8 * (e >> 2)) * hw] = __builtin_bit_cast(unsigned short, bv);
                }
        }
        if (a.gx16 != nullptr) {
            // as conv_f16k.hip: store_f16k_tile -- the two halves of the wave swap their middle quarters so that lane (j, h) writes
            // the 8 consecutive channels 8h .. 8h+7 of each 16-channel record with one 16-byte store (every lane takes part in
            // the swap; only valid pixels store)
            unsigned short* rec = a.gx16 + ((size_t)b * 8 * hw + pix) * 16 + 8 * h;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const unsigned a0 = gb_pack2bf(uv[m][8 * r + 0], uv[m][8 * r + 1]), a1 = gb_pack2bf(uv[m][8 * r + 2], uv[m][8 * r + 3]);
                    const unsigned b0 = gb_pack2bf(uv[m][8 * r + 4], uv[m][8 * r + 5]), b1 = gb_pack2bf(uv[m][8 * r + 6], uv[m][8 * r + 7]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    uint4 st;
                    st.x = s0[0]; st.y = s1[0]; st.z = s0[1]; st.w = s1[1];
                    if (ok && !(a.dbg & 1)) *reinterpret_cast<uint4*>(rec + (size_t)(2 * m + r) * rs16) = st;
                }
        }
        __syncthreads();                                           // both tiles complete
        // ---- d gamma^[32w + .][.] += t x^2^T over the 128 pixels, d beta^ with B = ones
        const unsigned char* ta = tl + (32 * w + j) * GB_TP + 16 * h;
        const unsigned char* xb = xl + j * GB_TP + 16 * h;
        if (!(a.dbg & 2))
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const gbf16x8 af = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(ta + 32 * ks));
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const gbf16x8 bf = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(xb + m * 32 * GB_TP + 32 * ks));
                dg[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, dg[m], 0, 0, 0);
            }
            db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, db, 0, 0, 0);
        }
    }

    float* pp = a.part + (size_t)blockIdx.x * GB_SLOT;
    if (a.want_sum) {
        // channel sums of dx: lanes of one half hold the same channels for 32 different pixels -> butterfly over j, then the four
        // waves through LDS (the t tile is free now), in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(tl);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = dsum[m][e];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (j == 0) red[w * 128 + 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h] = v;
            }
        __syncthreads();
        if (tid < 128) pp[128 * 128 + 128 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = 32 * w + (e & 3) + 8 * (e >> 2) + 4 * h;
            pp[i * 128 + 32 * m + j] = dg[m][e];
        }
    if (j == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) pp[128 * 128 + 32 * w + (e & 3) + 8 * (e >> 2) + 4 * h] = db[e];
    }
}

// ------------------------------------------------------------------------------------------ eight-wave form (F16K operands)
// The four-wave kernel above holds 400 registers per lane -- one wave per SIMD -- and is bound by that single wave's instruction
// issue: with no loads, no stores and no MFMAs at all it still takes 60 % of its time (tools/bench_gdn_bwd.py: MASIC_GDNB_DBG=15),
// a quarter of it 2-byte LDS stores of the [channel][pixel] tiles.  Here a workgroup is 8 waves = 4 pixel groups x 2 channel halves
// (two waves per SIMD, 64 channels of 32 pixels per wave: half the registers), and ONE pair of LDS tiles serves all three
// contractions: x^2 and t are written as F16K-like records [16-channel block s][pixel][16] (16 bytes per lane and k-step: the 8
// values of the lane's MFMA B fragment, in fragment order), so
//   n = gamma^ x^2 and u = gamma^T t read their B fragments back from the records (one ds_read_b128 per k-step; the other channel
//     half comes from the partner wave through the same tile), and
//   d gamma^ += t x^2^T contracts over the pixels of the records with transposed LDS reads (ds_read_b64_tr_b16, as wgrad_f16k.hip):
//     wave w owns the 32 x 64 block (rows 32 (w & 3), columns 64 (w >> 2)) of d gamma^ -- in RECORD order, mapped back to channels
//     when the partials are stored (position 8 h + c of a 16-channel block is channel 8 (c >> 2) + 4 h + (c & 3)).
constexpr int G8_PLANE = 128 * 32 + 128;             // [128 px][16 ch] bf16 records + the bank offset between planes
constexpr int G8_TILE = 8 * G8_PLANE;                // 33792
constexpr int G8_LDS = 2 * GB_IMG + 2 * G8_TILE + 512;

typedef unsigned g8v2u __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void g8_tr_read(g8v2u& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ gbf16x8 g8_frag(const g8v2u& lo, const g8v2u& hi) {
    return __builtin_bit_cast(gbf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}
__device__ __forceinline__ int g8_pos_chan(int q) { return 8 * ((q & 7) >> 2) + 4 * (q >> 3) + (q & 3); }   // record position -> channel of the block
__device__ __forceinline__ void g8_barrier() {       // LDS traffic of this wave done, then the workgroup barrier (no vmcnt: stores / loads stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// NML: 32-channel blocks per wave -- 2: eight waves (4 pixel groups x 2 channel halves), 1: sixteen waves (x 4 channel quarters, four
// waves per SIMD at <= 128 registers)
template <int NML>
__global__ __launch_bounds__(1024 / NML, 1) void gdn_bwd_c128_wn(const GdnBwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* tl = lds + 2 * GB_IMG;            // t records
    unsigned char* xl = tl + G8_TILE;                // x^2 records
    const float* bet = reinterpret_cast<const float*>(xl + G8_TILE);
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = w & 3, cq = w >> 2, m0 = NML * cq;          // this wave: pixels 32 pg .., channel blocks m0 .. m0 + NML - 1
    const int j = lane & 31, h = lane >> 5;

    for (int i = tid; i < (2 * GB_IMG + 512) / 16; i += 1024 / NML) {
        const int dst = i < 2 * GB_IMG / 16 ? i * 16 : 2 * GB_IMG + 2 * G8_TILE + (i - 2 * GB_IMG / 16) * 16;
        *reinterpret_cast<uint4*>(lds + dst) = a.img[i];
    }
    __syncthreads();

    f32x16 dg[NML], db, dsum[NML];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        db[e] = 0.0f;
#pragma unroll
        for (int n = 0; n < NML; ++n) { dg[n][e] = 0.0f; dsum[n][e] = 0.0f; }
    }
    gbf16x8 ones;
#pragma unroll
    for (int c = 0; c < 8; ++c) ones[c] = (__bf16)1.0f;

    const unsigned char* gi0 = lds + lane * 16;
    const unsigned char* gi1 = lds + GB_IMG + lane * 16;
    const unsigned hw = (unsigned)a.HW;
    const int px = 32 * pg + j;
    const int g4 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned lane_off = (g4 & 1) * G8_PLANE + ((8 * (g4 >> 1) + qq) * 32) + 8 * pp;
    const unsigned la = ldsb + 2 * GB_IMG + (2 * pg) * G8_PLANE + lane_off;                         // t rows: 32-channel block pg
    const unsigned lb = ldsb + 2 * GB_IMG + G8_TILE + (2 * m0) * G8_PLANE + lane_off;               // x^2 columns: blocks m0 .. m0 + NML - 1

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long long p = (long long)tile * 128 + px;
        const bool ok = p < a.npix;
        const long long b = ok ? p / a.HW : 0;
        const size_t pix = (size_t)(ok ? p - b * a.HW : 0);
        const size_t base = (size_t)b * 128 * hw + pix + (size_t)(4 * h) * hw;
        const size_t rs16 = (size_t)hw * 16;
        const unsigned short* xr = a.x16 + ((size_t)b * 8 * hw + pix) * 16 + 4 * h;
        const unsigned short* gr = a.g16 + ((size_t)b * 8 * hw + pix) * 16 + 4 * h;
        f32x16 xv[NML], gv[NML];
#pragma unroll
        for (int ml = 0; ml < NML; ++ml)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t ro = (size_t)(2 * (m0 + ml) + (q >> 1)) * rs16 + 8 * (q & 1);
                uint2 rx = make_uint2(0u, 0u), rg = make_uint2(0u, 0u);
                if (ok && !(a.dbg & 4)) { rx = *reinterpret_cast<const uint2*>(xr + ro); rg = *reinterpret_cast<const uint2*>(gr + ro); }
                xv[ml][4 * q + 0] = __builtin_bit_cast(float, rx.x << 16);
                xv[ml][4 * q + 1] = __builtin_bit_cast(float, rx.x & 0xffff0000u);
                xv[ml][4 * q + 2] = __builtin_bit_cast(float, rx.y << 16);
                xv[ml][4 * q + 3] = __builtin_bit_cast(float, rx.y & 0xffff0000u);
                gv[ml][4 * q + 0] = __builtin_bit_cast(float, rg.x << 16);
                gv[ml][4 * q + 1] = __builtin_bit_cast(float, rg.x & 0xffff0000u);
                gv[ml][4 * q + 2] = __builtin_bit_cast(float, rg.y << 16);
                gv[ml][4 * q + 3] = __builtin_bit_cast(float, rg.y & 0xffff0000u);
            }
        // ---- x^2 records of this wave's 16-channel blocks (k-steps 2 m0 .. 2 m0 + 2 NML - 1)
#pragma unroll
        for (int sl = 0; sl < 2 * NML; ++sl) {
            gbf16x8 bq;
#pragma unroll
            for (int c = 0; c < 8; ++c) { const float xe = xv[sl >> 1][8 * (sl & 1) + c]; bq[c] = (__bf16)__fmul_rn(xe, xe); }
            *reinterpret_cast<uint4*>(xl + (2 * m0 + sl) * G8_PLANE + px * 32 + 16 * h) = __builtin_bit_cast(uint4, bq);
        }
        g8_barrier();                                              // (A) all x^2 records of the tile are in place
        // ---- n = beta^ + gamma^ x^2 for this wave's two channel blocks
        f32x16 nv[NML];
#pragma unroll
        for (int ml = 0; ml < NML; ++ml)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 be = *reinterpret_cast<const float4*>(bet + (m0 + ml) * 32 + 8 * q + 4 * h);
                nv[ml][4 * q] = be.x; nv[ml][4 * q + 1] = be.y; nv[ml][4 * q + 2] = be.z; nv[ml][4 * q + 3] = be.w;
            }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const gbf16x8 bq = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(xl + s * G8_PLANE + px * 32 + 16 * h));
#pragma unroll
            for (int ml = 0; ml < NML; ++ml) {
                const gbf16x8 ga = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(gi0 + (((m0 + ml) * 8 + s) * 1024)));
                nv[ml] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bq, nv[ml], 0, 0, 0);
            }
        }
        // ---- s (kept in nv) and t (kept in gv)
#pragma unroll
        for (int ml = 0; ml < NML; ++ml)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float r = __builtin_amdgcn_rsqf(nv[ml][e]), gg = gv[ml][e], gxr = gg * xv[ml][e] * r;
                if (a.inverse) { nv[ml][e] = gg * __builtin_amdgcn_sqrtf(nv[ml][e]); gv[ml][e] = 0.5f * gxr; }
                else { nv[ml][e] = gg * r; gv[ml][e] = -0.5f * gxr * r * r; }
            }
#pragma unroll
        for (int sl = 0; sl < 2 * NML; ++sl) {
            gbf16x8 bq;
#pragma unroll
            for (int c = 0; c < 8; ++c) bq[c] = (__bf16)gv[sl >> 1][8 * (sl & 1) + c];
            *reinterpret_cast<uint4*>(tl + (2 * m0 + sl) * G8_PLANE + px * 32 + 16 * h) = __builtin_bit_cast(uint4, bq);
        }
        g8_barrier();                                              // (B) all t records of the tile are in place
        // ---- u = gamma^T t
        f32x16 uv[NML];
#pragma unroll
        for (int ml = 0; ml < NML; ++ml)
#pragma unroll
            for (int e = 0; e < 16; ++e) uv[ml][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const gbf16x8 bq = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(tl + s * G8_PLANE + px * 32 + 16 * h));
#pragma unroll
            for (int ml = 0; ml < NML; ++ml) {
                const gbf16x8 ga = __builtin_bit_cast(gbf16x8, *reinterpret_cast<const uint4*>(gi1 + (((m0 + ml) * 8 + s) * 1024)));
                uv[ml] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bq, uv[ml], 0, 0, 0);
            }
        }
        // ---- dx = s + 2 x u; stores
#pragma unroll
        for (int ml = 0; ml < NML; ++ml)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                uv[ml][e] = fmaf(2.0f * xv[ml][e], uv[ml][e], nv[ml][e]);
                dsum[ml][e] += uv[ml][e];
            }
        if (ok && a.gx != nullptr && !(a.dbg & 1)) {
            float* op = a.gx + base;
#pragma unroll
            for (int ml = 0; ml < NML; ++ml)
#pragma unroll
                for (int e = 0; e < 16; ++e) op[(unsigned)(32 * (m0 + ml) + (e & 3) + 8 * (e >> 2)) * hw] = uv[ml][e];
        }
        if (ok && a.gxb != nullptr && !(a.dbg & 1)) {
            unsigned short* ob = a.gxb + base;
#pragma unroll
            for (int ml = 0; ml < NML; ++ml)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const __bf16 bv = (__bf16)uv[ml][e];
                    ob[(unsigned)(32 * (m0 + ml) + (e & 3) + 8 * (e >> 2)) * hw] = __builtin_bit_cast(unsigned short, bv);
                }
        }
        if (a.gx16 != nullptr) {
            unsigned short* rec = a.gx16 + ((size_t)b * 8 * hw + pix) * 16 + 8 * h;
#pragma unroll
            for (int ml = 0; ml < NML; ++ml)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const unsigned a0 = gb_pack2bf(uv[ml][8 * r + 0], uv[ml][8 * r + 1]), a1 = gb_pack2bf(uv[ml][8 * r + 2], uv[ml][8 * r + 3]);
                    const unsigned b0 = gb_pack2bf(uv[ml][8 * r + 4], uv[ml][8 * r + 5]), b1 = gb_pack2bf(uv[ml][8 * r + 6], uv[ml][8 * r + 7]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    uint4 st;
                    st.x = s0[0]; st.y = s1[0]; st.z = s0[1]; st.w = s1[1];
                    if (ok && !(a.dbg & 1)) *reinterpret_cast<uint4*>(rec + (size_t)(2 * (m0 + ml) + r) * rs16) = st;
                }
        }
        // ---- d gamma^ block (rows: t block pg, columns: x^2 blocks m0 .. m0 + NML - 1) over the 128 pixels; d beta^ with B = ones
        if (!(a.dbg & 2))
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            g8v2u af[2], bf[NML][2];
            // (template offsets need constants: the k-step loop is unrolled)
            switch (ks) {
#define G8_KS(K)                                                                                     \
                case K:                                                                              \
                    g8_tr_read<K * 16 * 32>(af[0], la); g8_tr_read<K * 16 * 32 + 128>(af[1], la);   \
                    g8_tr_read<K * 16 * 32>(bf[0][0], lb); g8_tr_read<K * 16 * 32 + 128>(bf[0][1], lb);                                 \
                    if constexpr (NML == 2) {                                                                                            \
                        g8_tr_read<2 * G8_PLANE + K * 16 * 32>(bf[NML - 1][0], lb); g8_tr_read<2 * G8_PLANE + K * 16 * 32 + 128>(bf[NML - 1][1], lb); \
                    }                                                                                                                    \
                    break;
                G8_KS(0) G8_KS(1) G8_KS(2) G8_KS(3) G8_KS(4) G8_KS(5) G8_KS(6) G8_KS(7)
#undef G8_KS
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(af[0]), "+v"(af[1]));
#pragma unroll
            for (int n = 0; n < NML; ++n) asm volatile("" : "+v"(bf[n][0]), "+v"(bf[n][1]));
            const gbf16x8 fa = g8_frag(af[0], af[1]);
#pragma unroll
            for (int n = 0; n < NML; ++n) dg[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, g8_frag(bf[n][0], bf[n][1]), dg[n], 0, 0, 0);
            if (cq == 0) db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, ones, db, 0, 0, 0);
        }
        g8_barrier();                                              // (C) the records have been read: the next tile may overwrite them
    }

    float* ppart = a.part + (size_t)blockIdx.x * GB_SLOT;
    if (a.want_sum) {
        float* red = reinterpret_cast<float*>(tl);
#pragma unroll
        for (int ml = 0; ml < NML; ++ml)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = dsum[ml][e];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (j == 0) red[pg * 128 + 32 * (m0 + ml) + (e & 3) + 8 * (e >> 2) + 4 * h] = v;
            }
        __syncthreads();
        if (tid < 128) ppart[128 * 128 + 128 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    }
    // accumulator (row r = 8 (e >> 2) + 4 h + (e & 3), column n = lane & 31) of block (pg, m0 + nb): record positions -> channels
#pragma unroll
    for (int nb = 0; nb < NML; ++nb) {
        const int jc = 32 * (m0 + nb) + 16 * (j >> 4) + g8_pos_chan(j & 15);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = 8 * (e >> 2) + 4 * h + (e & 3);
            const int ic = 32 * pg + 16 * (r >> 4) + g8_pos_chan(r & 15);
            ppart[ic * 128 + jc] = dg[nb][e];
        }
    }
    if (cq == 0 && j == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = 8 * (e >> 2) + 4 * h + (e & 3);
            ppart[128 * 128 + 32 * pg + 16 * (r >> 4) + g8_pos_chan(r & 15)] = db[e];
        }
    }
}

// fixed-order sum of the partials + the parametrizer's backward rule (gradient passes where the stored value is above the
// bound or the step would raise it): g = 2 max(p, bound) dL/dp^.  Block = 32 entries x 8 slices of the partial list.
__global__ __launch_bounds__(256) void gdn_bwd_reduce_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ beta,
                                                             const float* __restrict__ gamma, float* __restrict__ g_beta,
                                                             float* __restrict__ g_gamma, float* __restrict__ g_sum, float beta_bound, float gamma_bound) {
    __shared__ float red[8][32];
    const int l = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + l;                      // 128*128 + 128 (+ 128 with the dx sums) is a multiple of 32
    float s = 0.0f;
    for (int k = sl; k < nblk; k += 8) s += part[(size_t)k * GB_SLOT + idx];
    red[sl][l] = s;
    __syncthreads();
    if (sl != 0) return;
    for (int k = 1; k < 8; ++k) s += red[k][l];
    if (idx >= 128 * 128 + 128) { g_sum[idx - (128 * 128 + 128)] = s; return; }      // plain sums: no parametrizer in front of a bias
    const bool isg = idx < 128 * 128;
    const float p = isg ? gamma[idx] : beta[idx - 128 * 128], bound = isg ? gamma_bound : beta_bound;
    const float gr = s * 2.0f * fmaxf(p, bound);
    const float r = (p >= bound || gr < 0.0f) ? gr : 0.0f;
    if (isg) g_gamma[idx] = r; else g_beta[idx - 128 * 128] = r;
}

}  // namespace

extern "C" size_t masic_gdn_bwd_fused_workspace_bytes(void) {
    return (size_t)2 * GB_IMG + 512 + (size_t)GB_NBLK * GB_SLOT * sizeof(float);
}

// x / x_f16k, g / g_f16k: exactly one of each (float32 NCHW or F16K bf16 [B][8][HW][16]); gx / gx_f16k: at least one; g_sum [128] or
// NULL: per-channel sums of dx over batch and pixels (the bias gradient of the convolution whose output the GDN normalises).
extern "C" int masic_gdn_bwd_fused_ex(const float* x, const void* x_f16k, const float* g, const void* g_f16k, const float* beta, const float* gamma,
                                      float* gx, void* gx_f16k, float* g_sum, float* g_beta, float* g_gamma, void* workspace,
                                      int B, int C, int H, int W, int inverse, double beta_min, void* stream) {
    return masic_gdn_bwd_fused_ex2(x, x_f16k, g, g_f16k, beta, gamma, gx, gx_f16k, nullptr, g_sum, g_beta, g_gamma, workspace, B, C, H, W, inverse, beta_min, stream);
}

// ... and gx_bf16: dx as bf16 NCHW [B][128][H][W] (or NULL): the operand of masic_conv2d_wgrad_bf16in, half the bytes of gx
extern "C" int masic_gdn_bwd_fused_ex2(const float* x, const void* x_f16k, const float* g, const void* g_f16k, const float* beta, const float* gamma,
                                       float* gx, void* gx_f16k, void* gx_bf16, float* g_sum, float* g_beta, float* g_gamma, void* workspace,
                                       int B, int C, int H, int W, int inverse, double beta_min, void* stream) {
    MASIC_REQUIRE(((x != nullptr) != (x_f16k != nullptr)) && ((g != nullptr) != (g_f16k != nullptr)) && (gx || gx_f16k || gx_bf16), MASIC_ERR_ARG,
                  "gdn_bwd_fused: exactly one of x / x_f16k and of g / g_f16k, at least one of gx / gx_f16k / gx_bf16");
    MASIC_REQUIRE(beta && gamma && g_beta && g_gamma && workspace, MASIC_ERR_ARG, "gdn_bwd_fused: null pointer");
    MASIC_REQUIRE(C == 128, MASIC_ERR_UNSUPPORTED, "gdn_bwd_fused: C=%d (128 only; other widths use the unfused pieces)", C);
    MASIC_REQUIRE(B > 0 && H > 0 && W > 0 && (long long)H * W * 128 < (1ll << 31), MASIC_ERR_SHAPE, "gdn_bwd_fused: shape");
    hipStream_t st = (hipStream_t)stream;
    const double ped = 1.4551915228366852e-11;      // (2^-18)^2, parametrizers.py:47-56
    const float pedestal = (float)ped, beta_bound = (float)__builtin_sqrt(beta_min + ped), gamma_bound = (float)__builtin_sqrt(ped);
    uint4* img = (uint4*)workspace;
    float* part = reinterpret_cast<float*>((unsigned char*)workspace + 2 * GB_IMG + 512);
    hipLaunchKernelGGL(gdn_bwd_pack_kernel, dim3(16), dim3(256), 0, st, beta, gamma, img, beta_bound, gamma_bound, pedestal);
    GdnBwdArgs a{};
    a.x = x; a.g = g; a.gx = gx; a.img = img; a.part = part;
    a.x16 = (const unsigned short*)x_f16k; a.g16 = (const unsigned short*)g_f16k; a.gx16 = (unsigned short*)gx_f16k; a.gxb = (unsigned short*)gx_bf16;
    a.HW = H * W; a.npix = (long long)B * H * W; a.inverse = inverse; a.want_sum = g_sum != nullptr;
    a.ntiles = (int)((a.npix + 127) / 128);
    { const char* e = getenv("MASIC_GDNB_DBG"); a.dbg = e ? atoi(e) : 0; }
    const int nblk = a.ntiles < GB_NBLK ? a.ntiles : GB_NBLK;
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[4] = {(const void*)gdn_bwd_c128<false, false>, (const void*)gdn_bwd_c128<true, false>, (const void*)gdn_bwd_c128<false, true>,
                              (const void*)gdn_bwd_c128<true, true>};
        for (const void* fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS) != hipSuccess) {
                masic_set_error("gdn_bwd_fused: cannot reserve %d bytes of LDS", GB_LDS);
                return MASIC_ERR_LAUNCH;
            }
        attr_set = true;
    }
    static const bool w8 = [] { const char* e = getenv("MASIC_GDNB_W8"); return e == nullptr || e[0] != '0'; }();
    if (x_f16k && g_f16k && w8) {
        static bool attr8 = false;
        if (!attr8) {
            if (hipFuncSetAttribute((const void*)gdn_bwd_c128_wn<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)gdn_bwd_c128_wn<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) != hipSuccess) {
                masic_set_error("gdn_bwd_fused: cannot reserve %d bytes of LDS", G8_LDS);
                return MASIC_ERR_LAUNCH;
            }
            attr8 = true;
        }
        static const int waves = getenv("MASIC_GDNB_WAVES") ? atoi(getenv("MASIC_GDNB_WAVES")) : 8;
        if (waves == 16) hipLaunchKernelGGL(gdn_bwd_c128_wn<1>, dim3(nblk), dim3(1024), G8_LDS, st, a);
        else hipLaunchKernelGGL(gdn_bwd_c128_wn<2>, dim3(nblk), dim3(512), G8_LDS, st, a);
    } else if (x_f16k && g_f16k) hipLaunchKernelGGL((gdn_bwd_c128<true, true>), dim3(nblk), dim3(256), GB_LDS, st, a);
    else if (x_f16k) hipLaunchKernelGGL((gdn_bwd_c128<true, false>), dim3(nblk), dim3(256), GB_LDS, st, a);
    else if (g_f16k) hipLaunchKernelGGL((gdn_bwd_c128<false, true>), dim3(nblk), dim3(256), GB_LDS, st, a);
    else hipLaunchKernelGGL((gdn_bwd_c128<false, false>), dim3(nblk), dim3(256), GB_LDS, st, a);
    hipLaunchKernelGGL(gdn_bwd_reduce_kernel, dim3((128 * 128 + 128 + (g_sum ? 128 : 0)) / 32), dim3(256), 0, st, (const float*)part, nblk,
                       beta, gamma, g_beta, g_gamma, g_sum, beta_bound, gamma_bound);
    return masic_launch_status("gdn_bwd_fused");
}

extern "C" int masic_gdn_bwd_fused(const float* x, const float* g, const float* beta, const float* gamma, float* gx,
                                   float* g_beta, float* g_gamma, void* workspace, int B, int C, int H, int W, int inverse,
                                   double beta_min, void* stream) {
    MASIC_REQUIRE(x && g && gx, MASIC_ERR_ARG, "gdn_bwd_fused: null pointer");
    return masic_gdn_bwd_fused_ex(x, nullptr, g, nullptr, beta, gamma, gx, nullptr, nullptr, g_beta, g_gamma, workspace, B, C, H, W, inverse, beta_min, stream);
}
