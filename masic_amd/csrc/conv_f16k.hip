// conv_f16k.hip -- bf16 implicit-GEMM convolutions on channel-blocked activations (gfx950 / MI355X).
//
// Same layers and the same phase geometry as conv.hip (reference: coremasic/mywork/MASIC.py:510-622 analysis /
// synthesis transforms, :170-187 and :690-700 hyper transforms; factories compressai/models/utils.py:128-146), for the
// bf16-operand forward.  conv.hip's bf16 kernel reads float32 NCHW and converts while staging, which makes every
// workgroup re-convert its halo and leaves the matrix cores waiting on VGPR staging and on L2 latency of the weight
// fragments.  Here both operands reach LDS by DMA (global_load_lds, 16 bytes per lane, no VGPR round trip):
//
//   activations  F16K  [B][C/16][H*W][16] bf16 (gemm_bf16.hip): a pixel's 16 channels are one 32-byte record, so the
//                patch of a tile for one 16-channel block is rows of contiguous records and an MFMA B fragment
//                (8 consecutive k of one pixel) is one ds_read_b128;
//   weights      [phase-tap][ci/16][co][16] bf16 (pack_weight_bf16_kernel): one (tap, 16-channel) slab of a
//                128-channel block is 4 KiB contiguous, an A fragment is one ds_read_b128.
//
// Workgroup = 8 waves (2 along co x 4 along pixels), tile 128 co x 256 pixels, wave tile 64 x 64 (2 x 2 MFMA
// 32x32x16 accumulators).  K runs over chunks of KS 16-channel blocks; within a chunk over "steps" of T taps.
// Every step each wave issues the same number of DMA wave-instructions -- WI for the weight slab of step g+D
// (D+1-deep ring) and PS for a slice of the patch of chunk c+L (L+1 buffers) -- so that completion can be awaited
// with a counted `s_waitcnt vmcnt((D-1)*(WI+PS))` (never 0 inside the loop) followed by one raw s_barrier per step:
// DMA stays in flight across barriers.  Unused issue slots go to a dummy 1 KiB LDS region to keep the count fixed.
//
// LDS images are XOR-swizzled in 16-byte slots, slot' = slot ^ ((slot >> 4) & 3): the stride-2 pixel walk of a
// strided conv (64 B between lanes), the stride-1 walk (32 B) and the weight rows (32 B) all become conflict-free for
// ds_read_b128's four 16-lane groups.  The swizzle is applied on the DMA side by permuting which record a lane fetches.
//
// Epilogue: bias + activation, then either float32 NCHW (channel view of a concat buffer, optional gate) or F16K bf16
// for the next layer.
#include <stdlib.h>

#include "common.h"
#include <type_traits>
#include "conv_geom.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// LDS reads / waits the compiler does not see (see the K loop of conv_f16k)
template <int OFF>
__device__ __forceinline__ void ds_read128(v4u& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(v4u& a, v4u& b, v4u& c, v4u& d, v4u& e) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(v4u& a, v4u& b, v4u& c, v4u& d, v4u& e, v4u& f) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(v4u& a, v4u& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(v4u& a, v4u& b, v4u& c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(v4u& a, v4u& b, v4u& c, v4u& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(v4u& a, v4u& b, v4u& c, v4u& d, v4u& e, v4u& f, v4u& g, v4u& h) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(v4u& a, v4u& b, v4u& c, v4u& d, v4u& e, v4u& f, v4u& g, v4u& h, v4u& i, v4u& j) {
    asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i), "+v"(j) : "n"(N) : "memory");
}

// ---- fp8 (OCP e4m3fn) operands: v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 127) = a plain fp8 MFMA at
// twice the bf16 rate (MI355X_MICROARCH.md, Matrix cores).  A lane's operand is 32 consecutive k (two ds_read_b128 that the
// register coalescer places in one 8-register tuple); lanes 0-31 / 32-63 hold the two 32-k halves of the instruction's K = 64.
typedef int v8i __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v8i cat8(const v4u& lo, const v4u& hi) {
    return __builtin_bit_cast(v8i, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ f32x16 mfma_f8(const v8i& a, const v8i& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
}
__device__ __forceinline__ unsigned pack4f8(float a, float b, float c, float d, float inv) {
    // saturating: e4m3fn has no infinity, its largest finite value is 448
    a = __builtin_amdgcn_fmed3f(a * inv, -448.0f, 448.0f);
    b = __builtin_amdgcn_fmed3f(b * inv, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c * inv, -448.0f, 448.0f);
    d = __builtin_amdgcn_fmed3f(d * inv, -448.0f, 448.0f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void dma16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int slot) { return slot ^ ((slot >> 4) & 3); }

// F16K store of the 32 channels of accumulator tile `t` (lane (j, h) holds channels 4h + 8q + i, q, i < 4, of pixel j):
// the two halves of the wave swap their middle quarters (v_permlane32_swap) so that lane (j, h) ends up with the 8
// consecutive channels 8h .. 8h+7 of each 16-channel record and writes it with one 16-byte store -- the wave writes
// whole 32-byte records (full sectors; 1 KiB contiguous when the pixels are adjacent) instead of scattered 8-byte pieces.
// rec: address of the lane's pixel in the first of the two records, + 8h elements; rec_stride: elements between records.
__device__ __forceinline__ unsigned pack2bf(float lo, float hi);
// residual add in float32 BEFORE the rounding to bf16: lane (j, h) holds channels 8q + 4h + i of tile t, i.e. 4 consecutive
// bf16 (8 bytes) at element offset 8(q & 1) + 4h of record q >> 1 of the residual's tile.  res: address of the lane's pixel in
// the first record of the tile (element units), + 4h.
__device__ __forceinline__ void add_f16k_residual(f32x16& t, const unsigned short* res, unsigned rec_stride) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint2 r = *reinterpret_cast<const uint2*>(res + (size_t)(q >> 1) * rec_stride + 8 * (q & 1));
        t[4 * q + 0] += __builtin_bit_cast(float, r.x << 16);
        t[4 * q + 1] += __builtin_bit_cast(float, r.x & 0xffff0000u);
        t[4 * q + 2] += __builtin_bit_cast(float, r.y << 16);
        t[4 * q + 3] += __builtin_bit_cast(float, r.y & 0xffff0000u);
    }
}
// t *= act'(mask): the backward of ReLU / LeakyReLU(0.01) as an epilogue of the input-gradient convolution -- mask = the forward
// activation's OUTPUT in F16K (its sign is the pre-activation's), addressed like a residual tensor
__device__ __forceinline__ void mul_f16k_actmask(f32x16& t, const unsigned short* mask, unsigned rec_stride, float slope) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint2 r = *reinterpret_cast<const uint2*>(mask + (size_t)(q >> 1) * rec_stride + 8 * (q & 1));
        t[4 * q + 0] *= __builtin_bit_cast(float, r.x << 16) > 0.0f ? 1.0f : slope;
        t[4 * q + 1] *= __builtin_bit_cast(float, r.x & 0xffff0000u) > 0.0f ? 1.0f : slope;
        t[4 * q + 2] *= __builtin_bit_cast(float, r.y << 16) > 0.0f ? 1.0f : slope;
        t[4 * q + 3] *= __builtin_bit_cast(float, r.y & 0xffff0000u) > 0.0f ? 1.0f : slope;
    }
}
__device__ __forceinline__ void store_f16k_tile(const f32x16& t, unsigned short* rec, unsigned rec_stride) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned a0 = pack2bf(t[8 * r + 0], t[8 * r + 1]), a1 = pack2bf(t[8 * r + 2], t[8 * r + 3]);   // channels 4h .. 4h+3
        const unsigned b0 = pack2bf(t[8 * r + 4], t[8 * r + 5]), b1 = pack2bf(t[8 * r + 6], t[8 * r + 7]);   // channels 8 + 4h ..
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        uint4 st;
        st.x = s0[0]; st.y = s1[0]; st.z = s0[1]; st.w = s1[1];
        *reinterpret_cast<uint4*>(rec + (size_t)r * rec_stride) = st;
    }
}

// the same through a buffer resource: voff = byte offset of the lane's pixel in the first record + 16h, or any offset >= the resource's
// size for a lane that has nothing to store (dropped by the range check) -- the instruction is issued unconditionally, so the number
// of store instructions a wave has in flight is known at compile time (counted s_waitcnt vmcnt in conv_a_gdn_f16k)
template <int AUX = 0>
__device__ __forceinline__ void store_f16k_tile_buf(const f32x16& t, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned rec_stride_bytes) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned a0 = pack2bf(t[8 * r + 0], t[8 * r + 1]), a1 = pack2bf(t[8 * r + 2], t[8 * r + 3]);
        const unsigned b0 = pack2bf(t[8 * r + 4], t[8 * r + 5]), b1 = pack2bf(t[8 * r + 6], t[8 * r + 7]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        v4u st = {s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(st, rs, (int)(voff + r * rec_stride_bytes), 0, AUX);
    }
}

// F8K store (fp8 activations [B][C/32][H*W][32]: a pixel's 32 channels are one 32-byte record) of accumulator tile `t` = one
// whole record per pixel: lane (j, h) holds channels 8q + 4h + i; after two v_permlane32_swap it holds the 16 consecutive
// channels 16h .. 16h+15 and writes them with one 16-byte store.  rec: address of the lane's pixel record + 16h bytes.
__device__ __forceinline__ unsigned pack4f8(float a, float b, float c, float d, float inv);
__device__ __forceinline__ void store_f8k_tile(const f32x16& t, unsigned char* rec, float inv) {
    unsigned q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = pack4f8(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3], inv);
    const auto s0 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);     // h = 0: channels 0-3, 4-7;   h = 1: 16-19, 20-23
    const auto s1 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);     // h = 0: 8-11, 12-15;         h = 1: 24-27, 28-31
    uint4 st;
    st.x = s0[0]; st.y = s0[1]; st.z = s1[0]; st.w = s1[1];
    *reinterpret_cast<uint4*>(rec) = st;
}

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

struct F16kArgs {
    const unsigned short* x;      // F16K [B][in_c16tot][Hi*Wi][16]
    const unsigned short* w;      // slab stream, see pack_f16k_stream_kernel
    const float* bias;
    const float* gate;            // float32 [B][gate_ctot][Ho][Wo] or null (NCHW output only)
    float* y32;                   // float32 NCHW view, or null
    unsigned short* y16;          // F16K [B][out_c16tot][Ho*Wo][16], or null
    const uint4* gdn_img;         // GDN epilogue: gamma^ fragments (gdn.hip: gdn_pack_f16k_kernel), then beta^[128] floats
    const unsigned short* res1;   // F16K output only: up to two residual tensors [B][res_ctot/16][Ho*Wo][16] added after the activation
    const unsigned short* res2;   //   (ResidualBlock / Enhancement_Block identities, compressai/layers/layers.py:189, MASIC.py:163)
    int res_ctot;
    unsigned short* y16_pre;      // fused GDN + F16K output: also store the convolution's result BEFORE the GDN (same view), or null --
                                  //   what the GDN backward of a training step needs (masic_amd/autograd.py: AnalysisFn / SynthesisFn);
                                  //   residual form: the activation's output BEFORE the residual adds (view of res_ctot channels)
    unsigned long long* stamps;   // diagnostics (masic_conv_f16k_set_stamps): {s_memtime, s_memrealtime} at 4 points of the first and last workgroup
    const unsigned short* mask16; // F16K output only: multiply by act'(mask) before the residual adds (mask: F16K of res_ctot channels;
    float mask_slope;             //   slope for mask <= 0: 0.01 LeakyReLU, 0 ReLU) -- input gradients of the training step
    const float* res32;           // float32 NCHW output only: residual [B][cout_store][Ho][Wo] added after the activation, or null
    int cout_store;               // float32 NCHW output: channels actually stored (< Cout when the weight was zero-padded to a multiple of 32)
    const float* wscale;          // fp8 operands: per-output-channel dequantisation factor (weight scale x input scale), else null
    unsigned char* y8;            // F8K [B][out_c32tot][Ho*Wo][32] fp8 output (quantised with out_inv_scale), or null
    float out_inv_scale;
    int gdn_inverse;
    int d2s;                      // > 0: channel 4c + phase (2x2 phases, c < d2s); y32 is [B][out_ctot][2Ho][2Wo] (depth-to-space store)
    int in_c16tot, in_c16off, Cin16;
    int Hi, Wi, Cout, Ho, Wo;
    int out_ctot, out_coff;       // channel view of the output (NCHW: channels; F16K: channels, multiples of 16)
    int gate_ctot, gate_c, act;
    int TW, TWlog, SR, TH, tiles_w, ntiles;
    int PH, PW, PWh, NPIXp;       // patch rows / row length (pixels), ceil(PW/2), records per k-half plane of the LDS image (x32)
    int PB;                       // bytes per patch buffer
    unsigned mg_PW, mg_PWh, mg_gpk;   // ceil(2^32 / d): x / d == __umulhi(x, magic) for the small x, d of the patch maps
    unsigned phase_off[4];        // byte offset of each phase's slab streams
    unsigned stream_bytes[4];     // bytes of one (phase, co-block) slab stream
    GeomParams q;
    int nphase;
    int xcd_images;               // B % 8 == 0: image b runs on XCD b % 8 (the halos of its tiles meet in one L2)
    int msplit;                   // workgroups per 128-channel co-block (4 / NM for layers whose grid would not fill the chip, else 1):
                                  //   blockIdx.y = co-block * msplit + sub-block, a sub-block = NM accumulator tiles of the slab
    int exp_store;                // timing experiments only (MASIC_F16K_STORE_EXP; results are WRONG when set): 1 the F16K stores of the epilogue are
                                  //   skipped at run time (the arithmetic stays), 2 a transposed layer's phases store to phase-planar positions
                                  //   (pixel (r + oph Ho/2, c + opw Wo/2): every wave writes whole contiguous records, as if one workgroup owned both x-phases)
};

#ifndef F16K_DMA_SPREAD
#define F16K_DMA_SPREAD 1  // 0: the K loop's DMA pieces issued en bloc at the top of every step (A/B timing, tools/ablate_f16k.sh)
#endif
#ifndef F16K_ABLATE
#define F16K_ABLATE 0     // timing experiments only (tools/ablate_f16k.sh): 1 no DMA in the K loop, 2 no barriers, 3 no fragment reads, 4 a quarter of the MFMAs
#endif
constexpr int MAXTAPS = 64;   // tap table entries ([step][T rounded up to a power of two])

// Weights for this path, in the order the kernel consumes them ("slab stream"): for each phase, each 128-channel
// co-block, each chunk c of KS 16-channel blocks, each step t of T taps (taps padded with zeros to a multiple of T):
// T*KS slabs of 4 KiB, each the LDS image [k-half hh][co 128][8 ci] of W[co, (c*KS+ks)*16 + hh*8 + e, tap].
// A step's slab group is one contiguous WST = T*KS*4096 bytes, so the DMA of step g is base + g*WST.
struct PackStreamArgs {
    const float* w;
    int Cin, Cout, KH, KW, transposed;
    int KS, T, nchunks, ncb;
    ConvGeom gs[4];           // one launch packs every phase (blockIdx.y)
    unsigned phase_offs[4];   // bytes
};

__device__ __forceinline__ void pack_f16k_stream_body(const PackStreamArgs& a, unsigned short* __restrict__ wp, int phase, int bx, int nbx) {
    const ConvGeom g = a.gs[phase];
    const unsigned phase_off = a.phase_offs[phase];
    const int spc = (g.ntaps + a.T - 1) / a.T;
    const size_t total = (size_t)a.ncb * a.nchunks * spc * a.T * a.KS * 2048;        // bf16 elements
    for (size_t i = (size_t)bx * blockDim.x + threadIdx.x; i < total; i += (size_t)nbx * blockDim.x) {
        size_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int co = (int)(r & 127); r >>= 7;
        const int hh = (int)(r & 1); r >>= 1;
        const int ks = (int)(r % a.KS); r /= a.KS;
        const int tt = (int)(r % a.T); r /= a.T;
        const int t = (int)(r % spc); r /= spc;
        const int c = (int)(r % a.nchunks); r /= a.nchunks;
        const int cb = (int)r;
        const int tap = t * a.T + tt;
        const int ci = (c * a.KS + ks) * 16 + hh * 8 + e, cog = cb * 128 + co;
        float v = 0.0f;
        if (tap < g.ntaps && ci < a.Cin && cog < a.Cout) {
            const int ta = tap / g.ntw, tb = tap - ta * g.ntw;
            const int kh = g.kh0 + ta * g.khs, kw = g.kw0 + tb * g.kws;
            const size_t src = a.transposed ? (((size_t)ci * a.Cout + cog) * a.KH + kh) * a.KW + kw
                                            : (((size_t)cog * a.Cin + ci) * a.KH + kh) * a.KW + kw;
            v = a.w[src];
        }
        const __bf16 bv = (__bf16)v;
        wp[(phase_off >> 1) + i] = __builtin_bit_cast(unsigned short, bv);
    }
}

__global__ void pack_f16k_stream_kernel(const PackStreamArgs a, unsigned short* __restrict__ wp) {
    pack_f16k_stream_body(a, wp, blockIdx.y, blockIdx.x, gridDim.x);
}

// Many layers' streams in one launch (a training step re-packs every weight it touches -- each for its forward and for its input
// gradient -- ~50 launches of ~5 us whose work is a few microseconds): blockIdx.y = job, blockIdx.z = phase; a job is what one
// masic_conv_f16k_pack_weight call would launch.  The job table lives in device memory (built once per set of layers).
struct PackStreamJob { PackStreamArgs a; unsigned short* wp; int nb, np; };

__global__ void pack_f16k_stream_multi_kernel(const PackStreamJob* __restrict__ jobs) {
    const PackStreamJob& j = jobs[blockIdx.y];
    if ((int)blockIdx.z >= j.np || (int)blockIdx.x >= j.nb) return;
    pack_f16k_stream_body(j.a, j.wp, blockIdx.z, blockIdx.x, j.nb);
}

// fp8 operands: the same stream with 32-channel blocks -- slab (tap, block) = LDS image [k-half hh][co 128][16 ci] of
// fp8(W[co, (c*KS+ks)*32 + hh*16 + e, tap] / wscale[co]); wscale[co] = max |W[co]| / 448 (per-output-channel scaling).
__global__ void wscale_f8k_kernel(const float* __restrict__ w, float* __restrict__ ws, int Cin, int Cout, int taps, int transposed) {
    const int co = blockIdx.x;
    float m = 0.0f;
    for (int i = threadIdx.x; i < Cin * taps; i += blockDim.x) {
        const int ci = i / taps, t = i - ci * taps;
        m = fmaxf(m, fabsf(transposed ? w[((size_t)ci * Cout + co) * taps + t] : w[((size_t)co * Cin + ci) * taps + t]));
    }
    __shared__ float red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) ws[co] = red[0] > 0.0f ? red[0] / 448.0f : 1.0f;
}

__global__ void pack_f8k_stream_kernel(const PackStreamArgs a, const float* __restrict__ ws, unsigned char* __restrict__ wp) {
    const ConvGeom g = a.gs[blockIdx.y];
    const unsigned phase_off = a.phase_offs[blockIdx.y];
    const int spc = (g.ntaps + a.T - 1) / a.T;
    const size_t total = (size_t)a.ncb * a.nchunks * spc * a.T * a.KS * 4096;        // fp8 elements = bytes
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = (int)(r & 15); r >>= 4;
        const int co = (int)(r & 127); r >>= 7;
        const int hh = (int)(r & 1); r >>= 1;
        const int ks = (int)(r % a.KS); r /= a.KS;
        const int tt = (int)(r % a.T); r /= a.T;
        const int t = (int)(r % spc); r /= spc;
        const int c = (int)(r % a.nchunks); r /= a.nchunks;
        const int cb = (int)r;
        const int tap = t * a.T + tt;
        const int ci = (c * a.KS + ks) * 32 + hh * 16 + e, cog = cb * 128 + co;
        float v = 0.0f;
        if (tap < g.ntaps && ci < a.Cin && cog < a.Cout) {
            const int ta = tap / g.ntw, tb = tap - ta * g.ntw;
            const int kh = g.kh0 + ta * g.khs, kw = g.kw0 + tb * g.kws;
            const size_t src = a.transposed ? (((size_t)ci * a.Cout + cog) * a.KH + kh) * a.KW + kw
                                            : (((size_t)cog * a.Cin + ci) * a.KH + kh) * a.KW + kw;
            v = a.w[src] / ws[cog];
        }
        wp[phase_off + i] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.0f, 0, false) & 0xff);
    }
}

__device__ __forceinline__ void dma_buf16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// (Inverse) GDN over the 128 channels a wave holds for its 32 pixels (accumulator layout of four 32x32 MFMA tiles):
//   y_i = x_i * rsqrt(beta^_i + sum_j gamma^_ij x_j^2)  (inverse: * sqrt), compressai/layers/gdn.py:77-92.
// The 128 x 128 contraction runs on the matrix cores with gamma^ and x^2 each split into bf16 hi + lo (three products:
// error ~2^-16 of a term).  k-step s covers channels 32(s>>1) + 16(s&1) + 8(c>>2) + 4hh + (c&3), c = 0..7 -- exactly the
// 8 accumulator registers acc[s>>1][8(s&1) + c] the lane already holds for its pixel, so the B operand needs no data
// movement; the gamma^ fragments lie in LDS pre-arranged in that order (gdn.hip: gdn_pack_f16k_kernel).
// gimg: LDS address of the fragment image + lane * 16;  bet: LDS address of beta^[128] + 4h (read late: no registers held).
// X3 = false: one product (bf16 gamma^ x bf16 x^2, error ~2^-10 of the result -- the order of the bf16 rounding the result gets
// anyway when it is stored as F16K / fed to a bf16-operand layer): a third of the MFMAs, half the fragment reads, no lo split.
template <bool X3>
__device__ __forceinline__ void gdn_in_registers(f32x16 (&acc)[4], const unsigned char* gimg, const float* bet, int inverse) {
    f32x16 nrm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) nrm[m][e] = 0.0f;
#pragma unroll
    for (int sx = 0; sx < 8; ++sx) {
        bf16x8 bh, blo;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const float xv = acc[sx >> 1][8 * (sx & 1) + cc];
            const float sq = __fmul_rn(xv, xv);
            const __bf16 hi = (__bf16)sq;
            bh[cc] = hi;
            if constexpr (X3) blo[cc] = (__bf16)(sq - (float)hi);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bf16x8 gh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(gimg + ((m * 8 + sx) * 2 + 0) * 1024));
            if constexpr (X3) {
                const bf16x8 gl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(gimg + ((m * 8 + sx) * 2 + 1) * 1024));
                nrm[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, bh, nrm[m], 0, 0, 0);
                nrm[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, blo, nrm[m], 0, 0, 0);
            }
            nrm[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, bh, nrm[m], 0, 0, 0);
        }
    }
    // v_sqrt_f32 / v_rsq_f32 (1 ulp): the result is rounded to bf16 or feeds a bf16-operand convolution anyway.  `inverse` is uniform:
    // a BRANCH, not a select -- as `inverse ? sqrt : rsq` the compiler evaluated both for every element (128 quarter-rate instructions
    // and 64 selects per lane where 64 + 0 do)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 be = *reinterpret_cast<const float4*>(bet + m * 32 + 8 * q);
            nrm[m][4 * q + 0] += be.x; nrm[m][4 * q + 1] += be.y; nrm[m][4 * q + 2] += be.z; nrm[m][4 * q + 3] += be.w;
        }
    if (inverse) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] *= __builtin_amdgcn_sqrtf(nrm[m][e]);
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] *= __builtin_amdgcn_rsqf(nrm[m][e]);
    }
}

// KS 16-channel blocks per chunk, T taps per step, D weight-slab look-ahead (steps), PSP patch DMA wave-instructions per
// patch wave per step during the first 2 steps of a chunk, L patch look-ahead (chunks); GDN: (inverse) GDN over the 128
// output channels fused into the epilogue.  Output: float32 NCHW if a.y32 else F16K.
// NM: 32-channel accumulator tiles per wave -- 4 (a 128-channel block) or 1 (layers with <= 32 output channels).
// NP: 32-pixel sub-tiles per wave (tile = 256 NP pixels): larger tiles for the large stride-1-walk layers, whose
// per-workgroup fixed cost (launch, first DMA, epilogue) would otherwise rival their K loop.
// F8: fp8 (e4m3) operands -- the input is F8K (32-channel records), a slab is (tap, 32-channel block) and one MFMA (K = 64)
// consumes two consecutive slabs of the step, one per lane half: with KS = 2 the two channel blocks of a tap, with KS = 1
// two consecutive taps.  DMA, LDS images, ring and geometry are the bf16 kernel's (a record is 32 bytes in both layouts).
template <int KS, int T, int D, int PSP, int L, bool GDN, int NM, int NP, bool F8 = false>
__global__ __launch_bounds__(512, D == 1 ? 2 : 1) void conv_f16k(const F16kArgs a) {     // D == 1: the two-workgroups-per-CU ("lite") configurations
    static_assert(!GDN || NM == 4, "the fused GDN needs all 128 channels");
    static_assert(!F8 || (NM == 4 && NP == 1 && (T * KS) % 2 == 0 && T <= 4), "fp8 operands: 128-channel blocks, 256-pixel tiles, slab pairs");
    static_assert(NM + NP == 2 || NM + NP == 3 || NM + NP == 4 || NM + NP == 5 || NM + NP == 6, "fragment-wait helpers exist for 2, 4, 5 and 6 fragments per k-step");
    constexpr int WI = T * KS;                   // weight DMA wave-instructions per weight wave per step (4 waves x 1 KiB x WI = slab group)
    constexpr int NWS = D + 1;                   // weight ring slots
    constexpr int WST = T * KS * 4096;           // bytes per step of weights
    constexpr int NPI = PSP * 2;                 // patch DMA wave-instructions per patch wave per chunk (incl. dummies)
    constexpr int NB = L + 1;                    // patch buffers
    constexpr int PATCH0 = NWS * WST;            // LDS byte offset of the patch buffers
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int dummy_off = PATCH0 + NB * a.PB;    // 1 KiB sink for unused DMA slots, then the tap table
    const int table_off = dummy_off + 1024;
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;   // LDS byte address of lds[0]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    // DMA roles: waves 0-3 stream the weight slabs, waves 4-7 the activation patch.  vmcnt is per wave and retires in
    // order, so the roles are what keeps a long-latency patch load (HBM) from gating the wait for the next weight slab (L2).
    const bool wrole = wave < 4;
    const int wq = wave & 3;

    // block -> (tile, phase): the phases of a tile sit 8 block ids apart, i.e. on the same XCD and next to each other in
    // time, so their interleaved output pixels meet in one L2
    // With a multiple of 8 images, image b = 8 slot + k is walked tile by tile on XCD k (workgroup ids go round-robin over the 8
    // XCDs and gridDim.x is a multiple of 8): the 5x5 halos that neighbouring tiles share are then re-read from that XCD's L2
    // instead of HBM (measured on the 128 -> 128 stride-2 layers: 142 -> 129 MB per launch against 105 algorithmic).
    int phase, tile, b;
    if (a.xcd_images) {
        const int idx = (blockIdx.x >> 3) + (gridDim.x >> 3) * blockIdx.z;
        const int slot = idx / (int)gridDim.x, rest = idx - slot * (int)gridDim.x;
        b = slot * 8 + (blockIdx.x & 7);
        phase = rest % a.nphase;
        tile = rest / a.nphase;
    } else {
        const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
        phase = rest % a.nphase;
        tile = (rest / a.nphase) * 8 + xcd;
        b = blockIdx.z;
    }
    if (tile >= a.ntiles) return;
    if (F16K_ABLATE == 6) return;                 // launch cost only
    // diagnostics: core-clock and 100 MHz constant-clock stamps of wave 0 of the first and of the last workgroup
    unsigned long long* const stamp = (a.stamps != nullptr && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x == 0 || blockIdx.x + 1 == gridDim.x))
                                          ? a.stamps + (blockIdx.x == 0 ? 0 : 8) : nullptr;
    if (stamp) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[1] = __builtin_amdgcn_s_memrealtime(); }
    const int tw_i = tile % a.tiles_w, th_i = tile / a.tiles_w;
    const int r0 = th_i * a.TH, c0 = tw_i * a.TW;
    const int nchunks = a.Cin16 / KS;
    const int gpk = a.NPIXp >> 5;                             // DMA wave-instructions per 16-channel plane of the patch (2*NPIXp/64)
    const int plane_bytes = a.Hi * a.Wi * 32;                 // bytes per 16-channel plane of the input
    int SPC;                                                  // steps per chunk
    int bl[NP];                                               // lane part of the B-fragment addresses
    int goff[NPI];                                            // patch waves: byte offset of this lane's record in a chunk

    // Buffer resources: out-of-range offsets read as zero, so padding pixels (voffset = huge), chunks past the last one and
    // slab groups past the end of the stream need no special case, and the per-chunk / per-step offset rides in an SGPR.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + ((size_t)b * a.in_c16tot + a.in_c16off) * (size_t)(a.Hi * a.Wi * 16)), 0, a.Cin16 * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned char*)a.w + a.phase_off[phase] + (size_t)(blockIdx.y / a.msplit) * a.stream_bytes[phase]), 0, (int)a.stream_bytes[phase], 0x00020000);
    const int msub = (int)(blockIdx.y % a.msplit) * NM;       // first 32-channel tile of this workgroup inside the 128-channel slab
    if ((int)(blockIdx.y / a.msplit) * 128 + msub * 32 >= a.Cout) return;      // a sub-block past the last output channel (Cout = 192: second co-block)

    {   // ---- geometry-dependent setup (kept in a scope: nothing of the phase geometry stays live in the K loop)
        const ConvGeom g = make_geom(a.q, phase);
        SPC = (g.ntaps + T - 1) / T;
        const int ih0 = r0 * g.is + g.dh_min, iw0 = c0 * g.is + g.dw_min;
        const int ninstr = KS * gpk;
        // LDS images (16-byte records, the two k-halves of a 32-byte record in separate planes so that the 16 lanes of a
        // ds_read_b128 group read 256 contiguous bytes):
        //   weight slab (tap, ks): [hh][co 128]   (the packed stream already has this order: a linear copy)
        //   patch plane (ks):      [hh][pixel'] with pixel' = row * PW + col for stride-1 walks and, for strided convs,
        //                          columns de-interleaved by parity, pixel' = ((col & 1) * PH + row) * PWh + (col >> 1),
        //                          so that the stride-2 walk of a tap reads consecutive records too.
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
            const int I = k * 4 + wq;
            const int ks = (int)__umulhi((unsigned)I, a.mg_gpk), grp = I - ks * gpk;
            const int q = grp * 64 + lane;
            const int hh = q >= a.NPIXp ? 1 : 0, pp = q - hh * a.NPIXp;
            int pr, pc;
            bool ok = I < ninstr;
            if (g.is == 2) {
                const int half = a.PH * a.PWh;
                const int par = pp >= half ? 1 : 0, rem = pp - par * half;
                pr = (int)__umulhi((unsigned)rem, a.mg_PWh);
                pc = 2 * (rem - pr * a.PWh) + par;
                ok = ok && pp < 2 * half && pc < a.PW;
            } else {
                pr = (int)__umulhi((unsigned)pp, a.mg_PW);
                pc = pp - pr * a.PW;
                ok = ok && pp < a.PH * a.PW;
            }
            const int ih = ih0 + pr, iw = iw0 + pc;
            ok = ok && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
            goff[k] = ok ? ks * plane_bytes + (ih * a.Wi + iw) * 32 + hh * 16 : 0x7ffffff0;
        }
        const int PWe = g.is == 2 ? a.PWh : a.PW;             // row pitch (records) of the LDS patch image
        const int jr = j >> a.TWlog, jc = j & (a.TW - 1);
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            if constexpr (F8) bl[n] = ((((wave * NP + n) * a.SR + jr) * g.is) * PWe + jc) * 16 + (KS == 2 ? h * gpk * 1024 : 0);   // both k-half planes are read; lane half = channel block (KS = 2) or tap (KS = 1)
            else bl[n] = (h * a.NPIXp + (((wave * NP + n) * a.SR + jr) * g.is) * PWe + jc) * 16;
        }
        // tap table: byte offset of each tap's record inside the LDS patch image (padding taps alias tap 0; their weights are zero)
        if (tid < MAXTAPS) {                                   // [step][TP] entries, TP = T rounded up to a power of two
            constexpr int TP = T == 5 ? 8 : T;
            const int tidx = (tid / TP) * T + (tid % TP);
            const int tap = (tid % TP) < T && tidx < g.ntaps ? tidx : 0;
            const int ti = tap / g.ntw, tj = tap - ti * g.ntw;
            const int ta = (g.dh0 - g.dh_min) + ti * g.dsh, tb = (g.dw0 - g.dw_min) + tj * g.dsw;
            const int slot = g.is == 2 ? ((tb & 1) * a.PH + ta) * a.PWh + (tb >> 1) : ta * a.PW + tb;
            reinterpret_cast<int*>(lds + table_off)[tid] = slot * 16;
        }
    }
    const int ninstr = KS * gpk;
    const int wvoff = wq * 1024 + lane * 16;                  // weight waves: this lane's record inside a 4 KiB slab
    const int al = F8 ? j * 16 + h * 4096 : (h * 128 + j) * 16;   // lane part of the A-fragment address (fp8: lane half = odd / even slab of the pair)
    int pdst[NPI];                                            // patch waves: LDS offset inside a patch buffer (or the sink)
#pragma unroll
    for (int k = 0; k < NPI; ++k) pdst[k] = (k * 4 + wq) < ninstr ? (k * 4 + wq) * 1024 : -1;

    // wave w owns all 128 output channels of pixel sub-tile w (4 accumulator tiles): the GDN epilogue needs every channel
    // of a pixel, and this way it finds them in the wave's own registers
    f32x16 acc[NP][NM];
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][m][e] = 0.0f;

    // ---- prologue: weight slab groups of the first D steps, the first L patch chunks
    int wsoff = 0;                                            // weight producer: byte offset of the next slab group in the stream
    if (wrole) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int k = 0; k < WI; ++k) dma_buf16(rw, lds + d * WST + k * 4096 + wq * 1024, wvoff, wsoff + k * 4096);
            wsoff += WST;
        }
    } else {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < NPI; ++k)
                dma_buf16(rx, lds + (pdst[k] >= 0 ? PATCH0 + l * a.PB + pdst[k] : dummy_off), goff[k], l * (KS * plane_bytes));
    }
    // the accumulators start at the bias (its 16 ... 64 scalar loads per lane fly with the prologue's DMA) instead of at zero with 64 adds
    // behind a round trip to global memory in the epilogue; fp8 operands keep the add there (the dequantisation scale comes first)
    if constexpr (!F8) {
        if (a.bias != nullptr) {
            const float* bp = a.bias + (blockIdx.y / a.msplit) * 128 + msub * 32 + 4 * h;
            const int m0i = (blockIdx.y / a.msplit) * 128 + msub * 32;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (m0i + m * 32 < a.Cout) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float bv = bp[m * 32 + (e & 3) + 8 * (e >> 2)];
#pragma unroll
                        for (int n = 0; n < NP; ++n) acc[n][m][e] = bv;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (F16K_ABLATE == 7) return;                 // launch + prologue (setup, first DMA, wait)
    if (stamp) { stamp[2] = __builtin_amdgcn_s_memtime(); stamp[3] = __builtin_amdgcn_s_memrealtime(); }
    int cslot = 0, pslot = (D % NWS) * WST;                   // byte offsets of the consumer / producer ring slots
    int cb = PATCH0, pb = PATCH0 + (L % NB) * a.PB;           // byte offsets of the patch buffer of chunk c / of chunk c+L
    int xsoff = L * (KS * plane_bytes);                       // patch producer: byte offset of chunk c+L in the input

    v4u tvq;                                                  // tap offsets (LDS bytes inside a patch buffer) of the next step
    unsigned tv4q = 0;
    auto table_request = [&](int t) {
        if constexpr (T == 5) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(tvq) : "v"(ldsb + table_off + t * 32) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(tv4q) : "v"(ldsb + table_off + t * 32) : "memory");
        } else if constexpr (T == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(tvq) : "v"(ldsb + table_off + t * 16) : "memory");
        else asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<v2u*>(&tvq)) : "v"(ldsb + table_off + t * 8) : "memory");
    };
    table_request(0);

    // One step: slice S of the patch DMA (or none), T taps x KS k-steps of 2x2 MFMAs, counted wait, barrier.
    auto step = [&](auto slice, int t, bool last) {
        constexpr int S = decltype(slice)::value;
        // The step's DMA pieces -- weight waves: the slab group of step g+D; patch waves: slice S of chunk c+L -- are issued one or
        // two at a time AFTER the MFMAs of each k-step, in the shadow of the matrix pipe, instead of en bloc at the top of the step
        // (60-185 cycles apiece there, MI355X_MICROARCH.md).  Worth ~2 % (in-kernel stamps, tools/f16k_stamps.py: 47.0 -> 46.1 core
        // clocks per MFMA per SIMD in the K loop of the 128 -> 128 stride-2 layers): the K loop is not DMA-issue-bound -- see
        // DESIGN.md section 10 for what bounds it.  F16K_DMA_SPREAD=0 keeps the en-bloc issue (A/B timing).
        auto dma_pieces = [&](int first, int last) {
            if (F16K_ABLATE == 1) return;
            if (wrole) {
#pragma unroll
                for (int k = first; k < last; ++k)
                    if (k < WI) dma_buf16(rw, lds + pslot + k * 4096 + wq * 1024, wvoff, wsoff + k * 4096);
            } else if (S >= 0) {
#pragma unroll
                for (int u = first; u < last; ++u) {
                    constexpr int k = (S >= 0 ? S : 0) * PSP;
                    if (u < PSP) dma_buf16(rx, lds + (pdst[k + u] >= 0 ? pb + pdst[k + u] : dummy_off), goff[k + u], xsoff);
                }
            }
        };
        constexpr int NPIECES = WI > PSP ? WI : PSP;
        // D == 1: the slab group of the NEXT step is awaited at the end of this one, so it goes out first (its latency budget is the
        // step; the CU's other workgroup covers what is left of it)
        constexpr bool SPREAD = F16K_DMA_SPREAD != 0 && D > 1;
        if (!SPREAD) dma_pieces(0, NPIECES);
        // LDS reads of the K loop go through inline asm with hand-counted lgkmcnt: for a compiler-visible ds_read hipcc puts
        // `s_waitcnt vmcnt(0)` in front (the DMA in flight might alias it), which would drain the prefetch queue every step.
        // The fragments of k-step i+1 are requested before the MFMAs of k-step i are issued.
        // tap offsets of this step: requested a step ago (table_request), so the wait is over before it starts; the next step's
        // request goes out now and returns under this step's MFMAs (LDS returns in order: it is older than every fragment read the
        // counted waits below leave outstanding).  Read at the top of the step they cost every wave an LDS round trip right after
        // the barrier, with nothing else to issue.
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tvq), "+v"(tv4q)::"memory");
        const v4u tvv = tvq;
        const unsigned tv4 = tv4q;
        table_request(last ? 0 : t + 1);
        const unsigned wst = ldsb + al + cslot + msub * 512;
        constexpr int NQ = F8 ? 2 : 1;                            // ds_read_b128 per fragment
        constexpr int NKS = F8 ? (T * KS) / 2 : T * KS;           // MFMA k-steps per step
        v4u af[2][NM][NQ], bfr[2][NP][NQ];
        auto request = [&](auto ic, auto bc) {
            constexpr int i = decltype(ic)::value, buf = decltype(bc)::value;
            if constexpr (F8) {
                // slabs 2i (lanes 0-31) and 2i+1 (lanes 32-63) of the step; a fragment = the 16-byte records of both k-half planes
                unsigned toff = ldsb + cb;
                if constexpr (KS == 2) toff += tvv[i];                                   // same tap, the two channel blocks (lane part of bl)
                else toff += h ? tvv[2 * i + 1] : tvv[2 * i];                            // taps 2i / 2i+1
                const unsigned plane = (unsigned)a.NPIXp * 16;
                static_for<0, NP>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    ds_read128<0>(bfr[buf][n][0], toff + bl[n]);
                    ds_read128<0>(bfr[buf][n][1], toff + bl[n] + plane);
                });
                static_for<0, NM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    ds_read128<i * 8192 + m * 512>(af[buf][m][0], wst);
                    ds_read128<i * 8192 + m * 512 + 2048>(af[buf][m][1], wst);
                });
            } else {
                constexpr int tt = i / KS, ks = i % KS;
                const unsigned toff = ldsb + cb + (tt < 4 ? tvv[tt < 4 ? tt : 0] : tv4) + ks * gpk * 1024;
                static_for<0, NP>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    ds_read128<0>(bfr[buf][n][0], toff + bl[n]);
                });
                static_for<0, NM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    ds_read128<i * 4096 + m * 512>(af[buf][m][0], wst);
                });
            }
        };
        if (F16K_ABLATE != 3) request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, NKS>([&](auto ic) {
            constexpr int i = decltype(ic)::value, buf = i & 1;
            if constexpr (i + 1 < NKS && F16K_ABLATE != 3) request(std::integral_constant<int, i + 1>{}, std::integral_constant<int, (i + 1) & 1>{});
            constexpr int pending = i + 1 < NKS ? (NM + NP) * NQ : 0;     // LDS returns in order: what was requested for i+1 may stay out
            if constexpr (F8) {
                lgkm_wait<pending>(bfr[buf][0][0], bfr[buf][0][1], af[buf][0][0], af[buf][0][1], af[buf][1][0], af[buf][1][1],
                                   af[buf][2][0], af[buf][2][1], af[buf][3][0], af[buf][3][1]);
                const v8i b8 = cat8(bfr[buf][0][0], bfr[buf][0][1]);
                static_for<0, NM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    acc[0][m] = mfma_f8(cat8(af[buf][m][0], af[buf][m][1]), b8, acc[0][m]);
                });
            } else {
            if constexpr (NM == 4 && NP == 1) lgkm_wait<pending>(bfr[buf][0][0], af[buf][0][0], af[buf][1][0], af[buf][2][0], af[buf][3][0]);
            else if constexpr (NM == 4 && NP == 2) lgkm_wait<pending>(bfr[buf][0][0], bfr[buf][1][0], af[buf][0][0], af[buf][1][0], af[buf][2][0], af[buf][3][0]);
            else if constexpr (NM == 1 && NP == 4) lgkm_wait<pending>(bfr[buf][0][0], bfr[buf][1][0], bfr[buf][2][0], bfr[buf][3][0], af[buf][0][0]);
            else if constexpr (NM == 3 && NP == 2) lgkm_wait<pending>(bfr[buf][0][0], bfr[buf][1][0], af[buf][0][0], af[buf][1][0], af[buf][2][0]);
            else if constexpr (NM == 2 && NP == 2) lgkm_wait<pending>(bfr[buf][0][0], bfr[buf][1][0], af[buf][0][0], af[buf][1][0]);
            else if constexpr (NM == 2 && NP == 1) lgkm_wait<pending>(bfr[buf][0][0], af[buf][0][0], af[buf][1][0]);
            else lgkm_wait<pending>(bfr[buf][0][0], af[buf][0][0]);
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                static_for<0, NP>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    if (F16K_ABLATE != 4 || m == 0)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[buf][m][0]), __builtin_bit_cast(bf16x8, bfr[buf][n][0]), acc[n][m], 0, 0, 0);
                });
            });
            }
            if constexpr (SPREAD) {
                constexpr int per = (NPIECES + NKS - 1) / NKS;
                __builtin_amdgcn_sched_barrier(0);
                dma_pieces(i * per, (i + 1) * per);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if (wrole && F16K_ABLATE != 1) {
            wsoff += WST;
            pslot = pslot + WST == NWS * WST ? 0 : pslot + WST;
        }
        cslot = cslot + WST == NWS * WST ? 0 : cslot + WST;
        // weight waves: the slab group of the next step has landed, the D-1 groups after it stay in flight.
        // patch waves: at the end of a chunk the next chunk's patch has landed, the L-1 chunks after it stay in flight.
        if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * WI) : "memory");
        else if (last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((L - 1) * NPI) : "memory");
        if (F16K_ABLATE != 2) __builtin_amdgcn_s_barrier();
    };

    for (int c = 0; c < (F16K_ABLATE == 5 || F16K_ABLATE == 8 ? 0 : nchunks); ++c) {
        step(std::integral_constant<int, 0>{}, 0, false);
        step(std::integral_constant<int, 1>{}, 1, SPC == 2);
        for (int t = 2; t < SPC; ++t) step(std::integral_constant<int, -1>{}, t, t + 1 == SPC);
        cb = cb + a.PB == PATCH0 + NB * a.PB ? PATCH0 : cb + a.PB;
        pb = pb + a.PB == PATCH0 + NB * a.PB ? PATCH0 : pb + a.PB;
        xsoff += KS * plane_bytes;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tvq), "+v"(tv4q)::"memory");      // the last step's table request must not land in reused registers
    if (stamp) { stamp[4] = __builtin_amdgcn_s_memtime(); stamp[5] = __builtin_amdgcn_s_memrealtime(); }

    // ---- epilogue
    const ConvGeom g = make_geom(a.q, phase);
    const int m0 = (blockIdx.y / a.msplit) * 128 + msub * 32;
    const int jr = j >> a.TWlog, jc = j & (a.TW - 1);
    const size_t oplane = (size_t)a.Ho * a.Wo;
    // bias, then (inverse) GDN or the activation.  Cout is a multiple of 32, so a 32-channel block is valid or not as a whole.
    if constexpr (F8) {                      // dequantise: per-output-channel weight scale x the input tensor's scale
        const float* sp = a.wscale + m0 + 4 * h;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m0 + m * 32 < a.Cout) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float sv = sp[m * 32 + (e & 3) + 8 * (e >> 2)];
#pragma unroll
                    for (int n = 0; n < NP; ++n) acc[n][m][e] *= sv;
                }
            }
        }
    }
    if constexpr (F8) {
        if (a.bias != nullptr) {
            const float* bp = a.bias + m0 + 4 * h;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (m0 + m * 32 < a.Cout) {
                    float bv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) bv[e] = bp[m * 32 + (e & 3) + 8 * (e >> 2)];
#pragma unroll
                    for (int n = 0; n < NP; ++n)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[n][m][e] += bv[e];
                }
            }
        }
    }
    if constexpr (GDN && F16K_ABLATE != 10) {
        // The fragment image (+ beta^) is shared by the 8 waves through LDS (ring and patch buffers are free now): one DMA, one barrier.
        // Piece p = (m * 8 + k-step) * 2 + plane: the odd pieces are the low halves of the three-product split -- the one-product form
        // (the default) leaves them where they are: 32 KiB per workgroup instead of 64
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gdn_img, 0, 65536 + 512, 0x00020000);
        const bool x3 = (a.gdn_inverse & 2) != 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (x3 || (k & 1) == 0) dma_buf16(rg, lds + (wave * 8 + k) * 1024, lane * 16, (wave * 8 + k) * 1024);
        if (wave == 0) dma_buf16(rg, lds + 65536, lane * 16, 65536);          // 512 bytes of beta^; lanes >= 32 read past the end: zeros
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        if (NP > 1) __builtin_amdgcn_sched_barrier(0);      // one sub-tile's epilogue at a time (register pressure)
        const int r = r0 + (wave * NP + n) * a.SR + jr, c = c0 + jc;
        const bool pok = r < g.Hp && c < g.Wp;
        int oh = pok ? r * g.os + g.oph : 0, ow = pok ? c * g.os + g.opw : 0;
        if (a.exp_store == 2 && g.os == 2 && pok) { oh = r + g.oph * (a.Ho >> 1); ow = c + g.opw * (a.Wo >> 1); }
        const size_t opix = (size_t)oh * a.Wo + ow;
        if constexpr (GDN) {
            if (a.y16_pre != nullptr && pok) {
                const int c16 = (a.out_coff + m0) >> 4;
                unsigned short* yp = a.y16_pre + (((size_t)b * (a.out_ctot >> 4) + c16) * oplane + opix) * 16 + 8 * h;
                const unsigned op16 = (unsigned)oplane * 16;
#pragma unroll
                for (int m = 0; m < NM; ++m) store_f16k_tile(acc[n][m], yp + (size_t)(2 * m) * op16, op16);
            }
            if (F16K_ABLATE != 9 && F16K_ABLATE != 10) {
                if (a.gdn_inverse & 2) gdn_in_registers<true>(acc[n], lds + lane * 16, reinterpret_cast<const float*>(lds + 65536) + 4 * h, a.gdn_inverse & 1);
                else gdn_in_registers<false>(acc[n], lds + lane * 16, reinterpret_cast<const float*>(lds + 65536) + 4 * h, a.gdn_inverse & 1);
            }
        } else {
            float gv = 1.0f;
            if (a.gate != nullptr && pok) gv = a.gate[((size_t)b * a.gate_ctot + a.gate_c) * oplane + opix];      // (float32 NCHW or F16K output)
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[n][m][e] = apply_act(acc[n][m][e], a.act) * gv;
        }
        if (F16K_ABLATE == 8) continue;               // everything but the K loop and the stores
        // stores: one 64-bit base per lane, 32-bit channel offsets
        if (pok && a.d2s > 0) {
            // depth-to-space: channel 4c + phase of the equivalent stride-1 convolution -> pixel (2r + phase/2, 2c + phase%2) of
            // channel c.  With that channel order a lane holds whole 2x2 output blocks (channel c = h + 2q in registers 4q..4q+3):
            // two 8-byte stores per block, 256 contiguous bytes per row and half-wave.
            float* yb = a.y32 + ((size_t)b * a.out_ctot + a.out_coff) * (4 * oplane) + (size_t)(2 * oh) * (2 * a.Wo) + 2 * ow;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cc = h + 2 * q;
                if (cc < a.d2s) {
                    float* yc = yb + (size_t)cc * (4 * oplane);
                    *reinterpret_cast<float2*>(yc) = make_float2(acc[n][0][4 * q], acc[n][0][4 * q + 1]);
                    *reinterpret_cast<float2*>(yc + 2 * a.Wo) = make_float2(acc[n][0][4 * q + 2], acc[n][0][4 * q + 3]);
                }
            }
        } else if (pok) {
            if (a.y32 != nullptr && a.cout_store < a.Cout) {
                // zero-padded output channels (a 96 -> 3 layer run as 96 -> 32): store the real ones, with the float32 residual
                const int cbase = m0 + 4 * h;
                float* yb = a.y32 + ((size_t)b * a.out_ctot + a.out_coff + cbase) * oplane + opix;
                const float* rb = a.res32 != nullptr ? a.res32 + ((size_t)b * a.cout_store + cbase) * oplane + opix : nullptr;
                const unsigned op = (unsigned)oplane;
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int cc = m * 32 + (e & 3) + 8 * (e >> 2);
                        if (cbase + cc < a.cout_store) yb[(unsigned)cc * op] = acc[n][m][e] + (rb != nullptr ? rb[(unsigned)cc * op] : 0.0f);
                    }
            } else if (a.y32 != nullptr) {
                float* yb = a.y32 + ((size_t)b * a.out_ctot + a.out_coff + m0 + 4 * h) * oplane + opix;
                const unsigned op = (unsigned)oplane;
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    if (m0 + m * 32 < a.Cout) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) yb[(unsigned)(m * 32 + (e & 3) + 8 * (e >> 2)) * op] = acc[n][m][e];
                    }
            } else if (a.y8 != nullptr) {
                // fp8 output for an fp8-operand consumer: tile m = record m of this 128-channel block (out_coff, m0 multiples of 32)
                const int c32 = (a.out_coff + m0) >> 5;
                unsigned char* yb = a.y8 + (((size_t)b * (a.out_ctot >> 5) + c32) * oplane + opix) * 32 + 16 * h;
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    if (m0 + m * 32 < a.Cout) store_f8k_tile(acc[n][m], yb + (size_t)m * oplane * 32, a.out_inv_scale);
            } else {
                // tile m = records 2m, 2m+1 of this 128-channel block (out_coff and m0 are multiples of 16)
                const int c16 = (a.out_coff + m0) >> 4;
                unsigned short* yb = a.y16 + (((size_t)b * (a.out_ctot >> 4) + c16) * oplane + opix) * 16 + 8 * h;
                const unsigned op16 = (unsigned)oplane * 16;
                if (a.res1 != nullptr || a.mask16 != nullptr || (!GDN && a.y16_pre != nullptr)) {
                    const size_t ro = (((size_t)b * (a.res_ctot >> 4) + (m0 >> 4)) * oplane + opix) * 16 + 4 * h;
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        if (m0 + m * 32 < a.Cout) {
                            if (a.mask16 != nullptr) mul_f16k_actmask(acc[n][m], a.mask16 + ro + (size_t)(2 * m) * op16, op16, a.mask_slope);
                            if (!GDN && a.y16_pre != nullptr) store_f16k_tile(acc[n][m], a.y16_pre + ro + 4 * h + (size_t)(2 * m) * op16, op16);
                            if (a.res1 != nullptr) add_f16k_residual(acc[n][m], a.res1 + ro + (size_t)(2 * m) * op16, op16);
                            if (a.res2 != nullptr) add_f16k_residual(acc[n][m], a.res2 + ro + (size_t)(2 * m) * op16, op16);
                        }
                }
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    if (m0 + m * 32 < a.Cout && a.exp_store != 1) store_f16k_tile(acc[n][m], yb + (size_t)(2 * m) * op16, op16);
            }
        }
    }
    if (stamp) { stamp[6] = __builtin_amdgcn_s_memtime(); stamp[7] = __builtin_amdgcn_s_memrealtime(); }
}

// ------------------------------------------------------------------------------------------ first analysis layer
// g_a_conv1 + g_a_gdn1 (MASIC.py:515-516, :563-564): Conv2d(3 -> 128, k5, s2, p2) on the float32 NCHW image followed by
// GDN, written as F16K for the next convolution.  K = 75 (padded to 80 = 5 MFMA k-steps), so the layer is bound by its
// output (B x 128 x H/2 x W/2) and by the GDN contraction, not by the convolution: persistent workgroups (8 waves, one
// 8 x 32 output tile per iteration, wave w = row w) keep the gamma^ image (64 KiB), the weight fragments (20 KiB), bias
// and beta^ in LDS for all their tiles; the 3 x 19 x 67 input patch of the next tile is fetched into registers while the
// current one is contracted.  k = tap * 3 + ci; a lane's B fragment is 8 two-byte LDS reads of the bf16 patch.
extern unsigned long long* g_f16k_stamps;      // diagnostics buffer (masic_conv_f16k_set_stamps), defined below
struct ConvAArgs {
    const float* x;               // float32 NCHW [B][in_ctot][Hi][Wi], channels in_coff .. in_coff+2
    const uint4* wimg;            // [m 4][s 6][lane 64] x 8 bf16: W[32m + (lane & 31)][k-slot (s, lane >> 5, c)], pack_conv_a_kernel
    const float* bias;            // [128] or null
    const uint4* gdn_img;         // gdn_pack_f16k_kernel image, then beta^[128]
    unsigned short* y16;          // F16K [B][8][Ho*Wo][16]
    int gdn_inverse, Hi, Wi, in_ctot, in_coff, Ho, Wo, tiles_w, tiles_per_img, ntiles;
    unsigned char* y8;            // instead of y16: F8K [B][4][Ho*Wo][32] fp8, quantised with out_inv_scale
    float out_inv_scale;
    unsigned short* y16_pre;      // also store conv + bias before the GDN (F16K), or null
    unsigned long long* stamps;   // diagnostics: s_memrealtime (100 MHz) at 6 points of every tile of workgroup 0 / wave 0, or null
    int store_aux;                // cache policy of the output stores of the wave-private form: 0 plain, 2 nt, 16 sc1 (MASIC_CONVA_STORE_AUX)
};

#ifndef CONVA_ABLATE
#define CONVA_ABLATE 0     // timing experiments only: 1 no GDN math, 2 no convolution, 3 no patch fetch / stash after the first tile, 4 no stores
#endif
constexpr int CA_PH = 19, CA_PW = 67, CA_NEL = 3 * CA_PH * CA_PW;                     // patch of an 8 x 32 tile, stride 2, 5 x 5: [ci][row][col], float32
constexpr int CA_NPT = (CA_NEL + 511) / 512;                                          // patch elements (= 4-byte DMA pieces) per thread
constexpr int CA_KSTEPS = 6, CA_WIMG_BYTES = 4 * CA_KSTEPS * 1024;                    // k-steps of the convolution (see pack_conv_a_kernel); weight fragment image
constexpr int CA_PATCH_BYTES = CA_NPT * 2048;                                         // 3 * 19 * 67 * 4 = 15276, padded to whole wave-instructions

// k order of the first layer's contraction: k-step s = 2p + q covers the kernel-row pair (2p, 2p + 1) -- lane half h = the row within the
// pair -- and the (ci, kw) pairs 8q .. 8q + 7 of the 15 a row has (pair = 5 ci + kw; pair 15 and kernel row 5 are zero-weight padding):
// k = 96 slots for K = 75.  With this order the LDS offset of a lane's operand element is (ci * 19 + 2p + h) * 67 + kw: the only
// per-lane part, h * 67, goes into the lane's base address and every read takes a compile-time immediate offset (the k = tap * 3 + ci
// order of round 2 needed 40 per-lane offsets: 40 VGPRs the compiler kept live across the whole tile loop).
__global__ void pack_conv_a_kernel(const float* __restrict__ w, uint4* __restrict__ wimg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * CA_KSTEPS * 64) return;
    const int ms = idx >> 6, l = idx & 63, m = ms / CA_KSTEPS, sx = ms - m * CA_KSTEPS, r = l & 31, hh = l >> 5;
    const int kh = 2 * (sx >> 1) + hh;
    bf16x8 v;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int pair = 8 * (sx & 1) + c, ci = pair / 5, kw = pair - 5 * ci;
        v[c] = (__bf16)((pair < 15 && kh < 5) ? w[((size_t)(32 * m + r) * 3 + ci) * 25 + kh * 5 + kw] : 0.0f);
    }
    wimg[idx] = __builtin_bit_cast(uint4, v);
}

// The one-product GDN of gdn_in_registers and the per-channel vector add, with LDS reads the compiler does NOT see (inline asm, counted
// lgkmcnt): while a global -> LDS DMA is in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every ds_read it knows about (the DMA might
// alias it), which here would end the next patch's flight -- and wait for the previous tile's output stores -- right after the convolution.
// gaddr: LDS byte address of the gamma^ fragment image + lane * 16;  baddr: LDS byte address of the [128] float vector + 16 h.
template <int OP>      // 0: acc[m] += vec   1: acc[m] *= rsqrt(nrm[m] + vec)   2: acc[m] *= sqrt(nrm[m] + vec)
__device__ __forceinline__ void vec_apply_hidden(f32x16 (&acc)[4], const f32x16 (&nrm)[4], unsigned baddr) {
    static_for<0, 2>([&](auto HALF) {
        constexpr int half = decltype(HALF)::value;
        v4u v[8];
        static_for<0, 8>([&](auto I) {
            constexpr int i = decltype(I)::value, m = 2 * half + (i >> 2), q = i & 3;
            ds_read128<(m * 32 + 8 * q) * 4>(v[i], baddr);
        });
        lgkm_wait<0>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        static_for<0, 8>([&](auto I) {
            constexpr int i = decltype(I)::value, m = 2 * half + (i >> 2), q = i & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned u = v[i][e];            // (a copy first: __builtin_bit_cast on a vector-ELEMENT lvalue reads element 0 whatever e is)
                const float b = __builtin_bit_cast(float, u);
                if constexpr (OP == 0) acc[m][4 * q + e] += b;
                else if constexpr (OP == 1) acc[m][4 * q + e] *= __builtin_amdgcn_rsqf(nrm[m][4 * q + e] + b);
                else acc[m][4 * q + e] *= __builtin_amdgcn_sqrtf(nrm[m][4 * q + e] + b);
            }
        });
    });
}
__device__ __forceinline__ void gdn_in_registers_hidden(f32x16 (&acc)[4], unsigned gaddr, unsigned baddr, int inverse) {
    // the accumulators of the contraction start at beta^ (read straight into them: no add afterwards)
    f32x16 nrm[4];
    {
        v4u vb[16];
        static_for<0, 16>([&](auto I) { constexpr int i = decltype(I)::value; ds_read128<((i >> 2) * 32 + 8 * (i & 3)) * 4>(vb[i], baddr); });
        lgkm_wait<0>(vb[0], vb[1], vb[2], vb[3], vb[4], vb[5], vb[6], vb[7]);
        lgkm_wait<0>(vb[8], vb[9], vb[10], vb[11], vb[12], vb[13], vb[14], vb[15]);
        static_for<0, 4>([&](auto M) {
            constexpr int m = decltype(M)::value;
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            const f32x4v q0 = __builtin_bit_cast(f32x4v, vb[4 * m]), q1 = __builtin_bit_cast(f32x4v, vb[4 * m + 1]);
            const f32x4v q2 = __builtin_bit_cast(f32x4v, vb[4 * m + 2]), q3 = __builtin_bit_cast(f32x4v, vb[4 * m + 3]);
            nrm[m] = __builtin_shufflevector(__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7),
                                             0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        });
    }
    v4u g[2][4];
    static_for<0, 4>([&](auto M) { constexpr int m = decltype(M)::value; ds_read128<((m * 8 + 0) * 2) * 1024>(g[0][m], gaddr); });
    static_for<0, 8>([&](auto SX) {
        constexpr int sx = decltype(SX)::value, cur = sx & 1;
        if constexpr (sx + 1 < 8)
            static_for<0, 4>([&](auto M) { constexpr int m = decltype(M)::value; ds_read128<((m * 8 + sx + 1) * 2) * 1024>(g[cur ^ 1][m], gaddr); });
        v4u bq;                          // x^2 of the 8 channels of this k-step, packed pair by pair (one v_pk_mul + one v_cvt_pk per pair)
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            const float x0 = acc[sx >> 1][8 * (sx & 1) + 2 * cp], x1 = acc[sx >> 1][8 * (sx & 1) + 2 * cp + 1];
            bq[cp] = pack2bf(__fmul_rn(x0, x0), __fmul_rn(x1, x1));
        }
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bq);
        lgkm_wait<(sx + 1 < 8 ? 4 : 0)>(g[cur][0], g[cur][1], g[cur][2], g[cur][3]);
        static_for<0, 4>([&](auto M) {
            constexpr int m = decltype(M)::value;
            nrm[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g[cur][m]), bh, nrm[m], 0, 0, 0);
        });
    });
    if (inverse) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] *= __builtin_amdgcn_sqrtf(nrm[m][e]);
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] *= __builtin_amdgcn_rsqf(nrm[m][e]);
    }
}

// X3: three-product GDN contraction; OUT: 0 F16K, 1 F8K (fp8), 2 F16K + the pre-GDN result (training), 3 F16K WITHOUT the GDN (the
// plain Conv2d(3 -> 128, k5, s2): the input gradient of g_s_conv4 = ConvTranspose2d(128 -> 3), MASIC.py:550, in the training step --
// no gamma^ image is loaded).  Compile-time variants: with
// the choices as run-time branches both GDN forms were inlined and the kernel spilled 68 VGPRs in its tile loop (129 us per launch
// at 8x512x512 against 80 for the round-1 kernel).
template <bool X3, int OUT>
__global__ __launch_bounds__(512, 1) void conv_a_gdn_f16k(const ConvAArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int WIMG = 65536, VEC = WIMG + CA_WIMG_BYTES, PATCH = VEC + 1024;      // LDS map: gamma image | weights | bias, beta^ | 2 patches
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;   // LDS byte address of lds[0]

    // ---- once per workgroup: fragment images by DMA, bias / beta^ by plain stores
    {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gdn_img, 0, 65536 + 512, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, CA_WIMG_BYTES, 0x00020000);
        if constexpr (OUT != 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k) dma_buf16(rg, lds + (wave * 8 + k) * 1024, lane * 16, (wave * 8 + k) * 1024);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            dma_buf16(rw, lds + WIMG + (wave * 3 + k) * 1024, lane * 16, (wave * 3 + k) * 1024);            // 8 waves x 3 KiB = 24 KiB
        float* vec = reinterpret_cast<float*>(lds + VEC);
        if (tid < 128) {
            vec[tid] = a.bias != nullptr ? a.bias[tid] : 0.0f;
            if constexpr (OUT != 3) vec[128 + tid] = reinterpret_cast<const float*>(a.gdn_img + 4 * 8 * 2 * 64)[tid];
        }
    }
    // ---- patch element map of this thread (tile independent): element el = tid + 512 i -> (ci, row, col) of the float32 LDS patch
    // [ci][row 19][col 67]; the patch is filled by 4-byte global -> LDS DMA (wave-instruction i of wave w lands at dwords 512 i + 64 w ..),
    // so it costs no registers and no conversion pass, and elements outside the image come back as zeros from the buffer range check.
    // (Round 2 staged it through registers with a padding select right behind the loads: the compiler put `s_waitcnt vmcnt(0)` there, at
    // the TOP of the tile loop, which waited for the loads' full latency AND -- vmcnt retires in order -- for the previous tile's 64 KiB of
    // output stores before the first MFMA of every tile: load latency + store drain + compute in series, 8.5 us per tile for 1.7 us of MFMAs.)
    const size_t plane = (size_t)a.Hi * a.Wi;
    unsigned offrel[CA_NPT], rc[CA_NPT];       // byte offset of the element relative to the patch origin; row | col << 8 | valid << 16
#pragma unroll
    for (int i = 0; i < CA_NPT; ++i) {
        const int el = tid + 512 * i;
        const int ci = el / (CA_PH * CA_PW), rem = el - ci * (CA_PH * CA_PW);
        const int row = rem / CA_PW, col = rem - row * CA_PW;
        offrel[i] = (unsigned)(((size_t)ci * plane + (size_t)row * a.Wi + col) * 4);
        rc[i] = (unsigned)row | ((unsigned)col << 8) | (el < CA_NEL ? 1u << 16 : 0u);
    }
    auto fetch = [&](int tile, int buf) {           // global -> LDS, left in flight
        const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
        const int ih0 = (t / a.tiles_w) * 16 - 2, iw0 = (t % a.tiles_w) * 64 - 2;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + ((size_t)b * a.in_ctot + a.in_coff) * plane), 0,
                                                                            (int)(3 * plane * 4), 0x00020000);
        const unsigned origin = (unsigned)((ih0 * a.Wi + iw0) * 4);
        unsigned char* dst = lds + PATCH + buf * CA_PATCH_BYTES + wave * 256;
#pragma unroll
        for (int i = 0; i < CA_NPT; ++i) {
            const int ih = ih0 + (int)(rc[i] & 255u), iw = iw0 + (int)((rc[i] >> 8) & 255u);
            const unsigned ok = (rc[i] >> 16) & (unsigned)(ih >= 0) & (unsigned)(ih < a.Hi) & (unsigned)(iw >= 0) & (unsigned)(iw < a.Wi);
            const unsigned off = ok ? offrel[i] + origin : 0xC0000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + i * 2048), 4, (int)off, 0, 0, 0);
        }
    };

    // Memory schedule of the tile loop (OUT == 0; the other forms wait for everything, vmcnt(0)).  Per tile t, in this order:
    //   convolution + GDN on patch buffer t & 1  ->  s_waitcnt vmcnt(8)  ->  s_barrier  ->  issue the patch DMA of tile t + 2 into buffer
    //   t & 1 (every wave is done reading it)  ->  issue the 8 output stores of tile t.
    // vmcnt retires in order, so at the wait of tile t + 1 the queue holds [DMA(t + 2)][8 stores of t]: vmcnt(8) waits for the patch and
    // for everything older, and leaves the 8 stores IN FLIGHT across the barrier -- there is always a tile of stores (64 KiB per CU)
    // draining while the next tile computes, and the HBM pipe never idles between two tiles' bursts (with the stores issued before the
    // DMA, or vmcnt(0), each round of 256 tiles paid its drain in full: 60 us per launch for 15 us of arithmetic and 21 us of stores).
    // The stores go through a buffer resource with out-of-range offsets for lanes outside the picture: always exactly 8 instructions.
    constexpr bool COUNTED = OUT == 0;
    int tile = blockIdx.x;
    if (tile < a.ntiles) fetch(tile, 0);
    if (COUNTED && tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // The bias rides in the contraction: k-slot (k-step 1, element 7) is the 16th, unused (ci, kw) pair of kernel rows 0 / 1 -- its B operand
    // is set to 1 and its weight to bias_hi (lane half 0) / bias_lo (lane half 1), a bf16 split with 2^-17 relative error: no 64 adds and
    // 16 LDS reads per lane and tile.  The weight image arrived by DMA without it; it is patched here, once per workgroup.
    if (tid < 256) {
        const int co = tid & 127, hh = tid >> 7;
        const float bv = reinterpret_cast<const float*>(lds + VEC)[co];
        const __bf16 hi = (__bf16)bv;
        const __bf16 val = hh == 0 ? hi : (__bf16)(bv - (float)hi);
        reinterpret_cast<__bf16*>(lds + WIMG + (((co >> 5) * CA_KSTEPS + 1) * 64 + (co & 31) + 32 * hh) * 16)[7] = val;
    }
    __syncthreads();
    const unsigned char* gimg = lds + lane * 16;
    const float* vec = reinterpret_cast<const float*>(lds + VEC);
    const size_t oplane = (size_t)a.Ho * a.Wo;
    int buf = 0;
    unsigned long long* stamp = (a.stamps != nullptr && blockIdx.x == 0 && tid == 0) ? a.stamps : nullptr;
    for (; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
        const int next = tile + gridDim.x;
        if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
        if (!COUNTED && next < a.ntiles && CONVA_ABLATE != 3) fetch(next, buf ^ 1);   // in flight during this tile's contraction and GDN
        // ---- convolution: 5 k-steps x 4 channel blocks
        const float* pl = reinterpret_cast<const float*>(lds + PATCH + buf * CA_PATCH_BYTES) + (2 * wave + h) * CA_PW + 2 * j;
        f32x16 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
#pragma unroll
        for (int sx = 0; sx < (CONVA_ABLATE == 2 ? 0 : CA_KSTEPS); ++sx) {
            bf16x8 bfr;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int pair = 8 * (sx & 1) + c, ci = pair < 15 ? pair / 5 : 0, kw = pair < 15 ? pair % 5 : 0;
                if (sx == 1 && c == 7) bfr[c] = (__bf16)1.0f;                              // the bias slot (see above)
                else bfr[c] = (__bf16)pl[(ci * CA_PH + 2 * (sx >> 1)) * CA_PW + kw];      // (+ h rows: in the lane's base; kernel row 5 / pair 15: zero weights)
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + WIMG + (m * CA_KSTEPS + sx) * 1024 + lane * 16));
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[m], 0, 0, 0);
            }
        }
        if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
        // ---- GDN, F16K store (LDS reads hidden from the compiler: the next patch's DMA is in flight)
        if constexpr (OUT == 2) {
            const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
            const int oh = (t / a.tiles_w) * 8 + wave, ow = (t % a.tiles_w) * 32 + j;
            if (oh < a.Ho && ow < a.Wo) {
                unsigned short* yp = a.y16_pre + (((size_t)b * 8) * oplane + (size_t)oh * a.Wo + ow) * 16 + 8 * h;
                const unsigned op16 = (unsigned)oplane * 16;
#pragma unroll
                for (int m = 0; m < 4; ++m) store_f16k_tile(acc[m], yp + (size_t)(2 * m) * op16, op16);
            }
        }
        if constexpr (OUT != 3) {
            if (CONVA_ABLATE != 1) {
                if constexpr (X3) gdn_in_registers<true>(acc, gimg, vec + 128 + 4 * h, a.gdn_inverse & 1);
                else gdn_in_registers_hidden(acc, ldsb + lane * 16, ldsb + VEC + 512 + 16 * h, a.gdn_inverse & 1);
            }
        }
        // the next tile's patch has had the whole convolution + GDN to land: wait for it (vmcnt(0): that is also the PREVIOUS tile's output
        // stores, issued a full tile of MFMA work ago), the workgroup barrier, and only THEN this tile's stores -- nothing waits for them
        // until the end of the next tile.  The stores are the kernel's HBM floor (134 MB per launch at 8 x 512 x 512); they drain UNDER
        // the next tile's MFMAs.
        if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
        if constexpr (COUNTED) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime();
        if constexpr (COUNTED) {
            if (next + (int)gridDim.x < a.ntiles && CONVA_ABLATE != 3) fetch(next + gridDim.x, buf);
            const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
            const int oh = (t / a.tiles_w) * 8 + wave, ow = (t % a.tiles_w) * 32 + j;
            const unsigned rec_bytes = (unsigned)oplane * 32;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y16 + (size_t)b * 8 * oplane * 16), 0, (int)(8 * rec_bytes), 0x00020000);
            const unsigned voff = (oh < a.Ho && ow < a.Wo && CONVA_ABLATE != 4) ? ((unsigned)oh * a.Wo + ow) * 32 + 16 * h : 0xC0000000u;
#pragma unroll
            for (int m = 0; m < 4; ++m) store_f16k_tile_buf(acc[m], ry, voff + (unsigned)(2 * m) * rec_bytes, rec_bytes);
            if (stamp) { stamp[5] = __builtin_amdgcn_s_memrealtime(); stamp += 6; }
        } else if (CONVA_ABLATE != 4) {
            const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
            const int oh = (t / a.tiles_w) * 8 + wave, ow = (t % a.tiles_w) * 32 + j;
            if (oh < a.Ho && ow < a.Wo) {
                if constexpr (OUT == 1) {
                    unsigned char* yb = a.y8 + (((size_t)b * 4) * oplane + (size_t)oh * a.Wo + ow) * 32 + 16 * h;
#pragma unroll
                    for (int m = 0; m < 4; ++m) store_f8k_tile(acc[m], yb + (size_t)m * oplane * 32, a.out_inv_scale);
                } else {
                    unsigned short* yb = a.y16 + (((size_t)b * 8) * oplane + (size_t)oh * a.Wo + ow) * 16 + 8 * h;
                    const unsigned op16 = (unsigned)oplane * 16;
#pragma unroll
                    for (int m = 0; m < 4; ++m) store_f16k_tile(acc[m], yb + (size_t)(2 * m) * op16, op16);
                }
            }
        }
    }
}

// The inference form of the first layer (one-product GDN, F16K out, Wi % 4 == 0) with WAVE-PRIVATE patches and no barrier in the tile loop.
// Wave w of the workgroup owns output row w of the 8 x 32 tile: its own [15 rows = (ci, kh)][72 floats] float32 patch (4 320 bytes, two
// buffers) is filled by 16-byte global -> LDS DMA -- 270 pieces = 5 wave-instructions, against 8 of the 64 four-byte instructions per
// workgroup and tile of the shared patch, whose issue (~80 cycles each on the CU's one address path) cost more than the tile's MFMAs --
// and read by that wave alone, so the waves of a workgroup never wait for each other: they drift out of phase, and one wave's MFMAs run
// under the other's VALU work on the same SIMD (in lock step both waves of a SIMD were in the VALU-heavy epilogue at the same time).
// Memory schedule per wave and tile k:   s_waitcnt vmcnt(21) -> convolution on buffer k & 1 -> DMA of tile k + 2 into that buffer ->
// GDN -> 8 output stores.  Every iteration issues exactly 5 DMA + 8 store instructions (lanes / tiles with nothing to move get
// out-of-range offsets: loads write zeros, stores are dropped), so at the top of tile k the instructions younger than the DMA of tile k
// are [8 stores of k-2][5 DMA of k+1][8 stores of k-1] = 21: vmcnt(21) is exact, and two tiles of stores stay in flight.
constexpr int CW_PCS = 18, CW_ROWS = 15, CW_NP = CW_ROWS * CW_PCS, CW_NI = (CW_NP + 63) / 64, CW_BYTES = CW_NP * 16, CW_PITCH = CW_PCS * 4;
static_assert(CW_NI == 5 && CW_BYTES == 4320, "vmcnt(21) below counts 5 DMA instructions per tile");
__global__ __launch_bounds__(512, 1) void conv_a_gdn_f16k_w(const ConvAArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int WIMG = 65536, VEC = WIMG + CA_WIMG_BYTES, PATCH = VEC + 1024;      // LDS map: gamma image | weights | bias, beta^ | 8 x 2 patches
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned long long* const pst = (a.stamps != nullptr && tid == 0 && (blockIdx.x == 0 || blockIdx.x + 1 == gridDim.x)) ? a.stamps + 48 + (blockIdx.x == 0 ? 0 : 8) : nullptr;
    if (pst) pst[0] = __builtin_amdgcn_s_memrealtime();
    {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.gdn_img, 0, 65536 + 512, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, CA_WIMG_BYTES, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8; ++k) dma_buf16(rg, lds + (wave * 8 + k) * 1024, lane * 16, (wave * 8 + k) * 1024);
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_buf16(rw, lds + WIMG + (wave * 3 + k) * 1024, lane * 16, (wave * 3 + k) * 1024);
        float* vec = reinterpret_cast<float*>(lds + VEC);
        if (tid < 128) {
            vec[tid] = a.bias != nullptr ? a.bias[tid] : 0.0f;
            vec[128 + tid] = reinterpret_cast<const float*>(a.gdn_img + 4 * 8 * 2 * 64)[tid];
        }
    }
    // piece map of this lane (tile independent): piece p = lane + 64 i -> patch row r = p / 18 = 5 ci + kh, 16-byte column piece pc = p % 18
    const size_t plane = (size_t)a.Hi * a.Wi;
    unsigned offrel[CW_NI], rc[CW_NI];          // byte offset relative to the patch origin; kh | pc << 8 | valid << 16
#pragma unroll
    for (int i = 0; i < CW_NI; ++i) {
        const int p = lane + 64 * i, r = p / CW_PCS, pc = p - r * CW_PCS, ci = r / 5, kh = r - 5 * ci;
        offrel[i] = (unsigned)(((size_t)ci * plane + (size_t)kh * a.Wi + 4 * pc) * 4);
        rc[i] = (unsigned)kh | ((unsigned)pc << 8) | (p < CW_NP ? 1u << 16 : 0u);
    }
    unsigned char* const mypatch = lds + PATCH + wave * (2 * CW_BYTES);
    auto fetch = [&](int tile, int buf) {           // always CW_NI instructions; a tile past the end / a row below the picture loads zeros
        const bool live = tile < a.ntiles;
        const int tl = live ? tile : 0;
        const int b = tl / a.tiles_per_img, t = tl - b * a.tiles_per_img;
        const int oh = (t / a.tiles_w) * 8 + wave;
        const int ih0 = 2 * oh - 2, iw0 = (t % a.tiles_w) * 64 - 4;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + ((size_t)b * a.in_ctot + a.in_coff) * plane), 0,
                                                                            (int)(3 * plane * 4), 0x00020000);
        const unsigned origin = (unsigned)((ih0 * a.Wi + iw0) * 4);
        const unsigned rowok = (unsigned)(live && oh < a.Ho);
        unsigned char* dst = mypatch + buf * CW_BYTES;
#pragma unroll
        for (int i = 0; i < CW_NI; ++i) {
            const int ih = ih0 + (int)(rc[i] & 255u), iw = iw0 + 4 * (int)((rc[i] >> 8) & 255u);
            const unsigned ok = rowok & (rc[i] >> 16) & (unsigned)(ih >= 0) & (unsigned)(ih < a.Hi) & (unsigned)(iw >= 0) & (unsigned)(iw < a.Wi);
            const unsigned off = ok ? offrel[i] + origin : 0xC0000000u;
            if (i + 1 < CW_NI || lane < CW_NP - 64 * (CW_NI - 1))      // the last instruction covers 14 pieces: the other lanes are masked off (EXEC)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, (int)off, 0, 0, 0);
        }
    };
    int tile = blockIdx.x;
    fetch(tile, 0);
    fetch(tile + gridDim.x, 1);
    if (pst) pst[1] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pst) pst[2] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (pst) pst[3] = __builtin_amdgcn_s_memrealtime();
    if (tid < 256) {                // the bias as a k-slot of the contraction (see conv_a_gdn_f16k)
        const int co = tid & 127, hh = tid >> 7;
        const float bv = reinterpret_cast<const float*>(lds + VEC)[co];
        const __bf16 hi = (__bf16)bv;
        const __bf16 val = hh == 0 ? hi : (__bf16)(bv - (float)hi);
        reinterpret_cast<__bf16*>(lds + WIMG + (((co >> 5) * CA_KSTEPS + 1) * 64 + (co & 31) + 32 * hh) * 16)[7] = val;
    }
    __syncthreads();
    const size_t oplane = (size_t)a.Ho * a.Wo;
    const unsigned rec_bytes = (unsigned)oplane * 32;
    int buf = 0;
    unsigned long long* stamp = (a.stamps != nullptr && blockIdx.x == 0 && tid == 0) ? a.stamps : nullptr;
    for (; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
        if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
        if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
        // ---- convolution: 6 k-steps x 4 channel blocks; patch element (ci, kh = 2p + h, column 2 j + kw - 2) = row 5 ci + kh, float 2 j + kw + 2
        const float* pl = reinterpret_cast<const float*>(mypatch + buf * CW_BYTES) + h * CW_PITCH + 2 * j + 2;
        f32x16 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
#pragma unroll
        for (int sx = 0; sx < CA_KSTEPS; ++sx) {
            bf16x8 bfr;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int pair = 8 * (sx & 1) + c, ci = pair < 15 ? pair / 5 : 0, kw = pair < 15 ? pair % 5 : 0;
                // (kernel row 5 -- k-steps 4, 5, lane half 1 -- has zero weights and reads row 0 of the next ci / the 280 bytes behind this
                // buffer: the wave's other buffer, the next wave's, or past the allocation (reads as 0) -- all filled with finite floats
                // before the first tile)
                if (sx == 1 && c == 7) bfr[c] = (__bf16)1.0f;
                else bfr[c] = (__bf16)pl[(ci * 5 + 2 * (sx >> 1)) * CW_PITCH + kw];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + WIMG + (m * CA_KSTEPS + sx) * 1024 + lane * 16));
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[m], 0, 0, 0);
            }
        }
        if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of the buffer are done: refill it for the tile after next
        fetch(tile + 2 * (int)gridDim.x, buf);
        if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
        gdn_in_registers_hidden(acc, ldsb + lane * 16, ldsb + VEC + 512 + 16 * h, a.gdn_inverse & 1);
        if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime();
        {
            const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
            const int oh = (t / a.tiles_w) * 8 + wave, ow = (t % a.tiles_w) * 32 + j;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y16 + (size_t)b * 8 * oplane * 16), 0, (int)(8 * rec_bytes), 0x00020000);
            const unsigned voff = (oh < a.Ho && ow < a.Wo) ? ((unsigned)oh * a.Wo + ow) * 32 + 16 * h : 0xC0000000u;
            if (a.store_aux == 16) {
#pragma unroll
                for (int m = 0; m < 4; ++m) store_f16k_tile_buf<16>(acc[m], ry, voff + (unsigned)(2 * m) * rec_bytes, rec_bytes);
            } else if (a.store_aux == 2) {
#pragma unroll
                for (int m = 0; m < 4; ++m) store_f16k_tile_buf<2>(acc[m], ry, voff + (unsigned)(2 * m) * rec_bytes, rec_bytes);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) store_f16k_tile_buf<0>(acc[m], ry, voff + (unsigned)(2 * m) * rec_bytes, rec_bytes);
            }
        }
        if (stamp) { stamp[5] = __builtin_amdgcn_s_memrealtime(); stamp += 6; }
    }
    if (pst) pst[4] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pst) pst[5] = __builtin_amdgcn_s_memrealtime();
}

struct F16kCfg {
    int ok;
    int KS, T, L, NP, D;          // template selection: strided conv <1,4,D,6,1>, stride-1 walk <2,2,D,3,2>, large stride-1 walk <1,2,D,*,2> with NP > 1
    int TW, TWlog, SR, TH, PH, PW, PWh, NPIXp, PB;
    int Cin16, ncb;
    unsigned phase_off[4], stream_bytes[4];
    size_t packed_bytes, lds_bytes;
};

constexpr int F16K_D = 3;

unsigned long long* g_f16k_stamps = nullptr;

F16kCfg choose_f16k(const masic_conv_desc_t& d, const ConvGeom* g, int nphase) {
    F16kCfg c{};
    const bool f8 = d.prec == MASIC_PREC_FP8;       // fp8 operands: 32-channel blocks, 128-channel co-blocks only, 256-pixel tiles
    const int cblk = f8 ? 32 : 16;
    if (d.Cout < 32 || d.Cout % 32 != 0 || d.Cin < cblk || d.Cin % cblk != 0 || d.in_op != MASIC_INOP_NONE || d.act == MASIC_ACT_SOFTMAX_C) return c;
    if (f8 && d.Cout < 64) return c;
    int span_h = 0, span_w = 0, min_taps = 1 << 30, max_taps = 0;
    for (int p = 0; p < nphase; ++p) {
        span_h = span_h > g[p].nth ? span_h : g[p].nth;
        span_w = span_w > g[p].ntw ? span_w : g[p].ntw;
        min_taps = min_taps < g[p].ntaps ? min_taps : g[p].ntaps;
        max_taps = max_taps > g[p].ntaps ? max_taps : g[p].ntaps;
    }
    const int is = g[0].is, Wp = g[0].Wp, Hp = g[0].Hp;
    int NPI;
    c.NP = 1;
    c.D = F16K_D;
    if (is == 2) { c.KS = 1; c.T = 4; NPI = 12; c.L = 1; }    // strided conv: big patch, 16-channel chunks, 4 taps per step
    else { c.KS = 2; c.T = 2; NPI = 8; c.L = 2; }             // stride-1 walk (transposed phases, 3x3, 5x5 s1, masked): 32-channel chunks
    // strided 5x5: 5 taps per step -- 25 taps need no zero-weight padding (4 per step pad to 28: 12 % more MFMAs) and a chunk
    // takes 5 barriers instead of 7; the 20-KiB slab groups leave room for a 3-slot ring only (look-ahead 2 steps = 40 MFMAs per
    // wave).  MASIC_F16K_T5=0 keeps 4 taps per step (A/B timing).
    static const bool t5 = !(getenv("MASIC_F16K_T5") && getenv("MASIC_F16K_T5")[0] == '0');
    if (!f8 && t5 && is == 2 && min_taps == 25 && max_taps == 25) { c.T = 5; c.D = 2; }
    // large stride-1-walk layers: 512- (128-channel blocks) or 1024-pixel tiles (<= 32 channels), 16-channel chunks
    if (!f8 && is == 1 && Wp >= 32) {
        const int np = d.Cout <= 32 ? 4 : 2;
        const long tiles = (long)ceil_div(Wp, 32) * ceil_div(Hp, 8 * np) * nphase * d.B * ceil_div(d.Cout, 128);
        if (tiles >= 512 && ceil_div(min_taps, 2) >= 2) { c.NP = np; c.KS = 1; c.T = 2; c.L = 2; NPI = np == 4 ? 10 : 6; }
        // 128-channel blocks: 32-channel chunks (2 taps x 2 channel blocks = 32 MFMAs per wave and barrier instead of 16), one chunk of
        // patch look-ahead and a 3-slot ring.  MASIC_F16K_KS2=0 keeps 16-channel chunks (A/B timing).
        static const bool ks2 = !(getenv("MASIC_F16K_KS2") && getenv("MASIC_F16K_KS2")[0] == '0');
        if (ks2 && c.NP == 2 && d.Cin % 32 == 0) { c.KS = 2; c.D = 2; c.L = 1; NPI = 10; }
        // 64 / 96 output channels (no GDN: that needs 128) (the 3x3 layers of Independent_EN and their input gradients): the K loop is short (K = 9 x
        // 32 ... 96) and the per-workgroup prologue + epilogue were 35 % of a workgroup's life (in-kernel stamps: 4.9 + 17.5 + 5.6 us
        // on the 96 -> 96 layer).  "Lite" configuration: 256-pixel tiles, 2-slot weight ring, 2 patch buffers = 77 KiB of LDS and
        // <= 128 VGPRs, so that TWO workgroups share a CU and one's prologue / epilogue / barrier waits run under the other's MFMAs.
        // MASIC_F16K_LITE=0 keeps the 512-pixel single-workgroup form (A/B timing).
        static const bool lite = !(getenv("MASIC_F16K_LITE") && getenv("MASIC_F16K_LITE")[0] == '0');
        static const int lite_min = getenv("MASIC_F16K_LITE_MIN") ? atoi(getenv("MASIC_F16K_LITE_MIN")) : 33;     // smallest Cout that takes it
        if (lite && c.NP > 1 && d.Cout >= lite_min && d.Cout <= 96 && d.Cin % 32 == 0 && ceil_div(min_taps, 2) >= 2) {
            c.NP = 1; c.KS = 2; c.T = 2; c.D = 1; c.L = 1; NPI = 6;
        }
    }
    if (ceil_div(max_taps, c.T) * (c.T == 5 ? 8 : c.T) > MAXTAPS || ceil_div(min_taps, c.T) < 2) return c;
    c.Cin16 = d.Cin / cblk;                                   // channel blocks (records) per pixel
    if (c.Cin16 % c.KS != 0) return c;                        // whole chunks only
    c.TW = Wp > 16 ? 32 : (Wp > 8 ? 16 : 8);
    c.TWlog = 0;
    while ((1 << c.TWlog) < c.TW) ++c.TWlog;
    c.SR = 32 / c.TW;
    c.TH = c.SR * 8 * c.NP;
    c.PH = (c.TH - 1) * is + span_h;
    c.PW = (c.TW - 1) * is + span_w;
    c.PWh = (c.PW + 1) / 2;
    c.NPIXp = round_up(is == 2 ? 2 * c.PH * c.PWh : c.PH * c.PW, 32);
    const int ninstr = c.KS * (c.NPIXp / 32);
    if (ceil_div(ninstr, 4) > NPI) return c;
    c.PB = ninstr * 1024;
    c.ncb = ceil_div(d.Cout, 128);
    const int nchunks = c.Cin16 / c.KS;
    size_t off = 0;
    for (int p = 0; p < nphase; ++p) {
        const size_t sb = (size_t)nchunks * ceil_div(g[p].ntaps, c.T) * c.T * c.KS * 4096;
        c.phase_off[p] = (unsigned)off;
        c.stream_bytes[p] = (unsigned)sb;
        off += sb * c.ncb;
    }
    if (off >= (1u << 31)) return c;
    c.packed_bytes = off;
    c.lds_bytes = (size_t)(c.D + 1) * c.T * c.KS * 4096 + (size_t)(c.L + 1) * c.PB + 1024 + MAXTAPS * 4;
    if (c.lds_bytes > 160 * 1024 || c.lds_bytes < 65536 + 2048) return c;       // (the fused GDN parks its 64 KiB image + beta^ at offset 0)
    c.ok = 1;
    return c;
}

// Layers whose grid leaves most of the 256 CUs idle (the 128 -> 192 layer at 64^2 -> 32^2, hyper transforms, context models:
// 8 ... 128 workgroups): the 128-channel co-block is split over 2 or 4 workgroups of 2 / 1 accumulator tiles each.  The patch and
// the slab DMA are repeated per workgroup (they come from L2), the MFMA work is not.  MASIC_F16K_MSPLIT=1 disables it (A/B).
int f16k_msplit(const masic_conv_desc_t& d, const F16kCfg& c, int np, const ConvGeom* g, bool gdn, bool special) {
    if (d.prec == MASIC_PREC_FP8 || gdn || c.NP != 1 || special || d.Cout < 64) return 1;
    static const int forced = getenv("MASIC_F16K_MSPLIT") ? atoi(getenv("MASIC_F16K_MSPLIT")) : 0;
    const int ntiles = ceil_div(g[0].Wp, c.TW) * ceil_div(g[0].Hp, c.TH);
    const long wgs = (long)ntiles * np * c.ncb * d.B;
    const int ms = forced ? forced : (wgs <= 64 ? 4 : (wgs <= 160 ? 2 : 1));
    return (ms == 2 || ms == 4) ? ms : 1;
}
int msplit_nm(const masic_conv_desc_t& d, const F16kCfg& c, int np, const ConvGeom* g) { return 4 / f16k_msplit(d, c, np, g, false, false); }

}  // namespace

extern "C" int masic_conv_f16k_supported(const masic_conv_desc_t* d) {
    if (check_desc(d) != MASIC_OK) return 0;
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    return choose_f16k(*d, g, np).ok;
}

extern "C" int masic_conv_f16k_kernel_name(const masic_conv_desc_t* d, int gdn, char* buf, size_t n) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    MASIC_REQUIRE(buf && n > 0, MASIC_ERR_ARG, "conv_f16k_kernel_name: null buffer");
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const F16kCfg c = choose_f16k(*d, g, np);
    MASIC_REQUIRE(c.ok, MASIC_ERR_UNSUPPORTED, "conv_f16k: layer shape has no F16K configuration");
    const int nm = (d->Cout <= 32 && !gdn) ? 1 : 4;
    if (d->prec == MASIC_PREC_FP8) snprintf(buf, n, "conv_f16k<%d, %d, %d, %d, %d, %s, 4, 1, true>", c.KS, c.T, F16K_D, c.KS == 1 ? 6 : 4, c.L, gdn ? "true" : "false");
    else if (c.D == 1) snprintf(buf, n, "conv_f16k<2, 2, 1, 3, 1, false, %d, 1, false>", d->Cout <= 32 ? 1 : (d->Cout <= 64 ? 2 : 3));
    else if (c.NP == 2 && c.KS == 2 && !gdn && d->Cout <= 96) snprintf(buf, n, "conv_f16k<2, 2, 2, 5, 1, false, %d, 2, false>", d->Cout <= 64 ? 2 : 3);
    else if (c.NP == 2 && c.KS == 2) snprintf(buf, n, "conv_f16k<2, 2, 2, 5, 1, %s, 4, 2, false>", gdn ? "true" : "false");
    else if (c.NP > 1) snprintf(buf, n, "conv_f16k<1, 2, %d, %d, 2, %s, %d, %d, false>", F16K_D, c.NP == 4 ? 5 : 3, gdn ? "true" : "false", nm, c.NP);
    else if (c.KS == 1) snprintf(buf, n, "conv_f16k<1, %d, %d, 6, 1, %s, %d, 1, false>", c.T, c.D, gdn ? "true" : "false", gdn ? 4 : msplit_nm(*d, c, np, g));
    else snprintf(buf, n, "conv_f16k<2, 2, %d, 4, 2, %s, %d, 1, false>", F16K_D, gdn ? "true" : "false", (gdn || nm == 1) ? nm : msplit_nm(*d, c, np, g));
    return MASIC_OK;
}

extern "C" size_t masic_conv_f16k_packed_bytes(const masic_conv_desc_t* d) {
    if (check_desc(d) != MASIC_OK) return 0;
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const F16kCfg c = choose_f16k(*d, g, np);
    return c.ok ? c.packed_bytes : 0;
}

namespace {
int pack_stream_job(const float* w, void* w_packed, const masic_conv_desc_t* d, PackStreamJob& j) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    MASIC_REQUIRE(w && w_packed, MASIC_ERR_ARG, "conv_f16k_pack_weight: null pointer");
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const F16kCfg c = choose_f16k(*d, g, np);
    MASIC_REQUIRE(c.ok, MASIC_ERR_UNSUPPORTED, "conv_f16k: layer shape has no F16K configuration");
    j.a = PackStreamArgs{w, d->Cin, d->Cout, d->KH, d->KW, d->transposed, c.KS, c.T, c.Cin16 / c.KS, c.ncb, {}, {}};
    size_t tot = 0;
    for (int p = 0; p < np; ++p) {
        j.a.gs[p] = g[p];
        j.a.phase_offs[p] = c.phase_off[p];
        const size_t t = (size_t)c.stream_bytes[p] / 2 * c.ncb;
        tot = tot > t ? tot : t;
    }
    int nb = (int)((tot + 255) / 256);
    if (nb > 8192) nb = 8192;
    j.wp = (unsigned short*)w_packed;
    j.nb = nb;
    j.np = np;
    return MASIC_OK;
}
}  // namespace

extern "C" int masic_conv_f16k_pack_weight(const float* w, void* w_packed, const masic_conv_desc_t* d, void* stream) {
    PackStreamJob j{};
    const int rc = pack_stream_job(w, w_packed, d, j);
    if (rc != MASIC_OK) return rc;
    hipLaunchKernelGGL(pack_f16k_stream_kernel, dim3(j.nb, j.np), dim3(256), 0, (hipStream_t)stream, j.a, j.wp);
    return masic_launch_status("conv_f16k_pack_weight");
}

// Batched form: masic_conv_f16k_pack_job writes the job one masic_conv_f16k_pack_weight(w, w_packed, d) call stands for into host
// memory (masic_conv_f16k_pack_job_bytes() bytes per job; returns the job's grid width >= 1, or a negative error code);
// masic_conv_f16k_pack_jobs_run launches a table of such jobs -- in DEVICE memory, njobs <= 65535 -- as one kernel (max_nb: the
// largest grid width among them).  The pointers inside the jobs must stay valid; weights are read at launch time.
extern "C" size_t masic_conv_f16k_pack_job_bytes(void) { return sizeof(PackStreamJob); }

extern "C" int masic_conv_f16k_pack_job(const float* w, void* w_packed, const masic_conv_desc_t* d, void* job_host) {
    MASIC_REQUIRE(job_host != nullptr, MASIC_ERR_ARG, "conv_f16k_pack_job: null pointer");
    PackStreamJob j{};
    const int rc = pack_stream_job(w, w_packed, d, j);
    if (rc != MASIC_OK) return rc < 0 ? rc : -rc;
    if (j.nb > 96) j.nb = 96;            // the batched grid is (widest job, jobs, 4 phases): the loops are grid-stride, empty blocks are not free
    *(PackStreamJob*)job_host = j;
    return j.nb;
}

extern "C" int masic_conv_f16k_pack_jobs_run(const void* jobs_dev, int njobs, int max_nb, void* stream) {
    MASIC_REQUIRE(jobs_dev && njobs >= 1 && njobs <= 65535 && max_nb >= 1, MASIC_ERR_ARG, "conv_f16k_pack_jobs_run: bad arguments");
    hipLaunchKernelGGL(pack_f16k_stream_multi_kernel, dim3(max_nb, njobs, 4), dim3(256), 0, (hipStream_t)stream, (const PackStreamJob*)jobs_dev);
    return masic_launch_status("conv_f16k_pack_jobs_run");
}

namespace {
int f16k_launch(const void* x_f16k, const void* w_packed, const float* bias, const float* gate,
                const void* gdn_packed, int gdn_inverse, int d2s, float* y_nchw, void* y_f16k,
                const masic_conv_desc_t* d, void* stream, const float* wscale = nullptr, void* y_f8k = nullptr, float out_inv_scale = 0.0f,
                const void* res1 = nullptr, const void* res2 = nullptr, int res_ctot = 0, const float* res32 = nullptr, int cout_store = 0,
                void* y_pre = nullptr, const void* mask16 = nullptr, float mask_slope = 0.0f);
}

// Extended residual form for the training step of Independent_EN (masic_amd/autograd.py: EnhancementBlockFn), everything F16K:
//   y = act(conv(x) + bias) * act'(mask) + res1 + res2,   y_pre (optional) = the value before the residual adds.
// Forward: mask NULL, y_pre = the LeakyReLU output the backward needs for its mask.  Backward (input gradient = the transposed
// convolution on the same weight): mask = the forward activation's output of the PRODUCER layer, mask_slope = 0.01 / 0 (LeakyReLU /
// ReLU), res1 / res2 = the gradients arriving over the identity paths.  mask, res1, res2, y_pre: res_ctot channels.
extern "C" int masic_conv_f16k_res_ex_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                                          const void* mask, float mask_slope, void* y_pre_f16k, void* y_f16k, const masic_conv_desc_t* d, void* stream) {
    MASIC_REQUIRE(d != nullptr && y_f16k != nullptr, MASIC_ERR_ARG, "conv_f16k_res_ex_fwd: null pointer");
    MASIC_REQUIRE(res1 != nullptr || res2 == nullptr, MASIC_ERR_ARG, "conv_f16k_res_ex_fwd: res2 without res1");
    MASIC_REQUIRE((res1 == nullptr && mask == nullptr && y_pre_f16k == nullptr) || (res_ctot % 16 == 0 && res_ctot >= d->Cout), MASIC_ERR_SHAPE,
                  "conv_f16k_res_ex_fwd: residual / mask / pre tensors need >= Cout channels, a multiple of 16");
    return f16k_launch(x_f16k, w_packed, bias, nullptr, nullptr, 0, 0, nullptr, y_f16k, d, stream, nullptr, nullptr, 0.0f, res1, res2, res_ctot, nullptr, 0,
                       y_pre_f16k, mask, mask_slope);
}

// masic_conv_f16k_gdn_fwd with F16K output that ALSO stores the convolution's result before the GDN (y_pre_f16k, same layout as
// y_f16k): the training-mode forward keeps both for the backward pass (GDN backward needs its input, the next layer's weight
// gradient its output).
extern "C" int masic_conv_f16k_gdn_dual_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                                            void* y_pre_f16k, void* y_f16k, const masic_conv_desc_t* d, void* stream) {
    MASIC_REQUIRE(gdn_packed && y_pre_f16k && y_f16k, MASIC_ERR_ARG, "conv_f16k_gdn_dual_fwd: null pointer");
    return f16k_launch(x_f16k, w_packed, bias, nullptr, gdn_packed, gdn_inverse, 0, nullptr, y_f16k, d, stream, nullptr, nullptr, 0.0f, nullptr, nullptr, 0,
                       nullptr, 0, y_pre_f16k);
}

// A layer with few output channels (conv2 of Independent_EN: 96 -> 3, MASIC.py:1492-1496) on the MFMA path: `d` describes the
// convolution with its weight zero-padded to Cout = 32 (packed by the caller), y_nchw is [B][cout_store][Ho][Wo] float32 and
// receives channels 0 .. cout_store-1 of act(conv(x) + bias) + res32 (res32: float32 [B][cout_store][Ho][Wo] or NULL).
extern "C" int masic_conv_f16k_few_fwd(const void* x_f16k, const void* w_packed, const float* bias, const float* res32, float* y_nchw,
                                       int cout_store, const masic_conv_desc_t* d, void* stream) {
    MASIC_REQUIRE(d != nullptr && y_nchw != nullptr && cout_store > 0 && cout_store <= d->Cout && d->Cout == 32 && d->out_coff == 0, MASIC_ERR_ARG,
                  "conv_f16k_few_fwd: needs a 32-channel (zero-padded) descriptor and 0 < cout_store <= 32");
    return f16k_launch(x_f16k, w_packed, bias, nullptr, nullptr, 0, 0, y_nchw, nullptr, d, stream, nullptr, nullptr, 0.0f, nullptr, nullptr, 0, res32, cout_store);
}

// F16K in, F16K out (a channel view per d->out_ctot / out_coff) with up to two F16K residual tensors of res_ctot channels added
// after the activation: out = act(conv(x) + bias) + res1 [+ res2] -- the ResidualBlock / Enhancement_Block form of Independent_EN
// (compressai/layers/layers.py:160-190, MASIC.py:149-164) with bf16 operands.
extern "C" int masic_conv_f16k_res_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                                       void* y_f16k, const masic_conv_desc_t* d, void* stream) {
    MASIC_REQUIRE(d != nullptr && y_f16k != nullptr, MASIC_ERR_ARG, "conv_f16k_res_fwd: null pointer");
    MASIC_REQUIRE(res1 != nullptr || res2 == nullptr, MASIC_ERR_ARG, "conv_f16k_res_fwd: res2 without res1");
    MASIC_REQUIRE(res1 == nullptr || (res_ctot % 16 == 0 && res_ctot >= d->Cout), MASIC_ERR_SHAPE, "conv_f16k_res_fwd: residual tensors need >= Cout channels, a multiple of 16");
    return f16k_launch(x_f16k, w_packed, bias, nullptr, nullptr, 0, 0, nullptr, y_f16k, d, stream, nullptr, nullptr, 0.0f, res1, res2, res_ctot);
}

// fp8 (e4m3) operand form (BASELINE configs[4]): d->prec = MASIC_PREC_FP8 -- the input is F8K, weights come from
// masic_conv_f8k_pack_weight, wscale[Cout] = that call's per-channel weight scales times the input tensor's scale.  With
// d->prec = MASIC_PREC_BF16 this is masic_conv_f16k_gdn_fwd plus the option of an fp8 output (y_f8k, quantised with
// out_inv_scale = 1 / scale of the produced tensor) for an fp8-operand consumer.  Exactly one of y_nchw / y_f16k / y_f8k.
extern "C" int masic_conv_f8k_fwd(const void* x, const void* w_packed, const float* wscale, const float* bias, const float* gate,
                                  const void* gdn_packed, int gdn_inverse, float* y_nchw, void* y_f16k, void* y_f8k, float out_inv_scale,
                                  const masic_conv_desc_t* d, void* stream) {
    MASIC_REQUIRE(d != nullptr, MASIC_ERR_ARG, "conv_f8k_fwd: null descriptor");
    MASIC_REQUIRE((d->prec == MASIC_PREC_FP8) == (wscale != nullptr), MASIC_ERR_ARG, "conv_f8k_fwd: wscale goes with fp8 operands (prec = MASIC_PREC_FP8)");
    MASIC_REQUIRE(y_f8k == nullptr || out_inv_scale > 0.0f, MASIC_ERR_ARG, "conv_f8k_fwd: an fp8 output needs out_inv_scale > 0");
    return f16k_launch(x, w_packed, bias, gate, gdn_packed, gdn_inverse, 0, y_nchw, y_f16k, d, stream, wscale, y_f8k, out_inv_scale);
}

extern "C" size_t masic_f8k_bytes(int B, int C, int HW) { return (size_t)B * round_up(C, 32) * HW; }

extern "C" int masic_conv_f8k_pack_weight(const float* w, void* w_packed, float* wscale, const masic_conv_desc_t* d, void* stream) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    MASIC_REQUIRE(w && w_packed && wscale && d->prec == MASIC_PREC_FP8, MASIC_ERR_ARG, "conv_f8k_pack_weight: null pointer or prec != MASIC_PREC_FP8");
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const F16kCfg c = choose_f16k(*d, g, np);
    MASIC_REQUIRE(c.ok, MASIC_ERR_UNSUPPORTED, "conv_f8k: layer shape has no fp8 configuration");
    hipLaunchKernelGGL(wscale_f8k_kernel, dim3(d->Cout), dim3(256), 0, (hipStream_t)stream, w, wscale, d->Cin, d->Cout, d->KH * d->KW, d->transposed);
    PackStreamArgs a{w, d->Cin, d->Cout, d->KH, d->KW, d->transposed, c.KS, c.T, c.Cin16 / c.KS, c.ncb, {}, {}};
    size_t tot = 0;
    for (int p = 0; p < np; ++p) {
        a.gs[p] = g[p];
        a.phase_offs[p] = c.phase_off[p];
        const size_t t = (size_t)c.stream_bytes[p] * c.ncb;
        tot = tot > t ? tot : t;
    }
    int nb = (int)((tot + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_f8k_stream_kernel, dim3(nb, np), dim3(256), 0, (hipStream_t)stream, a, (const float*)wscale, (unsigned char*)w_packed);
    return masic_launch_status("conv_f8k_pack_weight");
}

extern "C" int masic_conv_f16k_gdn_fwd(const void* x_f16k, const void* w_packed, const float* bias, const float* gate,
                                       const void* gdn_packed, int gdn_inverse, float* y_nchw, void* y_f16k,
                                       const masic_conv_desc_t* d, void* stream);

extern "C" int masic_conv_f16k_fwd(const void* x_f16k, const void* w_packed, const float* bias, const float* gate,
                                   float* y_nchw, void* y_f16k, const masic_conv_desc_t* d, void* stream) {
    return masic_conv_f16k_gdn_fwd(x_f16k, w_packed, bias, gate, nullptr, 0, y_nchw, y_f16k, d, stream);
}

extern "C" int masic_conv_f16k_gdn_fwd(const void* x_f16k, const void* w_packed, const float* bias, const float* gate,
                                       const void* gdn_packed, int gdn_inverse, float* y_nchw, void* y_f16k,
                                       const masic_conv_desc_t* d, void* stream) {
    return f16k_launch(x_f16k, w_packed, bias, gate, gdn_packed, gdn_inverse, 0, y_nchw, y_f16k, d, stream);
}

// ConvTranspose2d(Cin -> C, k5, s2, p2, output_padding 1) weight [Cin][C][5][5] -> the weight [32][Cin][3][3] (+ bias [32]) of the
// equivalent stride-1 3x3 convolution to 4C (<= 32) channels whose 2x2 depth-to-space is the transposed convolution's output:
// row 4c + 2ph + pw, tap (u, v) = W[:, c, ph + 2(2-u), pw + 2(2-v)], zero where that kernel index exceeds 4 (masic_amd/ops.py:
// deconv_s2_as_conv_weight is the torch-op form the inference path caches; a training step changes the weight every iteration).
namespace {
__global__ void d2s_weight_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wo, float* __restrict__ bo, int Cin, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 32) bo[idx] = (bias != nullptr && idx < 4 * C) ? bias[idx >> 2] : 0.0f;
    if (idx >= 32 * Cin * 9) return;
    const int v = idx % 3, u = (idx / 3) % 3, ci = (idx / 9) % Cin, row = idx / (9 * Cin);
    const int c = row >> 2, ph = (row >> 1) & 1, pw = row & 1;
    const int kh = ph + 2 * (2 - u), kw = pw + 2 * (2 - v);
    wo[idx] = (c < C && kh <= 4 && kw <= 4) ? w[(((size_t)ci * C + c) * 5 + kh) * 5 + kw] : 0.0f;
}
}  // namespace
extern "C" int masic_deconv_s2_as_conv_weight(const float* w, const float* bias, float* w_out, float* bias_out, int Cin, int C, void* stream) {
    MASIC_REQUIRE(w && w_out && bias_out && Cin > 0 && C > 0 && 4 * C <= 32, MASIC_ERR_ARG, "deconv_s2_as_conv_weight: null pointer or more than 8 output channels");
    const int total = 32 * Cin * 9;
    hipLaunchKernelGGL(d2s_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, bias, w_out, bias_out, Cin, C);
    return masic_launch_status("deconv_s2_as_conv_weight");
}

// A stride-2 transposed convolution to few channels (g_s_conv4: 128 -> 3, MASIC.py:550, :598) run as its equivalent
// stride-1 3x3 convolution to (4 phases x C) channels with a depth-to-space store: `d` describes that equivalent
// Conv2d (Cout = 32, weights re-laid out by the caller), y_nchw is [B][out_ctot][2 Ho][2 Wo], channels out_coff..out_coff+C-1.
extern "C" int masic_conv_f16k_d2s_fwd(const void* x_f16k, const void* w_packed, const float* bias, float* y_nchw, int C,
                                       int y_ctot, int y_coff, const masic_conv_desc_t* d, void* stream) {
    MASIC_REQUIRE(d && C > 0 && 4 * C <= 32 && d->Cout == 32 && d->stride == 1 && !d->transposed, MASIC_ERR_UNSUPPORTED,
                  "conv_f16k_d2s: needs a stride-1 convolution to 32 (padded) channels and C <= 8");
    MASIC_REQUIRE(y_coff >= 0 && y_coff + C <= y_ctot, MASIC_ERR_SHAPE, "conv_f16k_d2s: output channel view out of range");
    return f16k_launch(x_f16k, w_packed, bias, nullptr, nullptr, 0, C | (y_ctot << 8) | (y_coff << 20), y_nchw, nullptr, d, stream);
}

namespace {
int f16k_launch(const void* x_f16k, const void* w_packed, const float* bias, const float* gate,
                const void* gdn_packed, int gdn_inverse, int d2s, float* y_nchw, void* y_f16k,
                const masic_conv_desc_t* d, void* stream, const float* wscale, void* y_f8k, float out_inv_scale,
                const void* res1, const void* res2, int res_ctot, const float* res32, int cout_store, void* y_pre, const void* mask16, float mask_slope) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    const bool f8 = d->prec == MASIC_PREC_FP8;
    const int cblk = f8 ? 32 : 16;
    MASIC_REQUIRE(x_f16k && w_packed && ((y_nchw != nullptr) + (y_f16k != nullptr) + (y_f8k != nullptr) == 1), MASIC_ERR_ARG,
                  "conv_f16k_fwd: need input, weights and exactly one output");
    MASIC_REQUIRE(y_f8k == nullptr || (d->out_ctot % 32 == 0 && d->out_coff % 32 == 0 && gate == nullptr && d2s == 0), MASIC_ERR_SHAPE,
                  "conv_f16k: an fp8 output needs a 32-aligned channel view and no gate");
    MASIC_REQUIRE(!f8 || (wscale != nullptr && d2s == 0), MASIC_ERR_ARG, "conv_f16k: fp8 operands need the dequantisation scales");
    MASIC_REQUIRE(d->Cout > 32 || d->KH * d->KW > 1, MASIC_ERR_UNSUPPORTED, "conv_f16k: no 1x1 configuration");
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const F16kCfg c = choose_f16k(*d, g, np);
    MASIC_REQUIRE(c.ok, MASIC_ERR_UNSUPPORTED, "conv_f16k: layer shape has no F16K configuration");
    MASIC_REQUIRE(d->in_ctot % cblk == 0 && d->in_coff % cblk == 0, MASIC_ERR_SHAPE, "conv_f16k: input channel view must be 16-aligned (32 for fp8)");
    MASIC_REQUIRE(y_f16k == nullptr || (d->out_ctot % 16 == 0 && d->out_coff % 16 == 0 && d->Cout % 4 == 0 && (gate == nullptr || gdn_packed == nullptr)),
                  MASIC_ERR_SHAPE, "conv_f16k: F16K output needs a 16-aligned channel view and Cout % 4 == 0 (a gate only without the fused GDN)");
    MASIC_REQUIRE(gdn_packed == nullptr || (d->Cout == 128 && d->act == MASIC_ACT_NONE && gate == nullptr), MASIC_ERR_UNSUPPORTED,
                  "conv_f16k: the fused GDN needs Cout = 128, no activation and no gate");
    if (g[0].Hp <= 0 || g[0].Wp <= 0) return MASIC_OK;
    const int tiles_w = ceil_div(g[0].Wp, c.TW), ntiles = tiles_w * ceil_div(g[0].Hp, c.TH);
    F16kArgs a{(const unsigned short*)x_f16k, (const unsigned short*)w_packed, bias, gate, y_nchw, (unsigned short*)y_f16k,
               (const uint4*)gdn_packed, (const unsigned short*)res1, (const unsigned short*)res2, res_ctot, (unsigned short*)y_pre, g_f16k_stamps, (const unsigned short*)mask16, mask_slope, res32, cout_store > 0 ? cout_store : d->Cout, wscale, (unsigned char*)y_f8k, out_inv_scale, gdn_inverse, d2s & 0xff, d->in_ctot / cblk, d->in_coff / cblk, c.Cin16,
               d->Hi, d->Wi, d->Cout, d->Ho, d->Wo, d2s ? ((d2s >> 8) & 0xfff) : (cout_store > 0 ? cout_store : d->out_ctot), d2s ? (d2s >> 20) : d->out_coff,
               d->gate_ctot, d->gate_c, d->act,
               c.TW, c.TWlog, c.SR, c.TH, tiles_w, ntiles,
               c.PH, c.PW, c.PWh, c.NPIXp, c.PB,
               (unsigned)((0x100000000ull + c.PW - 1) / c.PW), (unsigned)((0x100000000ull + c.PWh - 1) / c.PWh),
               (unsigned)((0x100000000ull + c.NPIXp / 32 - 1) / (c.NPIXp / 32)),
               {c.phase_off[0], c.phase_off[1], c.phase_off[2], c.phase_off[3]},
               {c.stream_bytes[0], c.stream_bytes[1], c.stream_bytes[2], c.stream_bytes[3]}, geom_params(*d), np, d->B % 8 == 0, 1};
    const int msplit = f16k_msplit(*d, c, np, g, gdn_packed != nullptr, d2s != 0 || cout_store != 0 || y_f8k != nullptr);
    a.msplit = msplit;
    static const int exp_store = getenv("MASIC_F16K_STORE_EXP") ? atoi(getenv("MASIC_F16K_STORE_EXP")) : 0;
    a.exp_store = exp_store;
    dim3 grid(round_up(ntiles, 8) * np, c.ncb * msplit, d->B);
    hipStream_t st = (hipStream_t)stream;
#define F16K_LAUNCH_D(KSV, TV, DV, PSPV, LV, GDNV, NMV, NPV)                                                                   \
    do {                                                                                                             \
        auto kfn = conv_f16k<KSV, TV, DV, PSPV, LV, GDNV, NMV, NPV>;                                                         \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL(kfn, grid, dim3(512), c.lds_bytes, st, a);                                                \
    } while (0)
#define F16K_LAUNCH(KSV, TV, PSPV, LV, GDNV, NMV, NPV) F16K_LAUNCH_D(KSV, TV, F16K_D, PSPV, LV, GDNV, NMV, NPV)
    MASIC_REQUIRE(c.NP == 1 || (d->Cout <= 32) == (c.NP == 4), MASIC_ERR_UNSUPPORTED, "conv_f16k: tile configuration");
#define F8K_LAUNCH(KSV, TV, PSPV, LV, GDNV)                                                                          \
    do {                                                                                                             \
        auto kfn = conv_f16k<KSV, TV, F16K_D, PSPV, LV, GDNV, 4, 1, true>;                                           \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL(kfn, grid, dim3(512), c.lds_bytes, st, a);                                                \
    } while (0)
    if (f8) {
        if (c.KS == 1) {
            if (gdn_packed) F8K_LAUNCH(1, 4, 6, 1, true);
            else F8K_LAUNCH(1, 4, 6, 1, false);
        } else {
            if (gdn_packed) F8K_LAUNCH(2, 2, 4, 2, true);
            else F8K_LAUNCH(2, 2, 4, 2, false);
        }
    } else
    if (c.NP == 4) {
        F16K_LAUNCH(1, 2, 5, 2, false, 1, 4);
    } else if (c.NP == 2) {
        if (c.KS == 2) {
            if (gdn_packed) F16K_LAUNCH_D(2, 2, 2, 5, 1, true, 4, 2);
            else if (d->Cout <= 64) F16K_LAUNCH_D(2, 2, 2, 5, 1, false, 2, 2);       // 64 / 96 output channels (Independent_EN): 2 / 3 accumulator
            else if (d->Cout <= 96) F16K_LAUNCH_D(2, 2, 2, 5, 1, false, 3, 2);       // tiles per pixel sub-tile instead of 4 with idle MFMAs
            else F16K_LAUNCH_D(2, 2, 2, 5, 1, false, 4, 2);
        } else if (gdn_packed) F16K_LAUNCH(1, 2, 3, 2, true, 4, 2);
        else F16K_LAUNCH(1, 2, 3, 2, false, 4, 2);
    } else if (c.KS == 1 && c.T == 5) {
        if (gdn_packed) F16K_LAUNCH_D(1, 5, 2, 6, 1, true, 4, 1);
        else if (msplit == 4) F16K_LAUNCH_D(1, 5, 2, 6, 1, false, 1, 1);
        else if (msplit == 2) F16K_LAUNCH_D(1, 5, 2, 6, 1, false, 2, 1);
        else F16K_LAUNCH_D(1, 5, 2, 6, 1, false, 4, 1);
    } else if (c.KS == 1) {
        if (gdn_packed) F16K_LAUNCH(1, 4, 6, 1, true, 4, 1);
        else if (msplit == 4) F16K_LAUNCH(1, 4, 6, 1, false, 1, 1);
        else if (msplit == 2) F16K_LAUNCH(1, 4, 6, 1, false, 2, 1);
        else F16K_LAUNCH(1, 4, 6, 1, false, 4, 1);
    } else if (c.D == 1) {
        MASIC_REQUIRE(gdn_packed == nullptr && d->Cout <= 96, MASIC_ERR_UNSUPPORTED, "conv_f16k: lite configuration with GDN / > 96 channels");
        if (d->Cout <= 32) F16K_LAUNCH_D(2, 2, 1, 3, 1, false, 1, 1);
        else if (d->Cout <= 64) F16K_LAUNCH_D(2, 2, 1, 3, 1, false, 2, 1);
        else F16K_LAUNCH_D(2, 2, 1, 3, 1, false, 3, 1);
    } else {
        if (gdn_packed) F16K_LAUNCH(2, 2, 4, 2, true, 4, 1);
        else if (d->Cout <= 32 || msplit == 4) F16K_LAUNCH(2, 2, 4, 2, false, 1, 1);
        else if (msplit == 2) F16K_LAUNCH(2, 2, 4, 2, false, 2, 1);
        else F16K_LAUNCH(2, 2, 4, 2, false, 4, 1);
    }
#undef F16K_LAUNCH
#undef F16K_LAUNCH_D
#undef F8K_LAUNCH
    return masic_launch_status("conv_f16k_fwd");
}
}  // namespace

// First analysis layer: Conv2d(3 -> 128, k5, s2, p2) + GDN -> F16K (conv_a_gdn_f16k above).  x: float32 NCHW channel view.
extern "C" size_t masic_conv_a_packed_bytes(void) { return CA_WIMG_BYTES; }
extern "C" int masic_conv_a_pack_weight(const float* w, void* w_packed, void* stream) {
    MASIC_REQUIRE(w && w_packed, MASIC_ERR_ARG, "conv_a_pack_weight: null pointer");
    hipLaunchKernelGGL(pack_conv_a_kernel, dim3(4 * CA_KSTEPS * 64 / 256), dim3(256), 0, (hipStream_t)stream, w, (uint4*)w_packed);
    return masic_launch_status("conv_a_pack_weight");
}
extern "C" int masic_conv_a_gdn_fwd_ex(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                                       void* y_f16k, void* y_f8k, float out_inv_scale, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream);
namespace { int conv_a_launch(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse, void* y_f16k, void* y_f8k,
                              float out_inv_scale, void* y_pre, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream); }
// training-mode form: F16K output plus the pre-GDN result (see masic_conv_f16k_gdn_dual_fwd)
extern "C" int masic_conv_a_gdn_dual_fwd(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                                         void* y_pre_f16k, void* y_f16k, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream) {
    MASIC_REQUIRE(y_pre_f16k && y_f16k, MASIC_ERR_ARG, "conv_a_gdn_dual_fwd: null pointer");
    return conv_a_launch(x, w_packed, bias, gdn_packed, gdn_inverse, y_f16k, nullptr, 0.0f, y_pre_f16k, B, Hi, Wi, in_ctot, in_coff, stream);
}
extern "C" int masic_conv_a_gdn_fwd(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                                    void* y_f16k, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream) {
    return masic_conv_a_gdn_fwd_ex(x, w_packed, bias, gdn_packed, gdn_inverse, y_f16k, nullptr, 0.0f, B, Hi, Wi, in_ctot, in_coff, stream);
}
// the same with the result optionally written as F8K fp8 (y_f8k, quantised with out_inv_scale) for an fp8-operand second layer
extern "C" int masic_conv_a_gdn_fwd_ex(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse,
                                       void* y_f16k, void* y_f8k, float out_inv_scale, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream) {
    return conv_a_launch(x, w_packed, bias, gdn_packed, gdn_inverse, y_f16k, y_f8k, out_inv_scale, nullptr, B, Hi, Wi, in_ctot, in_coff, stream);
}
// the convolution alone (no GDN), F16K out: Conv2d(3 -> 128, k5, s2, p2) on a float32 NCHW channel view
extern "C" int masic_conv_a_fwd(const float* x, const void* w_packed, const float* bias, void* y_f16k, int B, int Hi, int Wi, int in_ctot, int in_coff,
                                void* stream) {
    MASIC_REQUIRE(y_f16k != nullptr, MASIC_ERR_ARG, "conv_a_fwd: null pointer");
    return conv_a_launch(x, w_packed, bias, nullptr, 0, y_f16k, nullptr, 0.0f, nullptr, B, Hi, Wi, in_ctot, in_coff, stream);
}
namespace {
int conv_a_launch(const float* x, const void* w_packed, const float* bias, const void* gdn_packed, int gdn_inverse, void* y_f16k, void* y_f8k,
                  float out_inv_scale, void* y_pre, int B, int Hi, int Wi, int in_ctot, int in_coff, void* stream) {
    MASIC_REQUIRE(x && w_packed && ((y_f16k != nullptr) != (y_f8k != nullptr)), MASIC_ERR_ARG, "conv_a_gdn_fwd: null pointer / exactly one output");
    MASIC_REQUIRE(gdn_packed != nullptr || (y_f16k != nullptr && y_pre == nullptr), MASIC_ERR_ARG, "conv_a_fwd: without a GDN the output is one F16K tensor");
    MASIC_REQUIRE(y_f8k == nullptr || out_inv_scale > 0.0f, MASIC_ERR_ARG, "conv_a_gdn_fwd: an fp8 output needs out_inv_scale > 0");
    MASIC_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && in_coff >= 0 && in_coff + 3 <= in_ctot, MASIC_ERR_SHAPE, "conv_a_gdn_fwd: bad shape");
    const int Ho = (Hi + 4 - 5) / 2 + 1, Wo = (Wi + 4 - 5) / 2 + 1;
    const int tiles_w = ceil_div(Wo, 32), tiles_per_img = tiles_w * ceil_div(Ho, 8), ntiles = tiles_per_img * B;
    ConvAArgs a{x, (const uint4*)w_packed, bias, (const uint4*)gdn_packed, (unsigned short*)y_f16k, gdn_inverse, Hi, Wi, in_ctot, in_coff,
                Ho, Wo, tiles_w, tiles_per_img, ntiles, (unsigned char*)y_f8k, out_inv_scale, (unsigned short*)y_pre, g_f16k_stamps,
                getenv("MASIC_CONVA_STORE_AUX") ? atoi(getenv("MASIC_CONVA_STORE_AUX")) : 0};
    const size_t lds_bytes = 65536 + CA_WIMG_BYTES + 1024 + 2 * CA_PATCH_BYTES;
    const dim3 grid(ntiles < 256 ? ntiles : 256);
#define CONV_A_LAUNCH(X3V, OUTV)                                                                                                   \
    do {                                                                                                                           \
        auto kfn = conv_a_gdn_f16k<X3V, OUTV>;                                                                                     \
        static bool attr_set = false;                                                                                              \
        if (!attr_set) {                                                                                                           \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
            attr_set = true;                                                                                                       \
        }                                                                                                                          \
        hipLaunchKernelGGL(kfn, grid, dim3(512), lds_bytes, (hipStream_t)stream, a);                                               \
    } while (0)
    const bool x3 = (gdn_inverse & 2) != 0;
    const int out = y_f8k != nullptr ? 1 : (y_pre != nullptr ? 2 : 0);
    static const bool priv_on = !(getenv("MASIC_CONVA_PRIV") && getenv("MASIC_CONVA_PRIV")[0] == '0');      // 0: the shared-patch kernel (A/B timing)
    if (priv_on && gdn_packed != nullptr && !x3 && out == 0 && Wi % 4 == 0 && ((size_t)x & 15) == 0 && ((size_t)Hi * Wi) % 4 == 0) {
        static bool attr_w = false;
        if (!attr_w) {
            (void)hipFuncSetAttribute((const void*)conv_a_gdn_f16k_w, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_w = true;
        }
        hipLaunchKernelGGL(conv_a_gdn_f16k_w, grid, dim3(512), 65536 + CA_WIMG_BYTES + 1024 + 16 * CW_BYTES, (hipStream_t)stream, a);
        return masic_launch_status("conv_a_gdn_fwd");
    }
    if (gdn_packed == nullptr) CONV_A_LAUNCH(false, 3);
    else if (x3) { if (out == 0) CONV_A_LAUNCH(true, 0); else if (out == 1) CONV_A_LAUNCH(true, 1); else CONV_A_LAUNCH(true, 2); }
    else { if (out == 0) CONV_A_LAUNCH(false, 0); else if (out == 1) CONV_A_LAUNCH(false, 1); else CONV_A_LAUNCH(false, 2); }
#undef CONV_A_LAUNCH
    return masic_launch_status("conv_a_gdn_fwd");
}
}  // namespace

// ------------------------------------------------------------------------------------------ 1x1 layers as DMA-staged GEMMs
// The nine 1x1 (transposed) convolutions of each GMM head (MASIC.py:330-468) with the K-loop machinery of conv_f16k: the
// register-streamed kernel of gemm_bf16.hip waits an L2 round trip every other k-step; here both operands are staged by
// buffer_load ... lds three 64-channel chunks deep, with counted vmcnt and inline-asm LDS reads.
//   tile 128 co x 256 px, 8 waves, wave w = all 128 co of pixels 32w .. 32w+31 (4 accumulator tiles)
//   stage (48 KiB) = weights [k16 4][hh][co 128][8] (16 KiB, a linear copy of the packed stream) | activations [k16 4][hh][px 256][8]
//   waves 0-3 fetch the weights (4 wave-instructions per chunk each), waves 4-7 the activations (8 each)
namespace {

struct GemmF16kArgs {
    const unsigned short* x;      // F16K [B][Cin16][HW][16]
    const unsigned short* w;      // per 128-co block: [chunk][k16 4][hh][co 128][8]
    const float* bias;
    unsigned short* y16;          // F16K [B][out_ctot/16][HW][16] or null
    float* y32;                   // float32 NCHW view or null
    int Cin16, nchunks, Cout, HW, out_ctot, out_coff, act;
    const float* wscale;          // fp8 operands: per-output-channel dequantisation factor, else null
    unsigned char* y8;            // F8K output [B][out_ctot/32][HW][32] (quantised with out_inv_scale) or null
    float out_inv_scale;
};

// Up to three independent GEMMs over the same pixels in ONE launch (masic_gemm_f16k_group_fwd): the three entropy-parameter stacks
// of a GMM head (MASIC.py:330-468) are three 3-layer chains of 6 / 8 / 9 co-blocks x 32 pixel tiles each at 8 x 32 x 32 latents --
// 192 ... 288 workgroups on 256 CUs per launch, most of whose time is launch ramp, prologue and tail.  Layer i of the three stacks
// as one grid (blockIdx.y walks the co-blocks of group 0, then 1, then 2) is 20 ... 27 co-blocks = 640 ... 864 workgroups.
constexpr int GEMM_MAXG = 3;
struct GemmGroupArgs {
    GemmF16kArgs g[GEMM_MAXG];
    int cb_end[GEMM_MAXG];        // first co-block (blockIdx.y) past group i
    int n;
};

// fp8 operands: [co block][k32][hh][co 128][16 fp8] with per-output-channel scales ws[co] = max|W[co]| / 448
__global__ void wscale_gemm_f8k_kernel(const float* __restrict__ w, float* __restrict__ ws, int Cin, int Cout, int transposed) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= Cout) return;
    float m = 0.0f;
    for (int ci = 0; ci < Cin; ++ci) m = fmaxf(m, fabsf(transposed ? w[(size_t)ci * Cout + co] : w[(size_t)co * Cin + ci]));
    ws[co] = m > 0.0f ? m / 448.0f : 1.0f;
}
__global__ void pack_gemm_f8k_kernel(const float* __restrict__ w, const float* __restrict__ ws, unsigned char* __restrict__ wp, int Cin, int Cout,
                                     int nk32, int ncb, int transposed) {
    const size_t total = (size_t)ncb * nk32 * 4096;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = (int)(r & 15); r >>= 4;
        const int co = (int)(r & 127); r >>= 7;
        const int hh = (int)(r & 1); r >>= 1;
        const int k32 = (int)(r % nk32);
        const int cb = (int)(r / nk32);
        const int ci = k32 * 32 + hh * 16 + e, cog = cb * 128 + co;
        float v = 0.0f;
        if (ci < Cin && cog < Cout) v = (transposed ? w[(size_t)ci * Cout + cog] : w[(size_t)cog * Cin + ci]) / ws[cog];
        wp[i] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.0f, 0, false) & 0xff);
    }
}

__global__ void pack_gemm_f16k_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout, int nk16,
                                      int ncb, int transposed) {
    // [co block][k16][hh][co 128][8]: chunks of any KC consecutive k16 blocks are contiguous
    const int nchunks = nk16;
    const size_t total = (size_t)ncb * nk16 * 2048;                          // bf16 elements
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int co = (int)(r & 127); r >>= 7;
        const int hh = (int)(r & 1); r >>= 1;
        const int k16 = (int)(r % nchunks);
        const int cb = (int)(r / nchunks);
        const int ci = k16 * 16 + hh * 8 + e, cog = cb * 128 + co;
        float v = 0.0f;
        if (ci < Cin && cog < Cout) v = transposed ? w[(size_t)ci * Cout + cog] : w[(size_t)cog * Cin + ci];
        const __bf16 bv = (__bf16)v;
        wp[i] = __builtin_bit_cast(unsigned short, bv);
    }
}

// KC: 16-channel blocks per chunk.  KC = 2 keeps a stage at 24 KiB: two workgroups per CU, which is what a grid of
// 288 workgroups (nine 128-channel blocks x 32 pixel tiles) on 256 CUs needs.
// F8: fp8 operands (F8K input, 32-channel blocks; one MFMA = two consecutive blocks of the chunk, one per lane half)
template <int KC, bool F8 = false>
__global__ __launch_bounds__(512, KC == 2 ? 2 : 1) void gemm_f16k(const GemmGroupArgs ga) {
    static_assert(!F8 || KC % 2 == 0, "fp8 operands consume channel blocks in pairs");
    // group of this workgroup (uniform: scalar selects on blockIdx.y)
    GemmF16kArgs a = ga.g[0];
    int cb0 = 0;
    if (ga.n > 1 && (int)blockIdx.y >= ga.cb_end[0]) { a = ga.g[1]; cb0 = ga.cb_end[0]; }
    if (ga.n > 2 && (int)blockIdx.y >= ga.cb_end[1]) { a = ga.g[2]; cb0 = ga.cb_end[1]; }
    const int cblk = (int)blockIdx.y - cb0;                                  // co-block inside the group
    constexpr int ACT0 = KC * 4096, STAGE = KC * 12288, NS = 3;
    constexpr int WPW = KC, APW = 2 * KC;                                    // weight / activation DMA pieces per wave per chunk
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const bool wrole = wave < 4;
    const int wq = wave & 3;
    const int p0 = blockIdx.x * 256, m0 = cblk * 128, b = blockIdx.z;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (size_t)b * a.Cin16 * a.HW * 16), 0, a.Cin16 * a.HW * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.w + (size_t)cblk * a.nchunks * (KC * 2048)), 0, a.nchunks * (KC * 4096), 0x00020000);
    // activation pieces of this wave: piece I = k*4 + wq, k < 8: k16 = I >> 3, hh = (I >> 2) & 1, pixel block = I & 3 = wq
    const int px = wq * 64 + lane;
    const int avoff = p0 + px < a.HW ? px * 32 : 0x7ffffff0;                  // + hh*16 per piece
    auto issue = [&](int c, int stage) {
        unsigned char* st = lds + stage * STAGE;
        if (wrole) {
#pragma unroll
            for (int k = 0; k < WPW; ++k) dma_buf16(rw, st + (k * 4 + wq) * 1024, lane * 16, c * (KC * 4096) + (k * 4 + wq) * 1024);
        } else {
#pragma unroll
            for (int k = 0; k < APW; ++k) {
                const int k16 = k >> 1, hh = k & 1;                           // I = k*4 + wq -> (I >> 3, (I >> 2) & 1)
                dma_buf16(rx, st + ACT0 + k16 * 8192 + hh * 4096 + wq * 1024, avoff + hh * 16, ((c * KC + k16) * a.HW + p0) * 32);
            }
        }
    };
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
    issue(0, 0);
    issue(1, 1);
    if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APW) : "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned al = ldsb + (F8 ? j * 16 + h * 4096 : (h * 128 + j) * 16);
    const unsigned bl = ldsb + ACT0 + (F8 ? (wave * 32 + j) * 16 + h * 8192 : (h * 256 + wave * 32 + j) * 16);
    int cs = 0, ps = 2;                                                      // consumer / producer stage
    for (int c = 0; c < a.nchunks; ++c) {
        issue(c + 2, ps);                                                     // past the end: reads as zeros into a free stage
        ps = ps == NS - 1 ? 0 : ps + 1;
        const unsigned wa = al + cs * STAGE, ba = bl + cs * STAGE;
        constexpr int NQ = F8 ? 2 : 1, NKS = F8 ? KC / 2 : KC;
        v4u af[2][4][NQ], bfr[2][NQ];
        auto request = [&](auto ic, auto bc) {
            constexpr int i = decltype(ic)::value, buf = decltype(bc)::value;
            if constexpr (F8) {
                ds_read128<i * 16384>(bfr[buf][0], ba);
                ds_read128<i * 16384 + 4096>(bfr[buf][1], ba);
                static_for<0, 4>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    ds_read128<i * 8192 + m * 512>(af[buf][m][0], wa);
                    ds_read128<i * 8192 + m * 512 + 2048>(af[buf][m][1], wa);
                });
            } else {
                ds_read128<i * 8192>(bfr[buf][0], ba);
                static_for<0, 4>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    ds_read128<i * 4096 + m * 512>(af[buf][m][0], wa);
                });
            }
        };
        request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, NKS>([&](auto ic) {
            constexpr int i = decltype(ic)::value, buf = i & 1;
            if constexpr (i + 1 < NKS) request(std::integral_constant<int, i + 1>{}, std::integral_constant<int, (i + 1) & 1>{});
            if constexpr (F8) {
                lgkm_wait<(i + 1 < NKS ? 10 : 0)>(bfr[buf][0], bfr[buf][1], af[buf][0][0], af[buf][0][1], af[buf][1][0], af[buf][1][1],
                                                   af[buf][2][0], af[buf][2][1], af[buf][3][0], af[buf][3][1]);
                const v8i b8 = cat8(bfr[buf][0], bfr[buf][1]);
                static_for<0, 4>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    acc[m] = mfma_f8(cat8(af[buf][m][0], af[buf][m][1]), b8, acc[m]);
                });
            } else {
                lgkm_wait<(i + 1 < NKS ? 5 : 0)>(bfr[buf][0], af[buf][0][0], af[buf][1][0], af[buf][2][0], af[buf][3][0]);
                static_for<0, 4>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[buf][m][0]), __builtin_bit_cast(bf16x8, bfr[buf][0]), acc[m], 0, 0, 0);
                });
            }
        });
        cs = cs == NS - 1 ? 0 : cs + 1;
        if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");   // chunk c+1 has landed, chunk c+2 may stay in flight
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APW) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: (dequantisation) + bias + activation -> F16K, F8K or float32 NCHW view
    if constexpr (F8) {
        const float* sp = a.wscale + m0 + 4 * h;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m0 + m * 32 < a.Cout) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][e] *= sp[m * 32 + (e & 3) + 8 * (e >> 2)];
            }
    }
    if (a.bias != nullptr) {
        const float* bp = a.bias + m0 + 4 * h;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m0 + m * 32 < a.Cout) {
                float bv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) bv[e] = bp[m * 32 + (e & 3) + 8 * (e >> 2)];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][e] += bv[e];
            }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = apply_act(acc[m][e], a.act);
    const int p = p0 + wave * 32 + j;
    if (p >= a.HW) return;
    if (a.y32 != nullptr) {
        float* yb = a.y32 + ((size_t)b * a.out_ctot + a.out_coff + m0 + 4 * h) * a.HW + p;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m0 + m * 32 < a.Cout) {
#pragma unroll
                for (int e = 0; e < 16; ++e) yb[(unsigned)(m * 32 + (e & 3) + 8 * (e >> 2)) * (unsigned)a.HW] = acc[m][e];
            }
    } else if (a.y8 != nullptr) {
        const int c32 = (a.out_coff + m0) >> 5;
        unsigned char* yb = a.y8 + (((size_t)b * (a.out_ctot >> 5) + c32) * a.HW + p) * 32 + 16 * h;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m0 + m * 32 < a.Cout) store_f8k_tile(acc[m], yb + (size_t)m * a.HW * 32, a.out_inv_scale);
    } else {
        const int c16 = (a.out_coff + m0) >> 4;
        unsigned short* yb = a.y16 + (((size_t)b * (a.out_ctot >> 4) + c16) * a.HW + p) * 16 + 8 * h;
        const unsigned op16 = (unsigned)a.HW * 16;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m0 + m * 32 < a.Cout) store_f16k_tile(acc[m], yb + (size_t)(2 * m) * op16, op16);
    }
}


// ---- gemm_f16k2: the same GEMMs with 128 co x 64 px per WAVE (8 accumulator tiles) and the activations NOT staged through LDS.
// In gemm_f16k every wave owns all 128 output channels of only 32 pixels: each of its MFMAs needs a fresh weight fragment from LDS, so a
// k-step is 5 ds_read_b128 for 4 MFMAs -- 320 LDS cycles against 256 MFMA cycles per CU and k-step: the kernel was bound by its LDS reads
// (0.54 PFLOP/s alone), and a 32-channel chunk was only 8 MFMAs per wave and barrier.  Here:
//   * a wave holds 4 (co) x 2 (px) tiles: 4 weight fragments feed 8 MFMAs;
//   * its activation operand is a pixel's own F16K half-record -- 16 contiguous bytes per lane, 1 KiB per wave-instruction, nothing another
//     wave of the workgroup needs (all four waves share the CO block, not the pixels): it is loaded straight into registers
//     (buffer_load_dwordx4, one 64-channel chunk ahead), so LDS holds the weight ring only (3 x 16 KiB) and carries 4 reads per 8 MFMAs;
//   * a chunk is 64 channels = 32 MFMAs per wave and barrier; workgroup = 4 waves = 128 co x 256 px, two workgroups per CU (independent
//     barriers: one's prologue / epilogue under the other's K loop);
//   * per-iteration issue order [8 activation loads of chunk c+1][4 weight DMA pieces of chunk c+2], then s_waitcnt vmcnt(4): the
//     activations of the next chunk and the weights issued an iteration earlier have landed, the newest weight pieces stay in flight;
//   * workgroup n runs on XCD n % 8 (hardware round-robin): image b = n % 8 (+ 8 per pass), so an image's activations are fetched into ONE
//     XCD's L2, and inside an image the pixel tile is the fastest index: the workgroups resident together share few weight blocks.
constexpr int G2_KC = 4, G2_STAGE = G2_KC * 4096, G2_NS = 3;
template <int N>
__device__ __forceinline__ void vm_wait(v4u& a, v4u& b, v4u& c, v4u& d, v4u& e, v4u& f, v4u& g, v4u& h) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N) : "memory");
}
__device__ __forceinline__ void buf_load128(v4u& dst, unsigned voff, v4u rsrc, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__global__ __launch_bounds__(256, 2) void gemm_f16k2(const GemmGroupArgs ga, int B, int npx, int tiles_per_img) {
    // workgroup -> (image, co-block over all groups, pixel tile): workgroup n runs on XCD n % 8 (hardware round-robin), image b = n % 8 (+ 8
    // per pass), so an image's activations are fetched into ONE XCD's L2; inside an image the pixel tile is the fastest index.  Measured
    // (PMC FETCH_SIZE x 2, the 768 -> 3 x 1152 layer at 8 x 32 x 32): 97 MB fetched per launch against 204 MB for gemm_f16k's blockIdx.y-major
    // order (operands: 18 MB).  A 4 x 2 XCD grid (pixel quarter x co-block half, half of the weights resident per L2) fetched 109 MB and ran
    // within 3 % of this order: the launch is not bound by its fabric traffic.
    // (Fewer than 8 images -- the bitstream coder runs one: plain order, every workgroup live; the XCD order would park B of 8 XCDs' worth.)
    const int n = blockIdx.x, kx = B >= 8 ? n >> 3 : n % tiles_per_img;
    const int b = B >= 8 ? (n & 7) + 8 * ((n >> 3) / tiles_per_img) : n / tiles_per_img;
    if (b >= B) return;
    const int t = kx % tiles_per_img, pxt = t % npx, cbg = t / npx;
    GemmF16kArgs a = ga.g[0];
    int cb0 = 0;
    if (ga.n > 1 && cbg >= ga.cb_end[0]) { a = ga.g[1]; cb0 = ga.cb_end[0]; }
    if (ga.n > 2 && cbg >= ga.cb_end[1]) { a = ga.g[2]; cb0 = ga.cb_end[1]; }
    const int cblk = cbg - cb0;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int p0 = pxt * 256, m0 = cblk * 128;
    const int nchunks = a.nchunks;                                                 // 64-channel chunks (Cin padded to a multiple of 64 by the pack)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (size_t)cblk * nchunks * (G2_KC * 2048)), 0, nchunks * G2_STAGE, 0x00020000);
    // activation descriptor by hand (inline-asm loads take it as four SGPRs): base, stride 0, bytes, the flags make_buffer_rsrc is given above
    const unsigned long long xb = (unsigned long long)(a.x + (size_t)b * a.Cin16 * a.HW * 16);
    v4u rx;
    rx[0] = __builtin_amdgcn_readfirstlane((unsigned)xb);
    rx[1] = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32) & 0xffffu);
    rx[2] = __builtin_amdgcn_readfirstlane((unsigned)(a.Cin16 * a.HW * 32));
    rx[3] = 0x00020000u;
    unsigned bvoff[2];                                                             // the lane's half-record in k16 block 0, per pixel sub-tile
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int p = p0 + wave * 64 + q * 32 + j;
        bvoff[q] = p < a.HW ? (unsigned)p * 32u + 16u * h : 0xC0000000u;
    }
    const unsigned kstride = (unsigned)a.HW * 32u;                                 // bytes between k16 blocks
    auto issue_w = [&](int c) {                                                    // 16 KiB of weights: 4 pieces per wave
        unsigned char* st = lds + (c % G2_NS) * G2_STAGE;
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_buf16(rw, st + (k * 4 + wave) * 1024, lane * 16, c * G2_STAGE + (k * 4 + wave) * 1024);
    };
    v4u bq[2][G2_KC][2];                                                            // [parity][k16 of the chunk][pixel sub-tile]
    auto issue_b = [&](auto PC, int c) {
        constexpr int pc = decltype(PC)::value;
        static_for<0, G2_KC>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(c * G2_KC + i) * kstride);
            buf_load128(bq[pc][i][0], bvoff[0], rx, so);
            buf_load128(bq[pc][i][1], bvoff[1], rx, so);
        });
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][m][e] = 0.0f;
    // prologue: activations of chunk 0, weights of chunks 0 and 1
    issue_b(std::integral_constant<int, 0>{}, 0);
    issue_w(0);
    issue_w(1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                              // (the activation registers are claimed by the first body's wait below)
    __builtin_amdgcn_s_barrier();
    const unsigned al = ldsb + (h * 128 + j) * 16;
    auto body = [&](auto PC, int c) {
        constexpr int pc = decltype(PC)::value;
        issue_b(std::integral_constant<int, pc ^ 1>{}, c + 1);                     // past the end: out of range, reads as zeros
        issue_w(c + 2);
        const unsigned wa = al + (c % G2_NS) * G2_STAGE;
        v4u af[2][4];
        static_for<0, 4>([&](auto M) { constexpr int m = decltype(M)::value; ds_read128<m * 512>(af[0][m], wa); });
        static_for<0, G2_KC>([&](auto I) {
            constexpr int i = decltype(I)::value, fb = i & 1;
            if constexpr (i + 1 < G2_KC)
                static_for<0, 4>([&](auto M) { constexpr int m = decltype(M)::value; ds_read128<(i + 1) * 4096 + m * 512>(af[fb ^ 1][m], wa); });
            lgkm_wait<(i + 1 < G2_KC ? 4 : 0)>(af[fb][0], af[fb][1], af[fb][2], af[fb][3]);
            static_for<0, 2>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                const bf16x8 bv = __builtin_bit_cast(bf16x8, bq[pc][i][q]);
                static_for<0, 4>([&](auto M) {
                    constexpr int m = decltype(M)::value;
                    acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[fb][m]), bv, acc[q][m], 0, 0, 0);
                });
            });
        });
        // chunk c+1: its activations (issued above, the 8 oldest of this iteration) and its weights (issued one iteration ago) have landed when at
        // most the 4 newest weight pieces are still in flight
        vm_wait<4>(bq[pc ^ 1][0][0], bq[pc ^ 1][0][1], bq[pc ^ 1][1][0], bq[pc ^ 1][1][1], bq[pc ^ 1][2][0], bq[pc ^ 1][2][1], bq[pc ^ 1][3][0], bq[pc ^ 1][3][1]);
        __builtin_amdgcn_s_barrier();
    };
    {
        // claim chunk 0's activation registers (their loads completed with the prologue's wait: they are older than the weight pieces)
        vm_wait<4>(bq[0][0][0], bq[0][0][1], bq[0][1][0], bq[0][1][1], bq[0][2][0], bq[0][2][1], bq[0][3][0], bq[0][3][1]);
    }
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
        body(std::integral_constant<int, 0>{}, c);
        body(std::integral_constant<int, 1>{}, c + 1);
    }
    if (c < nchunks) body(std::integral_constant<int, 0>{}, c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: bias + activation -> F16K or float32 NCHW view
    if (a.bias != nullptr) {
        const float* bp = a.bias + m0 + 4 * h;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m0 + m * 32 < a.Cout) {
                float bv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) bv[e] = bp[m * 32 + (e & 3) + 8 * (e >> 2)];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[q][m][e] += bv[e];
            }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][m][e] = apply_act(acc[q][m][e], a.act);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int p = p0 + wave * 64 + q * 32 + j;
        if (p >= a.HW) continue;
        if (a.y32 != nullptr) {
            float* yb = a.y32 + ((size_t)b * a.out_ctot + a.out_coff + m0 + 4 * h) * a.HW + p;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m0 + m * 32 < a.Cout) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) yb[(unsigned)(m * 32 + (e & 3) + 8 * (e >> 2)) * (unsigned)a.HW] = acc[q][m][e];
                }
        } else {
            const int c16 = (a.out_coff + m0) >> 4;
            unsigned short* yb = a.y16 + (((size_t)b * (a.out_ctot >> 4) + c16) * a.HW + p) * 16 + 8 * h;
            const unsigned op16 = (unsigned)a.HW * 16;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m0 + m * 32 < a.Cout) store_f16k_tile(acc[q][m], yb + (size_t)(2 * m) * op16, op16);
        }
    }
}

}  // namespace

// Up to 18 weight packs in one launch (blockIdx.y = job): the nine 1x1 layers of an entropy-parameter head, each for the forward GEMM
// and -- transposed -- for the input-gradient GEMM of the training step, whose weights change with every optimizer step (one launch per
// pack was 36 launches of ~5 us per head pair and step).
namespace {
constexpr int GEMM_PACK_MAXJ = 18;
struct GemmPackJobs {
    const float* w[GEMM_PACK_MAXJ];
    unsigned short* wp[GEMM_PACK_MAXJ];
    int Cin[GEMM_PACK_MAXJ], Cout[GEMM_PACK_MAXJ], transposed[GEMM_PACK_MAXJ];
};
__global__ void pack_gemm_f16k_multi_kernel(const GemmPackJobs J) {
    const int jb = blockIdx.y;
    const float* __restrict__ w = J.w[jb];
    unsigned short* __restrict__ wp = J.wp[jb];
    const int Cin = J.Cin[jb], Cout = J.Cout[jb], transposed = J.transposed[jb];
    const int nk16 = ((Cin + 15) / 16 + 3) / 4 * 4, ncb = (Cout + 127) / 128;
    const size_t total = (size_t)ncb * nk16 * 2048;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int co = (int)(r & 127); r >>= 7;
        const int hh = (int)(r & 1); r >>= 1;
        const int k16 = (int)(r % nk16);
        const int cb = (int)(r / nk16);
        const int ci = k16 * 16 + hh * 8 + e, cog = cb * 128 + co;
        float v = 0.0f;
        if (ci < Cin && cog < Cout) v = transposed ? w[(size_t)ci * Cout + cog] : w[(size_t)cog * Cin + ci];
        const __bf16 bv = (__bf16)v;
        wp[i] = __builtin_bit_cast(unsigned short, bv);
    }
}
}  // namespace

// jobs: n (<= 18) packs as masic_gemm_f16k_pack_weight would make them, one launch: w[i] / wp[i] / Cin[i] / Cout[i] / transposed[i]
extern "C" int masic_gemm_f16k_pack_weights(const float* const* w, void* const* wp, const int* Cin, const int* Cout, const int* transposed, int n, void* stream) {
    MASIC_REQUIRE(w && wp && Cin && Cout && transposed && n >= 1 && n <= GEMM_PACK_MAXJ, MASIC_ERR_ARG, "gemm_f16k_pack_weights: 1 ... %d jobs", GEMM_PACK_MAXJ);
    GemmPackJobs J{};
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        MASIC_REQUIRE(w[i] && wp[i] && Cin[i] > 0 && Cout[i] > 0, MASIC_ERR_ARG, "gemm_f16k_pack_weights: bad job %d", i);
        J.w[i] = w[i]; J.wp[i] = (unsigned short*)wp[i]; J.Cin[i] = Cin[i]; J.Cout[i] = Cout[i]; J.transposed[i] = transposed[i];
        const size_t total = (size_t)ceil_div(Cout[i], 128) * round_up(ceil_div(Cin[i], 16), 4) * 2048;
        most = most > total ? most : total;
    }
    int nb = (int)((most + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(pack_gemm_f16k_multi_kernel, dim3(nb, n), dim3(256), 0, (hipStream_t)stream, J);
    return masic_launch_status("gemm_f16k_pack_weights");
}

extern "C" size_t masic_gemm_f16k_packed_bytes(int Cin, int Cout) {
    return (size_t)ceil_div(Cout, 128) * round_up(ceil_div(Cin, 16), 4) * 4096;
}

extern "C" int masic_gemm_f16k_pack_weight(const float* w, void* wp, int Cin, int Cout, int transposed, void* stream) {
    MASIC_REQUIRE(w && wp && Cin > 0 && Cout > 0, MASIC_ERR_ARG, "gemm_f16k_pack_weight: bad argument");
    const int nchunks = round_up(ceil_div(Cin, 16), 4), ncb = ceil_div(Cout, 128);          // k16 blocks, padded to a multiple of 4
    const size_t total = (size_t)ncb * nchunks * 2048;
    int nb = (int)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_gemm_f16k_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wp, Cin, Cout, nchunks, ncb, transposed);
    return masic_launch_status("gemm_f16k_pack_weight");
}

// y = act(W x + b), x in F16K with Cin a multiple of 16; output F16K (y_f16k: out_ctot channels, written at out_coff) or a
// float32 NCHW channel view (y_nchw).  Cout must be a multiple of 32.
extern "C" int masic_gemm_f16k_fwd(const void* x_f16k, const void* w_packed, const float* bias, void* y_f16k, float* y_nchw,
                                   int B, int Cin, int Cout, int HW, int out_ctot, int out_coff, int act, void* stream) {
    const masic_gemm_group_t g{x_f16k, w_packed, nullptr, bias, y_f16k, nullptr, y_nchw, 0.0f, Cin, Cout, out_ctot, out_coff, act};
    return masic_gemm_f16k_group_fwd(&g, 1, B, HW, stream);
}

// Up to three such layers over the same B x HW pixels in one launch (GemmGroupArgs above).  All groups bf16 operands (wscale NULL
// everywhere: x is F16K, Cin % 16 == 0) or all fp8 (wscale set everywhere: x is F8K, Cin % 32 == 0, weights from
// masic_gemm_f8k_pack_weight); per group exactly one of y_f16k / y_f8k / y_nchw.  Results are those of the single-layer calls, bit for bit.
extern "C" int masic_gemm_f16k_group_fwd(const masic_gemm_group_t* groups, int ngroups, int B, int HW, void* stream) {
    MASIC_REQUIRE(groups != nullptr && ngroups >= 1 && ngroups <= GEMM_MAXG, MASIC_ERR_ARG, "gemm_f16k_group_fwd: 1 ... %d groups", GEMM_MAXG);
    MASIC_REQUIRE(B > 0 && HW > 0, MASIC_ERR_SHAPE, "gemm_f16k_group_fwd: non-positive size");
    const bool f8 = groups[0].wscale != nullptr;
    const int cblk = f8 ? 32 : 16;
    GemmGroupArgs ga{};
    int ncb = 0;
    for (int i = 0; i < ngroups; ++i) {
        const masic_gemm_group_t& g = groups[i];
        MASIC_REQUIRE((g.wscale != nullptr) == f8, MASIC_ERR_ARG, "gemm_f16k_group_fwd: bf16 and fp8 groups cannot share a launch");
        MASIC_REQUIRE(g.x && g.w_packed && ((g.y_f16k != nullptr) + (g.y_f8k != nullptr) + (g.y_nchw != nullptr) == 1), MASIC_ERR_ARG,
                      "gemm_f16k_fwd: need input, weights and exactly one output");
        MASIC_REQUIRE(g.Cin > 0 && g.Cin % cblk == 0 && g.Cout > 0 && g.Cout % 32 == 0, MASIC_ERR_UNSUPPORTED,
                      "gemm_f16k_fwd: needs Cin %% %d == 0 and Cout %% 32 == 0", cblk);
        const int oalign = f8 ? 32 : 16;
        MASIC_REQUIRE(g.out_coff >= 0 && g.out_coff + g.Cout <= g.out_ctot && (g.y_nchw != nullptr || (g.out_ctot % oalign == 0 && g.out_coff % oalign == 0)),
                      MASIC_ERR_SHAPE, "gemm_f16k_fwd: output channel view");
        MASIC_REQUIRE(g.y_f8k == nullptr || g.out_inv_scale > 0.0f, MASIC_ERR_ARG, "gemm_f8k_fwd: an fp8 output needs out_inv_scale > 0");
        MASIC_REQUIRE((long)(g.Cin / cblk) * HW * 32 < (1l << 31), MASIC_ERR_UNSUPPORTED, "gemm_f16k_fwd: activation plane too large for 32-bit offsets");
        ncb += ceil_div(g.Cout, 128);
        ga.cb_end[i] = ncb;
    }
    ga.n = ngroups;
    dim3 grid(ceil_div(HW, 256), ncb, B);
    // bf16 operands: 32-channel chunks, 72 KiB of LDS -- two workgroups per CU, or one next to a workgroup of another stream's kernel.
    // The 64-channel form (144 KiB, a CU to itself) is as fast when the launch has the chip to itself, but inside the three-stream
    // forward it shuts the other streams' kernels out of its CUs: 2.54 -> 2.41 ms per step at 8 x 512 x 512 with KC = 2 everywhere
    // (A/B on one box, 100 replays each, twice).  fp8 operands: 128-channel chunks only.  MASIC_GEMM_KC = 4 restores the old rule
    // (64-channel chunks when the grid is a whole number of 256-workgroup rounds).
    const long blocks = (long)grid.x * grid.y * grid.z;
    static const int forced = getenv("MASIC_GEMM_KC") ? atoi(getenv("MASIC_GEMM_KC")) : 0;
    const bool kc2 = !f8 && (forced != 4 || blocks % 256 != 0 || blocks < 256);
    for (int i = 0; i < ngroups; ++i) {
        const masic_gemm_group_t& g = groups[i];
        const int cin_b = g.Cin / cblk, nkb = round_up(cin_b, 4);
        ga.g[i] = GemmF16kArgs{(const unsigned short*)g.x, (const unsigned short*)g.w_packed, g.bias, (unsigned short*)g.y_f16k, g.y_nchw,
                               cin_b, nkb / (kc2 ? 2 : 4), g.Cout, HW, g.out_ctot, g.out_coff, g.act, g.wscale, (unsigned char*)g.y_f8k, g.out_inv_scale};
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_f16k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_f16k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_f16k<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    static const bool v2 = !(getenv("MASIC_GEMM_V2") && getenv("MASIC_GEMM_V2")[0] == '0');      // 0: gemm_f16k (A/B timing)
    if (!f8 && v2) {
        bool ok = true;
        for (int i = 0; i < ngroups; ++i) ok = ok && groups[i].y_f8k == nullptr;
        if (ok) {
            for (int i = 0; i < ngroups; ++i) ga.g[i].nchunks = round_up(groups[i].Cin / 16, 4) / G2_KC;
            const int npx = ceil_div(HW, 256), tiles_per_img = npx * ncb;
            static bool attr2 = false;
            if (!attr2) {
                (void)hipFuncSetAttribute((const void*)gemm_f16k2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr2 = true;
            }
            hipLaunchKernelGGL(gemm_f16k2, dim3(B >= 8 ? 8 * tiles_per_img * ceil_div(B, 8) : B * tiles_per_img), dim3(256), G2_NS * G2_STAGE, (hipStream_t)stream, ga, B, npx, tiles_per_img);
            return masic_launch_status("gemm_f16k_fwd");
        }
    }
    if (f8) hipLaunchKernelGGL((gemm_f16k<4, true>), grid, dim3(512), 3 * 4 * 12288, (hipStream_t)stream, ga);
    else if (kc2) hipLaunchKernelGGL(gemm_f16k<2>, grid, dim3(512), 3 * 2 * 12288, (hipStream_t)stream, ga);
    else hipLaunchKernelGGL(gemm_f16k<4>, grid, dim3(512), 3 * 4 * 12288, (hipStream_t)stream, ga);
    return masic_launch_status("gemm_f16k_fwd");
}

// fp8 (e4m3) operand form of masic_gemm_f16k_*: x in F8K (Cin % 32 == 0), weights from masic_gemm_f8k_pack_weight (which also
// writes the per-output-channel weight scales); wscale[Cout] = those scales x the input tensor's scale.  Exactly one output:
// y_f16k (bf16, next layer runs bf16 operands), y_f8k (fp8, quantised with out_inv_scale) or y_nchw (float32 view).
extern "C" size_t masic_gemm_f8k_packed_bytes(int Cin, int Cout) {
    return (size_t)ceil_div(Cout, 128) * round_up(ceil_div(Cin, 32), 4) * 4096;
}

extern "C" int masic_gemm_f8k_pack_weight(const float* w, void* wp, float* wscale, int Cin, int Cout, int transposed, void* stream) {
    MASIC_REQUIRE(w && wp && wscale && Cin > 0 && Cout > 0, MASIC_ERR_ARG, "gemm_f8k_pack_weight: bad argument");
    const int nk32 = round_up(ceil_div(Cin, 32), 4), ncb = ceil_div(Cout, 128);
    hipLaunchKernelGGL(wscale_gemm_f8k_kernel, dim3(ceil_div(Cout, 128)), dim3(128), 0, (hipStream_t)stream, w, wscale, Cin, Cout, transposed);
    const size_t total = (size_t)ncb * nk32 * 4096;
    int nb = (int)((total + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_gemm_f8k_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (const float*)wscale, (unsigned char*)wp, Cin, Cout, nk32, ncb, transposed);
    return masic_launch_status("gemm_f8k_pack_weight");
}

extern "C" int masic_gemm_f8k_fwd(const void* x_f8k, const void* w_packed, const float* wscale, const float* bias, void* y_f16k, void* y_f8k,
                                  float* y_nchw, float out_inv_scale, int B, int Cin, int Cout, int HW, int out_ctot, int out_coff, int act, void* stream) {
    MASIC_REQUIRE(wscale != nullptr, MASIC_ERR_ARG, "gemm_f8k_fwd: need input, weights, scales and exactly one output");
    const masic_gemm_group_t g{x_f8k, w_packed, wscale, bias, y_f16k, y_f8k, y_nchw, out_inv_scale, Cin, Cout, out_ctot, out_coff, act};
    return masic_gemm_f16k_group_fwd(&g, 1, B, HW, stream);
}

// Diagnostics: a device buffer of 16 uint64 that every following conv_f16k launch fills with {s_memtime (core clock), s_memrealtime
// (100 MHz)} pairs at kernel entry, K-loop entry, K-loop exit and kernel exit of its first [0..7] and last [8..15] workgroup;
// NULL switches it off.  tools/f16k_stamps.py reads the phases and the shader clock the chip held from them.
extern "C" void masic_conv_f16k_set_stamps(void* device_buffer) { g_f16k_stamps = (unsigned long long*)device_buffer; }

// ------------------------------------------------------------------------------------------ 3x3 layers with resident weights
// Conv2d(CI -> CO, k3, s1, p1) (or its input gradient) on F16K for CI in {16, 32, 64} (16: a 3- or 6-channel picture zero-padded to
// one record per pixel), CO in {32, 64}: the 32- and 64-channel stages of Independent_EN and its two input layers (reference
// MASIC.py:149-164, 1456-1482 -- 28 of its 40 convolutions, and the input gradients of 24 of them in training).  K = 288 is 18 MFMA k-steps: on
// conv_f16k such a layer is three quarters per-workgroup prologue / epilogue (171 us per launch at 8 x 512^2 against an HBM floor of
// ~45).  Here the whole weight tensor (18 KiB of bf16 fragments) stays in LDS for the life of a PERSISTENT workgroup that walks
// over 16 x 32-pixel tiles of one XCD's share of the images:
//   * waves 8-11 are loaders: per tile they issue the 39 `buffer_load ... lds` pieces of the (18 x 34 pixel x 32 channel) patch two
//     tiles ahead into a ring of three buffers (10 pieces each: a single wave needs 60-185 cycles per piece, 121 -> see DESIGN.md us
//     per launch with one loader) and wait -- with a counted vmcnt, nothing else is on their counter -- for the patch of the
//     NEXT tile; padding pixels are out-of-range buffer offsets (zero fill);
//   * waves 0-7 own two image rows of the tile each: 18 k-steps x (1 weight + 2 patch fragments by ds_read_b128, 2 MFMAs), then
//     the F16K epilogue of conv_f16k (bias, activation, act'(mask), pre-residual copy, residual adds, 16-byte stores);
//   * ONE barrier per tile hands the landed patch to the compute waves and the drained buffer back to the loader.
// HBM-bound by design: 39 KiB in + 32 KiB out (+ residuals) per tile against 36 MFMAs per wave.
struct C3Args {
    const unsigned short* x; const unsigned short* w; const float* bias;
    const unsigned short* res1; const unsigned short* res2; const unsigned short* mask16; unsigned short* y_pre; unsigned short* y16;
    int B, H, W, in_ctot16, in_c16off, out_ctot, out_coff, res_ctot, act;
    float mask_slope;
    int tiles_w, tiles_per_image, ntiles;
    // backward chains of a residual block (masic_amd/autograd.py: EnhancementBlockFn): a SECOND output y2 = y * act'(mask2) -- the next
    // layer's dy, which a separate elementwise pass used to make -- and per-channel sums (the bias gradient) of the value before the
    // residual adds (sum_of = 1) or of y2 (sum_of = 2), bf16-rounded as stored, as per-wave partials [gridDim.x * 8][32]
    const unsigned short* mask2; unsigned short* y2; float mask2_slope;
    float* sum_part; int sum_of;
};

namespace {

// s_barrier that the compiler does not move LDS accesses across (the intrinsic alone is "no memory"); no hardware wait is added:
// what crosses it here was waited for explicitly (the loader's vmcnt) or consumed by MFMAs (the compute waves' fragment reads)
__device__ __forceinline__ void wg_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int CI, int C, bool EX = false>      // input channels (a multiple of 16), output channels; EX: the second output / channel sums of C3Args
__global__ __launch_bounds__(768, 1) void conv3x3_resident_f16k(const C3Args a) {
    constexpr int KB = CI / 16;                        // 16-channel blocks = 32-byte records per input pixel
    constexpr int NM = C / 32;                         // 32-channel accumulator tiles per pixel; a wave owns one of them for two image rows
    constexpr int TH = 16 / NM, TW = 32, PWd = TW + 2, NPIX = (TH + 2) * PWd;
    constexpr int PREC = KB * 2 * NPIX;                // 16-byte records of a patch: [kb][k-half][pixel]
    constexpr int NI = (PREC + 63) / 64;               // DMA wave-instructions per patch
    constexpr int PBYTES = NI * 1024;
    constexpr int SLAB = 2 * C * 16;                   // one (tap, kb) weight slab: [k-half][co]
    constexpr int WBYTES = 9 * KB * SLAB;
    constexpr int NWI = WBYTES / 1024;
    constexpr int NBUF = WBYTES + 3 * PBYTES + 1024 <= 160 * 1024 ? 3 : 2;   // patch ring (64 -> 64: 72 KiB of weights leave room for two 43-KiB patches)
    constexpr int PD = NBUF - 1;                       // tiles of look-ahead
    constexpr int NLW = 4;                             // loader waves (8 .. 11): one wave issues a 1-KiB piece every 60-185 cycles
    constexpr int NIW = (NI + NLW - 1) / NLW;          // pieces per loader wave and patch (the last wave pads with sink pieces)
    constexpr int NWW = (NWI + NLW - 1) / NLW;         // weight pieces per loader wave
    constexpr int SINK = WBYTES + NBUF * PBYTES;       // 1 KiB that padding pieces write zeros to
    static_assert((C == 32 || C == 64) && (CI == 16 || CI == 32 || CI == 64), "8 compute waves = (16 / NM row pairs) x NM channel tiles");
    static_assert(PD * NIW + NWW <= 60 && WBYTES % 1024 == 0 && SINK + 1024 <= 160 * 1024, "vmcnt is 6 bits; whole DMA pieces; LDS");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // [weights][NBUF patches][sink]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    // tiles of XCD k (workgroup ids go round-robin over the 8 XCDs) are one contiguous eighth of the tile list -- whole images when
    // B % 8 == 0 -- so the halos neighbouring tiles share are re-read from that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int per = (a.ntiles + 7) >> 3;
    const int t_end = (xcd + 1) * per < a.ntiles ? (xcd + 1) * per : a.ntiles;
    const int t_first = xcd * per + slot;
    const int n_my = t_first < t_end ? (t_end - t_first + nslot - 1) / nslot : 0;
    if (n_my == 0) {
        if (EX && a.sum_part != nullptr && wave < 8 && lane < 32) a.sum_part[((size_t)blockIdx.x * 8 + wave) * 32 + lane] = 0.0f;
        return;
    }
    const int HW = a.H * a.W;
    const int plane_bytes = HW * 32;

    if (wave >= 8) {
        // ---- loaders: wave 8 + lw takes pieces lw, lw + 4, ... of every patch
        const int lw = wave - 8;
        int packed[NIW];                               // plane << 16 | patch row << 8 | patch column of this lane's record in piece I
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            const int q = (k * NLW + lw) * 64 + lane;
            const int plane = q / NPIX, p = q - plane * NPIX;
            const int pr = p / PWd, pc = p - pr * PWd;
            packed[k] = q < PREC ? (plane << 16 | pr << 8 | pc) : -1;
        }
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, WBYTES, 0x00020000);
#pragma unroll
        for (int k = 0; k < NWW; ++k) {
            const int piece = k * NLW + lw;
            dma_buf16(rw, lds + (piece < NWI ? piece * 1024 : SINK), piece < NWI ? lane * 16 : 0x7ffffff0, piece < NWI ? piece * 1024 : 0);
        }
        auto issue = [&](int i) {
            const int tile = t_first + i * nslot;
            const int b = tile / a.tiles_per_image, rem = tile - b * a.tiles_per_image;
            const int th_i = rem / a.tiles_w, tw_i = rem - th_i * a.tiles_w;
            const int ih0 = th_i * TH - 1, iw0 = tw_i * TW - 1;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.x + ((size_t)b * a.in_ctot16 + a.in_c16off) * (size_t)HW * 16), 0, KB * plane_bytes, 0x00020000);
            unsigned char* dst = lds + WBYTES + (i % NBUF) * PBYTES;
#pragma unroll
            for (int k = 0; k < NIW; ++k) {
                const int pk = packed[k], piece = k * NLW + lw;
                const int ih = ih0 + ((pk >> 8) & 0xff), iw = iw0 + (pk & 0xff);
                const bool ok = pk >= 0 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                const int plane = pk >> 16;
                const int voff = ok ? (plane >> 1) * plane_bytes + (ih * a.W + iw) * 32 + (plane & 1) * 16 : 0x7ffffff0;
                dma_buf16(rx, piece < NI ? dst + piece * 1024 : lds + SINK, voff, 0);
            }
        };
        issue(0);
        if (PD == 2 && n_my > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");       // weights + tile 0 have landed, tile 1 may be in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        wg_barrier();
        for (int i = 0; i < n_my; ++i) {
            // the buffer of tile i + PD is the one tile i - 1 was read from: every compute wave left it before the last barrier
            if (i + PD < n_my) {
                issue(i + PD);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NIW) : "memory");   // tile i + 1 has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wg_barrier();
        }
        return;
    }

    // ---- compute waves: wave w owns channel tile w % NM for rows 2 (w / NM), 2 (w / NM) + 1 of the tile (two 32 x 32 accumulator tiles)
    const int mt = wave % NM, row0 = 2 * (wave / NM);
    const unsigned char* wa = lds + (h * C + mt * 32 + j) * 16;                         // lane part of the weight fragments
    const int pbl = WBYTES + (h * NPIX + row0 * PWd + j) * 16;                          // lane part of the patch fragments
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = a.bias != nullptr ? a.bias[mt * 32 + 4 * h + (e & 3) + 8 * (e >> 2)] : 0.0f;
    float bs[EX ? 16 : 1];                                              // per-lane channel sums over all tiles of this workgroup (sum_of)
#pragma unroll
    for (int e = 0; e < (EX ? 16 : 1); ++e) bs[e] = 0.0f;
    wg_barrier();                                                       // weights + tile 0 are in LDS
    for (int i = 0; i < n_my; ++i) {
        const unsigned char* pb = lds + pbl + (i % NBUF) * PBYTES;
        const int tile = t_first + i * nslot;
        const int b = tile / a.tiles_per_image, rem = tile - b * a.tiles_per_image;
        const int th_i = rem / a.tiles_w, tw_i = rem - th_i * a.tiles_w;
        const unsigned op16 = (unsigned)HW * 16;
        const size_t opix0 = (size_t)(th_i * TH + row0) * a.W + tw_i * TW + j;
        const size_t ro0 = (((size_t)b * (a.res_ctot >> 4) + 2 * mt) * HW + opix0) * 16 + 4 * h;      // row n: + n * W * 16
        // residual / mask operands of the epilogue are requested BEFORE the MFMAs (their HBM latency runs under the tile's work):
        // lane (j, h) needs 4 consecutive bf16 at element 8(q & 1) + 4h of record q >> 1, q = 0..3 (add_f16k_residual)
        uint2 rv1[2][4], rv2[2][4], mv[2][4];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t o = ro0 + (size_t)n * a.W * 16 + (size_t)(q >> 1) * op16 + 8 * (q & 1);
                if (a.res1 != nullptr) rv1[n][q] = *reinterpret_cast<const uint2*>(a.res1 + o);
                if (a.res2 != nullptr) rv2[n][q] = *reinterpret_cast<const uint2*>(a.res2 + o);
                if (!EX && a.mask16 != nullptr) mv[n][q] = *reinterpret_cast<const uint2*>(a.mask16 + o);      // (EX: read in the epilogue -- registers)
            }
        f32x16 acc[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int ty = t / 3, tx = t - 3 * ty;
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(wa + (t * KB + kb) * SLAB);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bf16x8 bw = *reinterpret_cast<const bf16x8*>(pb + (kb * 2 * NPIX + (ty + n) * PWd + tx) * 16);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc[n], 0, 0, 0);
                }
            }
        }
        // ---- epilogue (the F16K store path of conv_f16k for one 32-channel tile)
        auto add4 = [](f32x16& tt, int q, const uint2 r) {
            tt[4 * q + 0] += __builtin_bit_cast(float, r.x << 16);
            tt[4 * q + 1] += __builtin_bit_cast(float, r.x & 0xffff0000u);
            tt[4 * q + 2] += __builtin_bit_cast(float, r.y << 16);
            tt[4 * q + 3] += __builtin_bit_cast(float, r.y & 0xffff0000u);
        };
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const size_t opix = opix0 + (size_t)n * a.W;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = apply_act(acc[n][e] + bv[e], a.act);
            if (a.mask16 != nullptr) {
                if constexpr (EX) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) mv[n][q] = *reinterpret_cast<const uint2*>(a.mask16 + ro0 + (size_t)n * a.W * 16 + (size_t)(q >> 1) * op16 + 8 * (q & 1));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[n][4 * q + 0] *= __builtin_bit_cast(float, mv[n][q].x << 16) > 0.0f ? 1.0f : a.mask_slope;
                    acc[n][4 * q + 1] *= __builtin_bit_cast(float, mv[n][q].x & 0xffff0000u) > 0.0f ? 1.0f : a.mask_slope;
                    acc[n][4 * q + 2] *= __builtin_bit_cast(float, mv[n][q].y << 16) > 0.0f ? 1.0f : a.mask_slope;
                    acc[n][4 * q + 3] *= __builtin_bit_cast(float, mv[n][q].y & 0xffff0000u) > 0.0f ? 1.0f : a.mask_slope;
                }
            }
            if (a.y_pre != nullptr) store_f16k_tile(acc[n], a.y_pre + ro0 + (size_t)n * a.W * 16 + 4 * h, op16);
            if constexpr (EX) {
                if (a.sum_of == 1) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) bs[e] += (float)(__bf16)acc[n][e];
                }
            }
            if (a.res1 != nullptr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) add4(acc[n], q, rv1[n][q]);
            }
            if (a.res2 != nullptr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) add4(acc[n], q, rv2[n][q]);
            }
            store_f16k_tile(acc[n], a.y16 + (((size_t)b * (a.out_ctot >> 4) + (a.out_coff >> 4) + 2 * mt) * HW + opix) * 16 + 8 * h, op16);
            if constexpr (EX) {
                if (a.y2 != nullptr) {       // y2 = bf16(y) * act'(mask2): what the elementwise pass computed from the stored y (mask2 is read here,
                                             // not ahead of the MFMAs: the kernel has no registers left for a third prefetched operand)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint2 m2 = *reinterpret_cast<const uint2*>(a.mask2 + ro0 + (size_t)n * a.W * 16 + (size_t)(q >> 1) * op16 + 8 * (q & 1));
                        acc[n][4 * q + 0] = (float)(__bf16)acc[n][4 * q + 0] * (__builtin_bit_cast(float, m2.x << 16) > 0.0f ? 1.0f : a.mask2_slope);
                        acc[n][4 * q + 1] = (float)(__bf16)acc[n][4 * q + 1] * (__builtin_bit_cast(float, m2.x & 0xffff0000u) > 0.0f ? 1.0f : a.mask2_slope);
                        acc[n][4 * q + 2] = (float)(__bf16)acc[n][4 * q + 2] * (__builtin_bit_cast(float, m2.y << 16) > 0.0f ? 1.0f : a.mask2_slope);
                        acc[n][4 * q + 3] = (float)(__bf16)acc[n][4 * q + 3] * (__builtin_bit_cast(float, m2.y & 0xffff0000u) > 0.0f ? 1.0f : a.mask2_slope);
                    }
                    store_f16k_tile(acc[n], a.y2 + ro0 + (size_t)n * a.W * 16 + 4 * h, op16);
                    if (a.sum_of == 2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) bs[e] += (float)(__bf16)acc[n][e];
                    }
                }
            }
        }
        wg_barrier();          // tile i + 1 has landed (loader); this tile's buffer may be refilled
    }
    if (EX && a.sum_part != nullptr) {       // lanes of one half-wave hold the same 16 channels for 32 different pixels
#pragma unroll
        for (int e = 0; e < (EX ? 16 : 1); ++e) {
            float v = bs[e];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
            if (j == 0) a.sum_part[((size_t)blockIdx.x * 8 + wave) * 32 + 4 * h + (e & 3) + 8 * (e >> 2)] = v;
        }
    }
}

// out[c] = sum of the per-wave partials of conv3x3_resident_f16k (wave w of a workgroup holds channel tile w % NM): one block per channel,
// a lane takes every 64th partial, then a tree -- float64, fixed order
__global__ __launch_bounds__(64) void c3_sum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int nblocks) {
    const int c = blockIdx.x, NM = C / 32, mt = c / 32, cl = c - mt * 32, per = 8 / NM, K = nblocks * per;
    double s = 0.0;
    for (int k = threadIdx.x; k < K; k += 64) {
        const int b = k / per, w = mt + (k - b * per) * NM;
        s += (double)part[((size_t)b * 8 + w) * 32 + cl];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// weights of Conv2d(cin -> C, 3x3) [C][cin][3][3] float32 -> bf16 fragment slabs [tap][kb][k-half][co][8 k] with the input channels
// zero-padded to CI;  transposed: the slabs of the INPUT gradient of a Conv2d(C -> cin) layer (the stride-1 transposed convolution
// on the same tensor [cin][C][3][3]: taps mirrored, channel roles swapped)
__global__ __launch_bounds__(256) void pack_c3_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cin, int CI, int C, int transposed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                 // one 16-byte record: (tap, kb, hh, co)
    const int KB = CI / 16;
    if (idx >= 9 * KB * 2 * C) return;
    const int co = idx % C, hh = (idx / C) & 1, kb = (idx / (2 * C)) % KB, t = idx / (2 * C * KB);
    unsigned v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float f[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ci = kb * 16 + hh * 8 + 2 * p + q;
            f[q] = ci >= cin ? 0.0f : (transposed ? w[((size_t)ci * C + co) * 9 + (8 - t)] : w[((size_t)co * cin + ci) * 9 + t]);
        }
        v[p] = pack2bf(f[0], f[1]);
    }
    reinterpret_cast<uint4*>(out)[idx] = make_uint4(v[0], v[1], v[2], v[3]);
}

}  // namespace

static bool c3_pair_ok(int CI, int C) { return (C == 32 || C == 64) && (CI == 16 || CI == 32 || CI == 64) && (CI <= C || CI == C); }

extern "C" size_t masic_conv3x3_resident_packed_bytes(int Cin, int Cout) {
    const int CI = round_up(Cin, 16);
    return c3_pair_ok(CI, Cout) ? (size_t)9 * (CI / 16) * 2 * Cout * 16 : 0;
}

// Cin: real input channels (any value <= 64; the F16K input buffer holds round_up(Cin, 16) of them, the padding zero or anything
// finite: its weights are zero); Cout: 32 (needs H % 16 == 0) or 64 (H % 8 == 0); W % 32 == 0
extern "C" int masic_conv3x3_resident_supported(int B, int Cin, int Cout, int H, int W) {
    const int CI = round_up(Cin, 16);
    return B > 0 && Cin > 0 && c3_pair_ok(CI, Cout) && H % (512 / Cout) == 0 && W % 32 == 0 && (long)H * W * 32 * (CI / 16) < (1l << 31);
}

extern "C" int masic_conv3x3_resident_pack_weight(const float* w, void* w_packed, int Cin, int Cout, int transposed, void* stream) {
    const int CI = round_up(Cin, 16);
    MASIC_REQUIRE(w && w_packed && c3_pair_ok(CI, Cout), MASIC_ERR_UNSUPPORTED, "conv3x3_resident_pack_weight: Cin <= 64, Cout = 32 or 64");
    hipLaunchKernelGGL(pack_c3_weights_kernel, dim3(ceil_div(9 * (CI / 16) * 2 * Cout, 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)w_packed, Cin, CI,
                       Cout, transposed);
    return masic_launch_status("conv3x3_resident_pack_weight");
}

// y = act(conv3x3(x) + bias) * act'(mask) + res1 + res2 on F16K buffers, y_pre = the value before the adds (see
// masic_conv_f16k_res_ex_fwd: same meaning of every operand); x: channels [in_coff, in_coff + round_up(Cin, 16)) of an in_ctot-channel
// buffer, y: channels [out_coff, out_coff + Cout) of an out_ctot-channel buffer, residual / mask / pre tensors: res_ctot channels.
extern "C" int masic_conv3x3_resident_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                                          const void* mask, float mask_slope, void* y_pre_f16k, void* y_f16k, int B, int Cin, int Cout, int H, int W,
                                          int in_ctot, int in_coff, int out_ctot, int out_coff, int act, void* stream) {
    return masic_conv3x3_resident_ex_fwd(x_f16k, w_packed, bias, res1, res2, res_ctot, mask, mask_slope, y_pre_f16k, y_f16k, nullptr, 0.0f, nullptr, 0, nullptr,
                                         nullptr, B, Cin, Cout, H, W, in_ctot, in_coff, out_ctot, out_coff, act, stream);
}
extern "C" size_t masic_conv3x3_resident_sum_workspace_bytes(void) { return (size_t)256 * 8 * 32 * sizeof(float); }
// ... + y2 = y * act'(mask2) (res_ctot-channel tensors) and sum_out[Cout] = per-channel sums of the value before the residual adds
// (sum_of = 1) or of y2 (sum_of = 2), bf16-rounded as stored; sum_workspace: masic_conv3x3_resident_sum_workspace_bytes()
extern "C" int masic_conv3x3_resident_ex_fwd(const void* x_f16k, const void* w_packed, const float* bias, const void* res1, const void* res2, int res_ctot,
                                             const void* mask, float mask_slope, void* y_pre_f16k, void* y_f16k, const void* mask2, float mask2_slope,
                                             void* y2_f16k, int sum_of, float* sum_out, void* sum_workspace, int B, int Cin, int Cout, int H, int W,
                                             int in_ctot, int in_coff, int out_ctot, int out_coff, int act, void* stream) {
    const int CI = round_up(Cin, 16), C = Cout;
    MASIC_REQUIRE((mask2 == nullptr) == (y2_f16k == nullptr), MASIC_ERR_ARG, "conv3x3_resident_fwd: mask2 and y2 come together");
    MASIC_REQUIRE(sum_of >= 0 && sum_of <= 2 && (sum_of == 0) == (sum_out == nullptr) && (sum_of == 0 || sum_workspace != nullptr) && (sum_of != 2 || y2_f16k != nullptr),
                  MASIC_ERR_ARG, "conv3x3_resident_fwd: sum_of / sum_out / sum_workspace");
    MASIC_REQUIRE(y2_f16k == nullptr || (res_ctot % 16 == 0 && res_ctot >= C), MASIC_ERR_SHAPE, "conv3x3_resident_fwd: y2 / mask2 need res_ctot >= Cout channels");
    MASIC_REQUIRE(x_f16k && w_packed && y_f16k, MASIC_ERR_ARG, "conv3x3_resident_fwd: null pointer");
    MASIC_REQUIRE(masic_conv3x3_resident_supported(B, Cin, Cout, H, W), MASIC_ERR_UNSUPPORTED,
                  "conv3x3_resident_fwd: needs Cin <= 64, Cout = 32 (H %% 16 == 0) or 64 (H %% 8 == 0), W %% 32 == 0");
    MASIC_REQUIRE(in_ctot % 16 == 0 && in_coff % 16 == 0 && in_coff >= 0 && in_coff + CI <= in_ctot && out_ctot % 16 == 0 && out_coff % 16 == 0 &&
                      out_coff >= 0 && out_coff + C <= out_ctot, MASIC_ERR_SHAPE, "conv3x3_resident_fwd: channel views");
    MASIC_REQUIRE(res1 != nullptr || res2 == nullptr, MASIC_ERR_ARG, "conv3x3_resident_fwd: res2 without res1");
    MASIC_REQUIRE((res1 == nullptr && mask == nullptr && y_pre_f16k == nullptr) || (res_ctot % 16 == 0 && res_ctot >= C), MASIC_ERR_SHAPE,
                  "conv3x3_resident_fwd: residual / mask / pre tensors need >= Cout channels, a multiple of 16");
    MASIC_REQUIRE(act == MASIC_ACT_NONE || act == MASIC_ACT_RELU || act == MASIC_ACT_LEAKY, MASIC_ERR_UNSUPPORTED, "conv3x3_resident_fwd: activation");
    const int TH = 512 / C, tiles_w = W / 32, tiles_h = H / TH;
    C3Args a{(const unsigned short*)x_f16k, (const unsigned short*)w_packed, bias, (const unsigned short*)res1, (const unsigned short*)res2,
             (const unsigned short*)mask, (unsigned short*)y_pre_f16k, (unsigned short*)y_f16k, B, H, W, in_ctot / 16, in_coff / 16, out_ctot, out_coff,
             res_ctot, act, mask_slope, tiles_w, tiles_w * tiles_h, B * tiles_w * tiles_h,
             (const unsigned short*)mask2, (unsigned short*)y2_f16k, mask2_slope, (float*)sum_workspace, sum_of};
    // LDS: weights + patch ring + sink (the kernel's constants)
    const int wbytes = 9 * (CI / 16) * 2 * C * 16, pbytes = ((CI / 16) * 2 * (TH + 2) * 34 + 63) / 64 * 1024;
    const int lds_bytes = wbytes + (wbytes + 3 * pbytes + 1024 <= 160 * 1024 ? 3 : 2) * pbytes + 1024;
    int grid = 256;                                                  // one persistent workgroup per CU
    if (a.ntiles < grid) grid = round_up(a.ntiles, 8);
    hipStream_t st = (hipStream_t)stream;
#define C3_LAUNCH_E(CIV, CV, EXV)                                                                                             \
    do {                                                                                                                       \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            (void)hipFuncSetAttribute((const void*)conv3x3_resident_f16k<CIV, CV, EXV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
        hipLaunchKernelGGL((conv3x3_resident_f16k<CIV, CV, EXV>), dim3(grid), dim3(768), lds_bytes, st, a);                    \
    } while (0)
#define C3_LAUNCH(CIV, CV)                                                                                                    \
    do {                                                                                                                       \
        if (y2_f16k != nullptr || sum_of) C3_LAUNCH_E(CIV, CV, true);                                                          \
        else C3_LAUNCH_E(CIV, CV, false);                                                                                      \
    } while (0)
    if (C == 32) {
        if (CI == 16) C3_LAUNCH(16, 32);
        else C3_LAUNCH(32, 32);
    } else {
        if (CI == 16) C3_LAUNCH(16, 64);
        else if (CI == 32) C3_LAUNCH(32, 64);
        else C3_LAUNCH(64, 64);
    }
#undef C3_LAUNCH
#undef C3_LAUNCH_E
    if (sum_of) hipLaunchKernelGGL(c3_sum_finish_kernel, dim3(C), dim3(64), 0, st, (const float*)sum_workspace, sum_out, C, grid);
    return masic_launch_status("conv3x3_resident_fwd");
}
