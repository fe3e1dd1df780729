// wgrad_f16k.hip -- weight gradient of the 3x3 stride-1 convolutions of Independent_EN on F16K operands (gfx950 / MI355X).
//
// Reference: what torch autograd computes (convolution_backward) for the 36 `conv3x3` layers of the CQE network
// (coremasic/mywork/MASIC.py:149-164, :1436-1501; compressai/layers/layers.py:81-83, :160-190) in the training step of
// newtrain_cqe_real.py:128-174:
//     dW[co][ci][kh][kw] = sum_{b,r,c} dy[b][co][r][c] * x[b][ci][r + kh - 1][c + kw - 1]
// a GEMM per tap whose K dimension is the PIXEL index.  Round 1 ran it on the tap-generic float32-tile kernel
// (conv_wgrad.hip: conv_wgrad_f32<1, 3, true>): 1.8-2.8 ms per layer at 8 x 96 x 512 x 512, 37 % of the CQE training step.
//
// Here both operands are read in the layout the forward / input-gradient kernels already use, F16K = [B][C/16][H*W][16] bf16
// (half the bytes of the float32 NCHW tensors, no conversion pass), and the pixel-major records are turned into MFMA operands
// (8 consecutive k = pixels of one channel per lane) by the hardware transpose read `ds_read_b64_tr_b16`: a 16-lane group reads
// 4 records (pixels) x 16 channels and every lane receives its channel's 4 pixels.
//
// Workgroup = NQ x 3 waves: wave (qb, kh) owns the 32 input channels of block qb, kernel row kh, all MA 32-channel output
// blocks and the three taps of its row: MA x 3 accumulators of 32 x 32 (dW_t[co][ci]); per k-step of 16 pixels it reads MA
// A fragments (dy) and 3 B fragments (x at column shifts -1, 0, +1: the same records 32 bytes apart) for MA x 3 MFMAs.
// Pixel tiles of 4 rows x 32 columns (+ halo) are staged by `buffer_load ... lds` DMA, three buffers deep with a counted vmcnt
// (every wave issues the same number of DMA instructions per tile, unused slots go to a sink: the tile after next stays in flight
// across the barrier; padding pixels and ragged edges read as zeros through the buffer range check); the planes of consecutive 16-channel blocks sit 128 bytes
// (mod 256) apart so that the two 16-lane groups of a half wave never share a bank.  Workgroups take a strided share of the
// tiles and add their partial dW into a [tap][co][ci] workspace with float atomics (ci contiguous: full 128-byte segments);
// a last pass transposes it into the weight layout.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int TR = 4, TC = 32;                       // tile: output rows x columns
constexpr int GP = TR * TC * 32 + 128;               // bytes per 16-channel plane of the dy tile (+128: bank offset between planes)
template <int KS>                                    // x patch of a KS x KS layer: columns / rows with the halo
struct XGeom {
    static constexpr int XW = TC + KS - 1, XR = TR + KS - 1;
    static constexpr int XROW = XW * 32;             // bytes per patch row (k3: 1088, k5: 1152)
    static constexpr int XP0 = XR * XROW;
    static constexpr int XP = XP0 + (XP0 % 256 == 128 ? 0 : 128);   // bytes per 16-channel plane, = 128 mod 256 (k3: 6528, k5: 9216 + 128)
};

struct Wg3Args {
    const unsigned short* g16;    // dy  F16K [B][CA16][H*W][16]
    const unsigned short* x16;    // x   F16K [B][CQ16][H*W][16]
    float* ws;                    // [9][CA][CQ] float32, zeroed by the caller
    int B, H, W, CA16, CQ16, CA, CQ;
    int a0, q0;                   // first output / input channel of this launch's first channel group
    int sa, sq;                   // channel strides between the groups of a launch (k3: 96 / 64, k5: 64 / 32)
    int tiles_w, tiles_h, ntiles;
    int gq;                       // channel groups of this launch: blockIdx.y = ga * gq + gq_i, group (a0 + 96 ga, q0 + 64 gq_i) -- equally
                                  //   shaped groups share a launch (a 192 -> 192 layer: 6 groups, at latent resolution 64 workgroups each)
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

template <int OFF>
__device__ __forceinline__ void tr_read(v2u& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

__device__ __forceinline__ void depend(v2u& a, v2u& b) { asm volatile("" : "+v"(a), "+v"(b)); }   // uses of a, b stay behind the preceding wait

__device__ __forceinline__ bf16x8 frag(const v2u& lo, const v2u& hi) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}

template <int MA, int NQ, int KS>
__global__ __launch_bounds__(NQ * KS * 64) void wgrad3x3_f16k(const Wg3Args a) {
    constexpr int XR = XGeom<KS>::XR, XROW = XGeom<KS>::XROW, XP = XGeom<KS>::XP;
    constexpr int NW = NQ * KS;
    constexpr int GBYTES = 2 * MA * GP, XBYTES = 2 * NQ * XP, BUF = GBYTES + XBYTES;
    constexpr int NBUF = 3;
    constexpr int NI_G = 2 * MA * TR, NI_X = 2 * NQ * XR * 2, NI = NI_G + NI_X, NPW = (NI + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qb = wave / KS, kh = wave - KS * qb;
    const int HW = a.H * a.W;
    const int ga0 = a.a0 + a.sa * ((int)blockIdx.y / a.gq), gq0 = a.q0 + a.sq * ((int)blockIdx.y % a.gq);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.g16, 0, a.B * a.CA16 * HW * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x16, 0, a.B * a.CQ16 * HW * 32, 0x00020000);

    // ---- this wave's DMA instructions (tile independent part): instruction i = wave + j * NW
    //   i < NI_G: dy plane (block, row): 32 pixels x 32 bytes;  else x plane (block, patch row, part): part 0 = patch columns 0..31,
    //   part 1 = columns 32, 33 (lanes 0..3 only)
    int d_lds[NPW], d_blk[NPW], d_row[NPW], d_col[NPW];
    bool d_isx[NPW], d_live[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int i = wave + j * NW;
        d_live[j] = i < NI;
        d_isx[j] = i >= NI_G;
        if (!d_isx[j]) {
            d_blk[j] = i / TR;
            d_row[j] = i - d_blk[j] * TR;
            d_col[j] = lane >> 1;
            d_lds[j] = d_blk[j] * GP + d_row[j] * (TC * 32);
        } else {
            const int k = i - NI_G, part = k & 1, br = k >> 1;
            d_blk[j] = br / XR;
            d_row[j] = br - d_blk[j] * XR - KS / 2;                  // image row relative to the tile's first row
            d_col[j] = part * 32 + (lane >> 1) - KS / 2;             // image column relative to the tile's first column
            d_lds[j] = GBYTES + d_blk[j] * XP + (d_row[j] + KS / 2) * XROW + part * 1024;
            if (part == 1 && lane >= 2 * (KS - 1)) d_live[j] = false;
        }
    }
    // every wave issues exactly NPW DMA instructions per call (dead slots and calls past the last tile read out of range into
    // the sink), so `s_waitcnt vmcnt(NPW)` always means "everything but the most recent call has landed"
    auto issue = [&](int tile, int buf) {
        const bool real = tile < a.ntiles;
        const int tt = real ? tile : 0;
        const int b = tt / (a.tiles_w * a.tiles_h), t = tt - b * (a.tiles_w * a.tiles_h);
        const int r0 = (t / a.tiles_w) * TR, c0 = (t % a.tiles_w) * TC;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int i = wave + j * NW;
            const int r = r0 + d_row[j], c = c0 + d_col[j];
            const bool ok = real && r >= 0 && r < a.H && c >= 0 && c < a.W;
            const int voff = ok ? (r * a.W + c) * 32 + (lane & 1) * 16 : 0x7ffffff0;
            if (!real || i >= NI) {
                dma16(rg, lds + NBUF * BUF, 0x7ffffff0, 0);
            } else if (d_isx[j]) {
                if (d_live[j]) dma16(rx, lds + buf * BUF + d_lds[j], voff, ((b * a.CQ16 + (gq0 >> 4) + d_blk[j]) * HW) * 32);
            } else {
                dma16(rg, lds + buf * BUF + d_lds[j], voff, ((b * a.CA16 + (ga0 >> 4) + d_blk[j]) * HW) * 32);
            }
        }
    };

    f32x16 acc[MA][KS];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.0f;

    // lane part of the transposed-read addresses: 16-lane group g4 = (channel half, k half); lane 4q + p of a group supplies
    // row (pixel) q, columns (channels) 4p .. 4p+3 of its 4 x 16 block
    const int g4 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned la = ldsb + (g4 & 1) * GP + ((8 * (g4 >> 1) + qq) * 32) + 8 * pp;                                          // + ma * 2 * GP
    const unsigned lb = ldsb + GBYTES + (2 * qb + (g4 & 1)) * XP + kh * XROW + ((8 * (g4 >> 1) + qq) * 32) + 8 * pp;        // + kw * 32

    int tile = blockIdx.x, buf = 0;
    issue(tile, 0);
    issue(tile + gridDim.x, 1);
    for (; tile < a.ntiles; tile += gridDim.x, buf = buf == NBUF - 1 ? 0 : buf + 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");   // this tile's records have landed (the next tile's may still fly) ...
        __builtin_amdgcn_s_barrier();                                // ... for every wave, and everyone is done reading the buffer of the tile before
        issue(tile + 2 * gridDim.x, buf >= 1 ? buf - 1 : NBUF - 1);  // two tiles ahead, into that free buffer
        const unsigned ba = la + buf * BUF, bb = lb + buf * BUF;
        v2u af[2][MA][2], bfr[2][KS][2];
        auto request = [&](auto kc, auto pc) {
            constexpr int ks = decltype(kc)::value, pb = decltype(pc)::value;
            constexpr int rr = ks >> 1, ch = ks & 1;                 // k-step = 16 pixels: row rr of the tile, column half ch
            sfor<0, MA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                tr_read<m * 2 * GP + (rr * TC + ch * 16) * 32>(af[pb][m][0], ba);
                tr_read<m * 2 * GP + (rr * TC + ch * 16) * 32 + 128>(af[pb][m][1], ba);
            });
            sfor<0, KS>([&](auto tc) {
                constexpr int kw = decltype(tc)::value;
                tr_read<rr * XROW + (ch * 16 + kw) * 32>(bfr[pb][kw][0], bb);
                tr_read<rr * XROW + (ch * 16 + kw) * 32 + 128>(bfr[pb][kw][1], bb);
            });
        };
        request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        sfor<0, TR * 2>([&](auto kc) {
            constexpr int ks = decltype(kc)::value, pb = ks & 1;
            if constexpr (ks + 1 < TR * 2) request(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, (ks + 1) & 1>{});
            constexpr int pending = ks + 1 < TR * 2 ? 2 * (MA + KS) : 0;     // LDS returns in order: the requests for ks+1 may stay out
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(pending) : "memory");
            sfor<0, MA>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                depend(af[pb][m][0], af[pb][m][1]);
            });
            sfor<0, KS>([&](auto tc) {
                constexpr int kw = decltype(tc)::value;
                depend(bfr[pb][kw][0], bfr[pb][kw][1]);
            });
            sfor<0, KS>([&](auto tc) {
                constexpr int kw = decltype(tc)::value;
                const bf16x8 bq = frag(bfr[pb][kw][0], bfr[pb][kw][1]);
                sfor<0, MA>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    acc[m][kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(af[pb][m][0], af[pb][m][1]), bq, acc[m][kw], 0, 0, 0);
                });
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- reduce into ws[tap][co][ci]: accumulator row (co) = 8 (e >> 2) + 4 h + (e & 3), column (ci) = lane & 31
    const int j = lane & 31, h = lane >> 5;
    const int ci = gq0 + qb * 32 + j;
    if (ci < a.CQ) {
#pragma unroll
        for (int m = 0; m < MA; ++m)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                float* wp = a.ws + ((size_t)(kh * KS + kw) * a.CA + ga0 + m * 32 + 4 * h) * a.CQ + ci;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = ga0 + m * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
                    if (co < a.CA) atomicAdd(wp + (size_t)((e & 3) + 8 * (e >> 2)) * a.CQ, acc[m][kw][e]);
                }
            }
    }
}

// ws [T][CA][CQ] -> dw [CA][CQ][T]
__global__ __launch_bounds__(256) void wgrad3_transpose_kernel(float* __restrict__ ws, float* __restrict__ dw, int AQ, int T, int rezero) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)T * AQ; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i / AQ);
        const size_t aq = i - (size_t)t * AQ;
        dw[aq * T + t] = ws[i];
        if (rezero) ws[i] = 0.0f;          // a persistent workspace is clean again (masic_conv3x3_wgrad_f16k_ws)
    }
}

// ------------------------------------------------------------------------------------------ 1x1 layers
// dW[co][ci] = sum_{b,p} dy[b][co][p] x[b][ci][p] for the 1x1 layers of the entropy-parameter stacks (MASIC.py:330-468: 768 ... 1152
// channels on both sides at 8 x 32 x 32 latents -- a 1152 x 768 x 8192 GEMM whose contraction index is the pixel).  The float32 NCHW
// kernel (conv_wgrad.hip: conv_wgrad_1x1_bf16) re-reads its 128 x 64 float32 tiles 6 ... 9 times and is bound by that traffic
// (77 us per layer, 0.19 PFLOP/s); the training step has both operands in F16K anyway (the forward GEMM's input, the input-gradient
// GEMM's dy).  Workgroup = 4 waves, output tile 128 x 128 (wave = 64 co x 64 ci, 2 x 2 accumulators); k-tiles of 64 pixels of both
// operands -- 16 planes of 64 records -- arrive by `buffer_load ... lds` DMA, two buffers (70 KiB: two workgroups per CU), 8 DMA
// instructions per wave and tile with a counted vmcnt; fragments by `ds_read_b64_tr_b16` exactly as above (2 reads per MFMA).
// Workgroups take a strided share of the (image, pixel tile) pairs and add their partial tile into dw with float atomics.
constexpr int W1_PX = 64;                            // pixels per k-tile
constexpr int W1_PLANE = W1_PX * 32 + 128;           // bytes per 16-channel plane (+128: bank offset between the planes of a 32-channel block)
constexpr int W1_HALF = 8 * W1_PLANE;                // 128 channels of one operand
constexpr int W1_BUF = 2 * W1_HALF;                  // dy planes | x planes

struct Wg1Args {
    const unsigned short* g16;    // rows (A): F16K [B][CA16][HW][16]
    const unsigned short* x16;    // columns (B): F16K [B][CQ16][HW][16]
    float* dw;                    // [CA][CQ] float32, zeroed by the caller
    int B, HW, CA16, CQ16, CA, CQ;
    int q_tiles, tpi, ntk, nsplit;
    float* db;                    // bias_mode 1: [CA] sums of the rows operand over batch and pixels, 2: [CQ] sums of the columns operand
    int bias_mode;                //   (the bias gradient of the 1x1 layer: dy is the rows operand of a Conv2d, the columns operand of a
                                  //   ConvTranspose2d); zeroed by the caller; 0: none
};

__global__ __launch_bounds__(256, 2) void wgrad1x1_f16k(const Wg1Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wq = wave >> 1;
    const int a0 = (blockIdx.x / a.q_tiles) * 128, q0 = (blockIdx.x % a.q_tiles) * 128;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.g16, 0, a.B * a.CA16 * a.HW * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x16, 0, a.B * a.CQ16 * a.HW * 32, 0x00020000);
    // DMA instruction i = wave + 4 j, j < 8: operand i >> 4, plane (i >> 1) & 7, pixel half i & 1 (32 records of 32 bytes, 2 lanes each)
    auto issue = [&](int kt, int buf) {
        const bool real = kt < a.ntk;
        const int kk = real ? kt : 0;
        const int b = kk / a.tpi, p0 = (kk - b * a.tpi) * W1_PX;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = wave + 4 * j;
            const int isx = i >> 4, pl = (i >> 1) & 7, half = i & 1;
            const int px = p0 + half * 32 + (lane >> 1);
            const int blk = ((isx ? q0 : a0) >> 4) + pl;
            const bool ok = real && px < a.HW && blk < (isx ? a.CQ16 : a.CA16);
            const int voff = ok ? px * 32 + (lane & 1) * 16 : 0x7ffffff0;
            unsigned char* dst = lds + buf * W1_BUF + isx * W1_HALF + pl * W1_PLANE + half * 1024;
            if (isx) dma16(rx, dst, voff, ((b * a.CQ16 + blk) * a.HW) * 32);
            else dma16(rg, dst, voff, ((b * a.CA16 + blk) * a.HW) * 32);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][0][e] = 0.0f; acc[0][1][e] = 0.0f; acc[1][0][e] = 0.0f; acc[1][1][e] = 0.0f; }
    // channel sums of dy next to the weight gradient (its bias gradient; a separate reduction pass over dy cost 20-30 us per layer):
    // one more MFMA per fragment against a matrix of ones.  The workgroups of a row (column) of output tiles all stage the same dy
    // tile, so the duty rotates: workgroup t of the row takes every q_tiles-th (a_tiles-th) of its k-tiles, and the two waves that hold
    // the same fragments split the k-steps -- with the whole job in the first workgroup of each row the launch waited for that one
    // (38 -> 55 us per layer).
    const int a_tiles = (int)gridDim.x / a.q_tiles;
    const int duty_n = a.bias_mode == 1 ? a.q_tiles : (a.bias_mode == 2 ? a_tiles : 1);
    const int duty_i = a.bias_mode == 1 ? (int)blockIdx.x % a.q_tiles : (int)blockIdx.x / a.q_tiles;
    const int duty_w = a.bias_mode == 1 ? wq : wa;           // this wave's parity of k-steps
    int duty_c = 0;                                           // counts this workgroup's k-tiles modulo duty_n
    f32x16 accb[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { accb[0][e] = 0.0f; accb[1][e] = 0.0f; }
    bf16x8 ones;
#pragma unroll
    for (int c = 0; c < 8; ++c) ones[c] = (__bf16)1.0f;
    const int g4 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned lane_off = (g4 & 1) * W1_PLANE + ((8 * (g4 >> 1) + qq) * 32) + 8 * pp;
    const unsigned la = ldsb + (4 * wa) * W1_PLANE + lane_off;                    // + m * 2 planes: this wave's 64 rows = 4 planes
    const unsigned lb = ldsb + W1_HALF + (4 * wq) * W1_PLANE + lane_off;

    int kt = blockIdx.z, buf = 0;
    issue(kt, 0);
    for (; kt < a.ntk; kt += a.nsplit, buf ^= 1) {
        issue(kt + a.nsplit, buf ^ 1);           // past the end: out-of-range reads (zeros), the count per wave stays 8
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // this tile has landed, the next one may still fly
        __builtin_amdgcn_s_barrier();                               // ... for every wave
        const unsigned ba = la + buf * W1_BUF, bb = lb + buf * W1_BUF;
        v2u af[2][2][2], bfr[2][2][2];
        auto request = [&](auto kc, auto pc) {
            constexpr int ks = decltype(kc)::value, pb = decltype(pc)::value;
            sfor<0, 2>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                tr_read<m * 2 * W1_PLANE + ks * 16 * 32>(af[pb][m][0], ba);
                tr_read<m * 2 * W1_PLANE + ks * 16 * 32 + 128>(af[pb][m][1], ba);
                tr_read<m * 2 * W1_PLANE + ks * 16 * 32>(bfr[pb][m][0], bb);
                tr_read<m * 2 * W1_PLANE + ks * 16 * 32 + 128>(bfr[pb][m][1], bb);
            });
        };
        request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        sfor<0, W1_PX / 16>([&](auto kc) {
            constexpr int ks = decltype(kc)::value, pb = ks & 1;
            if constexpr (ks + 1 < W1_PX / 16) request(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, (ks + 1) & 1>{});
            constexpr int pending = ks + 1 < W1_PX / 16 ? 8 : 0;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(pending) : "memory");
            sfor<0, 2>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                depend(af[pb][m][0], af[pb][m][1]);
                depend(bfr[pb][m][0], bfr[pb][m][1]);
            });
            sfor<0, 2>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                const bf16x8 bq = frag(bfr[pb][n][0], bfr[pb][n][1]);
                sfor<0, 2>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(af[pb][m][0], af[pb][m][1]), bq, acc[m][n], 0, 0, 0);
                });
            });
            if (a.bias_mode != 0 && duty_c == duty_i && (ks & 1) == duty_w) {
                sfor<0, 2>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    if (a.bias_mode == 1) accb[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(af[pb][m][0], af[pb][m][1]), ones, accb[m], 0, 0, 0);
                    else accb[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, frag(bfr[pb][m][0], bfr[pb][m][1]), accb[m], 0, 0, 0);
                });
            }
        });
        __builtin_amdgcn_s_barrier();            // everyone is done reading this buffer before the next iteration's DMA overwrites it
        duty_c = duty_c + 1 == duty_n ? 0 : duty_c + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int ci = q0 + 64 * wq + 32 * n + j;
            if (ci >= a.CQ) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = a0 + 64 * wa + 32 * m + 8 * (e >> 2) + 4 * h + (e & 3);
                if (co < a.CA) atomicAdd(a.dw + (size_t)co * a.CQ + ci, acc[m][n][e]);
            }
        }
    if (a.bias_mode != 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (a.bias_mode == 1) {            // row sums: every column of the accumulator holds them; column 0 writes
                if (j == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int co = a0 + 64 * wa + 32 * m + 8 * (e >> 2) + 4 * h + (e & 3);
                        if (co < a.CA) atomicAdd(a.db + co, accb[m][e]);
                    }
                }
            } else {                           // column sums: every row holds them; row 0 (h = 0, e = 0) writes
                const int ci = q0 + 64 * wq + 32 * m + j;
                if (h == 0 && ci < a.CQ) atomicAdd(a.db + ci, accb[m][0]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ picture-end layers
// dW[a][q][kh][kw] = sum_{b,r,c} P[b][a][r][c] * Q[b][q][2r + kh - 2][2c + kw - 2]: the weight gradient of the two 5x5 stride-2 layers
// at the picture -- g_a_conv1 = Conv2d(3 -> 128) (MASIC.py:515: P = dy, Q = x, dW [Cout][Cin][5][5]) and g_s_conv4 =
// ConvTranspose2d(128 -> 3) (:550: P = x, Q = dy, dW [Cin][Cout][5][5]) -- five times per training step (three analysis passes, two
// synthesis passes).  A 128 x 75 x (B Hc Wc) GEMM whose contraction index is the coarse pixel: 10 GFLOP, bound by reading P.  The
// tap-packing float32-tile kernel (conv_wgrad.hip: conv_wgrad_packed_f32) reads P as float32 NCHW and gathers its fragments with
// 4-byte LDS reads: 165 us.  Here P is read where it already lives in the bf16 mode, F16K (the GDN backward's dx / the saved input
// of the last synthesis layer: 134 MB instead of 268, by DMA, fragments by transposed LDS reads as above), and the Q operand is
// built in LDS as an im2col tile: lane = coarse pixel, wave = q channel, 15 coalesced 8-byte loads of its 5 x 6 fine window, rounded
// to bf16 and stored as the two 16-"channel" records (taps 0..15, 16..24 + zeros) of that pixel -- the layout the transposed read
// expects.  Workgroup = 4 waves, k-tiles of 64 coarse pixels, wave w owns output channels 32w .. 32w+31 x 3 q x 25 taps (three
// accumulators).  The kernel is bound by memory latency, not by its 12 MFMAs per tile and wave: P (HBM) is requested two tiles
// ahead -- a ring of three DMA buffers behind a counted vmcnt --, Q (25 MB, L2 resident) one tile ahead through registers; 78 KB of
// LDS: two workgroups per CU.  Each workgroup stores one partial of the 9 600 weights; a finishing
// pass adds the partials in index order (deterministic, no atomics, nothing to zero).
constexpr int WP_PLANE = 64 * 32 + 128;              // 64 records of 32 bytes + the bank offset between planes
constexpr int WP_PBYTES = 8 * WP_PLANE, WP_QBYTES = 6 * WP_PLANE;
constexpr int WP_LDS = 3 * WP_PBYTES + 2 * WP_QBYTES;      // P ring of three | Q ring of two
constexpr int WP_NOUT = 128 * 75;

struct WgPicArgs {
    const unsigned short* p16;    // F16K [B][8][HWc][16]
    const float* q;               // float32 NCHW [B][q_ctot][2 Hc][2 Wc], channels q_coff .. q_coff + 2
    float* part;                  // [gridDim.x][128][3][25]
    int B, Hc, Wc, q_ctot, q_coff, tpi, ntk;
};

__global__ __launch_bounds__(256, 2) void wgrad_pic_f16k(const WgPicArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = a.Hc * a.Wc, Hf = 2 * a.Hc, Wf = 2 * a.Wc;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.p16, 0, a.B * 8 * HW * 32, 0x00020000);
    // P: DMA instruction i = wave + 4 j, j < 4: plane i >> 1, pixel half i & 1 (32 records of 32 bytes, 2 lanes each)
    auto issue_p = [&](int kt, int pbuf) {
        const bool real = kt < a.ntk;
        const int kk = real ? kt : 0;
        const int b = kk / a.tpi, p0 = (kk - b * a.tpi) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave + 4 * j;
            const int pl = i >> 1, half = i & 1;
            const int px = p0 + half * 32 + (lane >> 1);
            const int voff = (real && px < HW) ? px * 32 + (lane & 1) * 16 : 0x7ffffff0;
            dma16(rp, lds + pbuf * WP_PBYTES + pl * WP_PLANE + half * 1024, voff, ((b * 8 + pl) * HW) * 32);
        }
    };
    // Q: waves 0..2 = channel q, lane = coarse pixel: fine rows 2r-2 .. 2r+2, columns 2c-2 .. 2c+3 as three 8-byte loads per row.
    // Always 15 load instructions (clamped addresses, values zeroed afterwards): the counted wait below relies on it.
    auto load_q = [&](int kt, float2 (&qv)[5][3], unsigned (&qok)[5]) {
        if (wave >= 3) return;
        const bool real = kt < a.ntk;
        const int kk = real ? kt : 0;
        const int b = kk / a.tpi, px = (kk - b * a.tpi) * 64 + lane;
        const int r = px / a.Wc, c = px - r * a.Wc;
        const float* qb = a.q + ((size_t)b * a.q_ctot + a.q_coff + wave) * (size_t)Hf * Wf;
#pragma unroll
        for (int rr = 0; rr < 5; ++rr)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int fh = 2 * r + rr - 2, fw = 2 * c - 2 + 2 * j;
                const bool ok = real && px < HW && fh >= 0 && fh < Hf && fw >= 0 && fw < Wf;
                const int ch = fh < 0 ? 0 : (fh >= Hf ? Hf - 1 : fh), cw = fw < 0 ? 0 : (fw >= Wf ? Wf - 2 : fw);
                qv[rr][j] = *(const float2*)(qb + (size_t)ch * Wf + cw);
                qok[rr] = (qok[rr] & ~(1u << j)) | ((unsigned)ok << j);
            }
    };
    auto store_q = [&](int qbuf, float2 (&qv)[5][3], const unsigned (&okm)[5]) {
        if (wave >= 3) return;
        // the loaded window is first touched HERE, behind the counted wait: without this the compiler converts each value to bf16 right
        // behind its load (fewer live registers) and waits for the loads before the MFMAs they were meant to run under
#pragma unroll
        for (int rr = 0; rr < 5; ++rr)
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(qv[rr][j].x), "+v"(qv[rr][j].y));
        // taps t = 5 kh + kw -> column t of this q's 32: plane 2q holds t 0..15, plane 2q+1 t 16..31 (25..31 zero)
        unsigned short h[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            float v = 0.0f;
            if (t < 25) {
                const int kh = t / 5, kw = t - 5 * kh;
                v = (kw & 1) ? qv[kh][kw >> 1].y : qv[kh][kw >> 1].x;
                v = ((okm[kh] >> (kw >> 1)) & 1u) ? v : 0.0f;
            }
            h[t] = __builtin_bit_cast(unsigned short, (__bf16)v);
        }
        unsigned char* dst = lds + 3 * WP_PBYTES + qbuf * WP_QBYTES + (2 * wave) * WP_PLANE + lane * 32;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                v4u w;
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = (unsigned)h[pl * 16 + hf * 8 + 2 * k] | ((unsigned)h[pl * 16 + hf * 8 + 2 * k + 1] << 16);
                *(v4u*)(dst + pl * WP_PLANE + hf * 16) = w;
            }
    };
    // this wave's LDS stores are done, then the workgroup barrier -- NOT __syncthreads(), whose fence also waits for every outstanding
    // global load / DMA (vmcnt(0)): the requests of the tile after next are meant to stay in flight across it
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    f32x16 acc[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;
    const int g4 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned lane_off = (g4 & 1) * WP_PLANE + ((8 * (g4 >> 1) + qq) * 32) + 8 * pp;
    const unsigned la = ldsb + (2 * wave) * WP_PLANE + lane_off;
    const unsigned lb = ldsb + 3 * WP_PBYTES + lane_off;               // + n * 2 planes
    auto contract = [&](int pbuf, int qbuf) {
        const unsigned ba = la + pbuf * WP_PBYTES, bb = lb + qbuf * WP_QBYTES;
        sfor<0, 4>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            v2u af[2], bfr[3][2];
            tr_read<ks * 16 * 32>(af[0], ba);
            tr_read<ks * 16 * 32 + 128>(af[1], ba);
            sfor<0, 3>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                tr_read<n * 2 * WP_PLANE + ks * 16 * 32>(bfr[n][0], bb);
                tr_read<n * 2 * WP_PLANE + ks * 16 * 32 + 128>(bfr[n][1], bb);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            depend(af[0], af[1]);
            sfor<0, 3>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                depend(bfr[n][0], bfr[n][1]);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(af[0], af[1]), frag(bfr[n][0], bfr[n][1]), acc[n], 0, 0, 0);
            });
        });
    };
    // Pipeline: P two tiles ahead by DMA (ring of three; the compiler does not see these requests, the counted wait below does), Q one
    // tile ahead through registers (ordinary loads requested BEFORE the tile's DMAs, first touched in store_q: the compiler's own
    // wait covers them).  Registers that carry loads across the loop's back edge were tried for Q: the compiler's wait insertion
    // then waits for most of the current tile's requests too, and inline-asm loads get copied while still in flight.
    float2 qa[5][3];
    unsigned oka[5] = {0, 0, 0, 0, 0};
    const int st = gridDim.x;
    int kt = blockIdx.x, pb = 0, qb_ = 0;
    issue_p(kt, 0);
    issue_p(kt + st, 1);
    load_q(kt, qa, oka);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_q(0, qa, oka);
    lds_barrier();
    for (; kt < a.ntk; kt += st, pb = pb == 2 ? 0 : pb + 1, qb_ ^= 1) {
        load_q(kt + st, qa, oka);
        asm volatile("" ::: "memory");
        issue_p(kt + 2 * st, pb == 0 ? 2 : pb - 1);     // (that slot was last read in the previous iteration, which ended with a barrier)
        contract(pb, qb_);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // everything but this iteration's four DMAs: the next tile's P records and Q window
        store_q(qb_ ^ 1, qa, oka);
        lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // accumulator n: row (a) = 32 wave + 8 (e >> 2) + 4 h + (e & 3), column (tap) = lane & 31
    const int t = lane & 31, h = lane >> 5;
    float* mine = a.part + (size_t)blockIdx.x * WP_NOUT;
    if (t < 25) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ch = 32 * wave + 8 * (e >> 2) + 4 * h + (e & 3);
                mine[(ch * 3 + n) * 25 + t] = acc[n][e];
            }
    }
}

__global__ __launch_bounds__(256) void wgrad_pic_finish(const float* __restrict__ part, float* __restrict__ dw, int nparts) {
    // 64 weights per block, four slices of the partials per weight (thread = (slice, weight)), fixed order
    __shared__ float red[4][64];
    const int j = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + j;
    float s = 0.0f;
    if (o < WP_NOUT)
        for (int i0 = sl; i0 < nparts; i0 += 32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = i0 + 4 * k < nparts ? part[(size_t)(i0 + 4 * k) * WP_NOUT + o] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
    red[sl][j] = s;
    __syncthreads();
    if (sl == 0 && o < WP_NOUT) dw[o] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}

template <int MA, int NQ, int KS = 3>
void launch(const Wg3Args& a, int grid, int groups, hipStream_t st) {
    auto kfn = wgrad3x3_f16k<MA, NQ, KS>;
    static bool attr_set = false;
    constexpr size_t lds = 3 * (size_t)(2 * MA * GP + 2 * NQ * XGeom<KS>::XP) + 1024;
    static_assert(lds <= 160 * 1024, "three tile buffers must fit the LDS");
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(grid, groups), dim3(NQ * KS * 64), lds, st, a);
}

}  // namespace

extern "C" size_t masic_conv3x3_wgrad_f16k_workspace_bytes(int Cin, int Cout) { return (size_t)9 * Cin * Cout * sizeof(float); }

// dW [Cout][Cin][3][3] (float32) of Conv2d(Cin -> Cout, k3, s1, p1) from x and dy in F16K (Cin, Cout multiples of 32).
extern "C" int masic_conv3x3_wgrad_f16k(const void* x_f16k, const void* dy_f16k, float* dw, void* workspace,
                                        int B, int Cin, int Cout, int H, int W, void* stream) {
    return masic_conv3x3_wgrad_f16k_ws(x_f16k, dy_f16k, dw, workspace, B, Cin, Cout, H, W, 0, stream);
}

// workspace_clean: as masic_conv2d_wgrad_ws -- zeros on entry, zeros again on exit, no fill launch
extern "C" int masic_conv3x3_wgrad_f16k_ws(const void* x_f16k, const void* dy_f16k, float* dw, void* workspace,
                                           int B, int Cin, int Cout, int H, int W, int workspace_clean, void* stream) {
    MASIC_REQUIRE(x_f16k && dy_f16k && dw && workspace, MASIC_ERR_ARG, "conv3x3_wgrad_f16k: null pointer");
    MASIC_REQUIRE(B > 0 && H > 0 && W > 0 && Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, MASIC_ERR_UNSUPPORTED,
                  "conv3x3_wgrad_f16k: needs Cin, Cout multiples of 32");
    MASIC_REQUIRE((long)B * (Cin > Cout ? Cin : Cout) * H * W * 2 < (1l << 31), MASIC_ERR_UNSUPPORTED, "conv3x3_wgrad_f16k: tensor too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    if (!workspace_clean && masic_zero_async(workspace, masic_conv3x3_wgrad_f16k_workspace_bytes(Cin, Cout), st, 2) != hipSuccess) {
        masic_set_error("conv3x3_wgrad_f16k: workspace memset failed");
        return MASIC_ERR_LAUNCH;
    }
    Wg3Args a{(const unsigned short*)dy_f16k, (const unsigned short*)x_f16k, (float*)workspace, B, H, W, Cout / 16, Cin / 16, Cout, Cin, 0, 0, 96, 64,
              ceil_div(W, TC), ceil_div(H, TR), 0};
    a.ntiles = a.tiles_w * a.tiles_h * B;
    // channel groups of up to 96 output x 64 input channels (6 waves: at most two per SIMD, so the 9 accumulator tiles of a wave
    // fit its register budget); each group: one workgroup per CU taking a strided share of the pixel tiles.  Equally shaped groups
    // -- (full, full), (remainder, full), (full, remainder), (remainder, remainder) -- go out as ONE launch each (blockIdx.y).
    const int na = Cout / 96, ra = (Cout % 96) / 32, nqf = Cin / 64, rq = (Cin % 64) / 32;
    auto go = [&](int a0, int q0, int ma, int nq, int ga, int gq) {
        if (ga <= 0 || gq <= 0 || ma <= 0 || nq <= 0) return;
        a.a0 = a0; a.q0 = q0; a.gq = gq;
        // workgroups per channel group: 256 where the picture has that many tiles (Independent_EN); at latent resolution (64 tiles,
        // up to 16 groups: the hyper transforms) one round of the CUs over ALL groups -- every workgroup ends with a full group tile
        // of float atomics (55 296 of them), and 1024 single-tile workgroups spent 190 us of which ~150 were those atomics
        int grid = a.ntiles < 256 ? a.ntiles : 256;
        if (a.ntiles < 256 && grid * ga * gq > 256) grid = 256 / (ga * gq) > 0 ? 256 / (ga * gq) : 1;
        switch (ma * 4 + nq) {
            case 1 * 4 + 1: launch<1, 1>(a, grid, ga * gq, st); break;
            case 1 * 4 + 2: launch<1, 2>(a, grid, ga * gq, st); break;
            case 2 * 4 + 1: launch<2, 1>(a, grid, ga * gq, st); break;
            case 2 * 4 + 2: launch<2, 2>(a, grid, ga * gq, st); break;
            case 3 * 4 + 1: launch<3, 1>(a, grid, ga * gq, st); break;
            default: launch<3, 2>(a, grid, ga * gq, st); break;
        }
    };
    go(0, 0, 3, 2, na, nqf);
    go(96 * na, 0, ra, 2, 1, nqf);
    go(0, 64 * nqf, 3, rq, na, 1);
    go(96 * na, 64 * nqf, ra, rq, 1, 1);
    const int AQ = Cin * Cout;
    int tb = (9 * AQ + 255) / 256;
    if (tb > 2048) tb = 2048;
    hipLaunchKernelGGL(wgrad3_transpose_kernel, dim3(tb), dim3(256), 0, st, (float*)workspace, dw, AQ, 9, workspace_clean);
    return masic_launch_status("conv3x3_wgrad_f16k");
}

extern "C" size_t masic_conv5x5_wgrad_f16k_workspace_bytes(int Cin, int Cout) { return (size_t)25 * Cin * Cout * sizeof(float); }

// dW [Cout][Cin][5][5] (float32) of Conv2d(Cin -> Cout, k5, s1, p2) from x and dy in F16K (Cin, Cout multiples of 32): the 5x5
// stride-1 layers at latent resolution -- encode_hyper's first layer (MASIC.py:222) and the context model (MaskedConv2d, :627; its
// masked taps get their dense gradient like everywhere else: the reference masks weight.data, not the gradient).  The 3x3 kernel
// with five kernel-row waves per 32 input channels: workgroup = 64 output x 32 input channels x 25 taps (5 waves, 10 accumulator
// tiles each).  workspace: masic_conv5x5_wgrad_f16k_workspace_bytes; workspace_clean as masic_conv3x3_wgrad_f16k_ws.
extern "C" int masic_conv5x5_wgrad_f16k_ws(const void* x_f16k, const void* dy_f16k, float* dw, void* workspace,
                                           int B, int Cin, int Cout, int H, int W, int workspace_clean, void* stream) {
    MASIC_REQUIRE(x_f16k && dy_f16k && dw && workspace, MASIC_ERR_ARG, "conv5x5_wgrad_f16k: null pointer");
    MASIC_REQUIRE(B > 0 && H > 0 && W > 0 && Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, MASIC_ERR_UNSUPPORTED,
                  "conv5x5_wgrad_f16k: needs Cin, Cout multiples of 32");
    MASIC_REQUIRE((long)B * (Cin > Cout ? Cin : Cout) * H * W * 2 < (1l << 31), MASIC_ERR_UNSUPPORTED, "conv5x5_wgrad_f16k: tensor too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    if (!workspace_clean && masic_zero_async(workspace, masic_conv5x5_wgrad_f16k_workspace_bytes(Cin, Cout), st, 3) != hipSuccess) {
        masic_set_error("conv5x5_wgrad_f16k: workspace memset failed");
        return MASIC_ERR_LAUNCH;
    }
    Wg3Args a{(const unsigned short*)dy_f16k, (const unsigned short*)x_f16k, (float*)workspace, B, H, W, Cout / 16, Cin / 16, Cout, Cin, 0, 0, 64, 32,
              ceil_div(W, TC), ceil_div(H, TR), 0};
    a.ntiles = a.tiles_w * a.tiles_h * B;
    const int na = Cout / 64, ra = (Cout % 64) / 32, nq = Cin / 32;
    auto go = [&](int a0, int ma, int ga) {
        if (ga <= 0 || ma <= 0) return;
        a.a0 = a0; a.q0 = 0; a.gq = nq;
        int grid = a.ntiles < 256 ? a.ntiles : 256;          // (one round of the CUs over all groups at latent resolution, as the 3x3 layers)
        if (a.ntiles < 256 && grid * ga * nq > 256) grid = 256 / (ga * nq) > 0 ? 256 / (ga * nq) : 1;
        if (ma == 2) launch<2, 1, 5>(a, grid, ga * nq, st);
        else launch<1, 1, 5>(a, grid, ga * nq, st);
    };
    go(0, 2, na);
    go(64 * na, ra, 1);
    const int AQ = Cin * Cout;
    int tb = (25 * AQ + 255) / 256;
    if (tb > 2048) tb = 2048;
    hipLaunchKernelGGL(wgrad3_transpose_kernel, dim3(tb), dim3(256), 0, st, (float*)workspace, dw, AQ, 25, workspace_clean);
    return masic_launch_status("conv5x5_wgrad_f16k");
}

constexpr int WP_MAX_WGS = 512;
extern "C" size_t masic_pic_wgrad_f16k_workspace_bytes() { return (size_t)WP_MAX_WGS * WP_NOUT * sizeof(float); }

// dW (float32, 128 x 3 x 5 x 5) of the picture-end 5x5 stride-2 layers in the bf16 mode: p_f16k = the 128-channel tensor at the
// coarse resolution (Hc x Wc) in F16K -- dy of Conv2d(3 -> 128) (dW = [Cout][Cin][5][5]) or x of ConvTranspose2d(128 -> 3) (dW =
// [Cin][Cout][5][5]) --, q = the 3-channel tensor at 2 Hc x 2 Wc, float32 NCHW (channels q_coff .. q_coff + 2 of q_ctot).
// workspace: masic_pic_wgrad_f16k_workspace_bytes() bytes, contents irrelevant on entry and exit.
extern "C" int masic_pic_wgrad_f16k(const void* p_f16k, const float* q, float* dw, void* workspace, int B, int Hc, int Wc,
                                    int q_ctot, int q_coff, void* stream) {
    MASIC_REQUIRE(p_f16k && q && dw && workspace, MASIC_ERR_ARG, "pic_wgrad_f16k: null pointer");
    MASIC_REQUIRE(B > 0 && Hc > 0 && Wc > 0 && q_coff >= 0 && q_coff + 3 <= q_ctot, MASIC_ERR_SHAPE, "pic_wgrad_f16k: bad shape");
    MASIC_REQUIRE((long)B * 128 * Hc * Wc * 2 < (1l << 31), MASIC_ERR_UNSUPPORTED, "pic_wgrad_f16k: tensor too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    WgPicArgs a{(const unsigned short*)p_f16k, q, (float*)workspace, B, Hc, Wc, q_ctot, q_coff, ceil_div(Hc * Wc, 64), 0};
    a.ntk = a.B * a.tpi;
    const int grid = a.ntk < WP_MAX_WGS ? a.ntk : WP_MAX_WGS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_pic_f16k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_pic_f16k, dim3(grid), dim3(256), WP_LDS, st, a);
    hipLaunchKernelGGL(wgrad_pic_finish, dim3(ceil_div(WP_NOUT, 64)), dim3(256), 0, st, (const float*)workspace, dw, grid);
    return masic_launch_status("pic_wgrad_f16k");
}

// dw [CA][CQ] (float32) = sum over batch and pixels of rows[b][a][p] * cols[b][q][p], both operands F16K ([B][C/16][HW][16] bf16),
// CA, CQ multiples of 16.  For a Conv2d(Cin -> Cout, k1) weight [Cout][Cin]: rows = dy, cols = x; for the ConvTranspose2d(k1) form
// [Cin][Cout] (reference MASIC.py:338-376): rows = x, cols = dy.
extern "C" int masic_gemm_wgrad_f16k(const void* rows_f16k, const void* cols_f16k, float* dw, int B, int CA, int CQ, int HW, void* stream) {
    return masic_gemm_wgrad_bias_f16k(rows_f16k, cols_f16k, dw, 0, B, CA, CQ, HW, stream);
}

// bias_of: 0 none; 1: dw[CA*CQ .. CA*CQ + CA) = sums of the rows operand over batch and pixels; 2: dw[CA*CQ .. + CQ) = sums of the
// columns operand -- the bias gradient of the layer (dy = rows of a Conv2d, columns of a ConvTranspose2d), in the same launch
extern "C" int masic_gemm_wgrad_bias_f16k(const void* rows_f16k, const void* cols_f16k, float* dw, int bias_of, int B, int CA, int CQ, int HW,
                                          void* stream) {
    MASIC_REQUIRE(rows_f16k && cols_f16k && dw, MASIC_ERR_ARG, "gemm_wgrad_f16k: null pointer");
    MASIC_REQUIRE(bias_of >= 0 && bias_of <= 2, MASIC_ERR_ARG, "gemm_wgrad_f16k: bias_of %d", bias_of);
    MASIC_REQUIRE(B > 0 && HW > 0 && CA > 0 && CQ > 0 && CA % 16 == 0 && CQ % 16 == 0, MASIC_ERR_UNSUPPORTED, "gemm_wgrad_f16k: channel counts must be multiples of 16");
    MASIC_REQUIRE((long)B * (CA > CQ ? CA : CQ) * HW * 2 < (1l << 31), MASIC_ERR_UNSUPPORTED, "gemm_wgrad_f16k: tensor too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = bias_of == 1 ? CA : (bias_of == 2 ? CQ : 0);
    if (masic_zero_async(dw, ((size_t)CA * CQ + nb) * sizeof(float), st, 4) != hipSuccess) {
        masic_set_error("gemm_wgrad_f16k: zero fill failed");
        return MASIC_ERR_LAUNCH;
    }
    Wg1Args a{(const unsigned short*)rows_f16k, (const unsigned short*)cols_f16k, dw, B, HW, CA / 16, CQ / 16, CA, CQ, ceil_div(CQ, 128), ceil_div(HW, W1_PX), 0, 0,
               dw + (size_t)CA * CQ, bias_of};
    a.ntk = a.B * a.tpi;
    const int base = ceil_div(CA, 128) * a.q_tiles;
    // pixel splits: as many as keep the grid within ONE round of the 256 CUs -- every split adds a full output tile of float atomics
    // (64 KiB per workgroup, ~13 us at the chip's atomic rate), which outweighs a second co-resident workgroup: 55 us per layer at
    // 540 workgroups, 40 at 216 (1152 x 768 x 8192; MASIC_WGRAD1_WGS overrides the 256 for A/B timing)
    static const int target = getenv("MASIC_WGRAD1_WGS") ? atoi(getenv("MASIC_WGRAD1_WGS")) : 256;
    int nsplit = target / base;
    if (nsplit > a.ntk) nsplit = a.ntk;
    a.nsplit = nsplit < 1 ? 1 : nsplit;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad1x1_f16k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad1x1_f16k, dim3(base, 1, a.nsplit), dim3(256), 2 * W1_BUF, st, a);
    return masic_launch_status("gemm_wgrad_f16k");
}
