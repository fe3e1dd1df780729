// conv_wgrad.hip -- weight gradients of Conv2d / ConvTranspose2d on the matrix cores (gfx950).
//
// What torch autograd computes for every conv of the reference graph (convolution_backward, 39 % of the
// reference's CPU step time).  Both layer kinds reduce to one form over a coarse tensor P and a
// fine tensor Q:
//     dW[a][q][kh][kw] = sum_{b,r,c} P[b][a][r][c] * Q[b][q][r*s + kh - p][c*s + kw - p]
//   Conv2d:          P = dy (a = co), Q = x  (q = ci)  -> dW laid out [Cout][Cin][kh][kw]
//   ConvTranspose2d: P = x  (a = ci), Q = dy (q = co)  -> dW laid out [Cin][Cout][kh][kw]
// i.e. a GEMM  dW_t[a][q] = P[a][pixel] * Q_t[pixel][q]  per tap t whose K dimension is the pixel index.
//
// A workgroup (8 waves) owns a 64(a) x 32(q) tile of dW for ALL taps (1x1 layers: 64 x 128) and a strided share of the
// pixel tiles (2 coarse rows x 32 columns): wave = (which 32 a-channels, tap group), 7 accumulators of 32x32 per wave
// for a 5x5 kernel.  Per pixel tile the P rows and the whole fine patch of Q (s+KH rows) are staged ONCE into LDS by
// global->LDS DMA -- double buffered, the next tile lands while this one is contracted -- with an odd channel pitch so
// that the 32 lanes of an MFMA operand (32 different channels) hit 32 different banks; each wave issues
// v_mfma_f32_32x32x2_f32 with k = a pixel pair.  Partial tiles of all workgroups are reduced with float atomics into a
// [tap][a][q] workspace (q contiguous: 128-byte segments per half wave = full atomic rate) and a final pass transposes
// it into the weight layout.
#include <type_traits>

#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) float g_zero_wg[4] = {0.0f, 0.0f, 0.0f, 0.0f};

__device__ __forceinline__ void dma4(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

struct WgradArgs {
    const float* P; const float* Q; float* ws;
    int B, CA, CQ;
    int p_ctot, p_coff, q_ctot, q_coff;
    int Hc, Wc, Hf, Wf;
    int s, pad, KH, KW;
    int tiles_w, tiles_h, ntiles, nsplit;
    int q_tiles;
    int PWq, QROWS, QPIX, QG, QS;   // fine patch: columns, rows, pixels, DMA groups per channel, channel pitch (odd)
};

constexpr int PS = 65;    // P-tile channel pitch (64 pixels + 1)

// 8 waves: wave = (wa: which 32 of the 64 a-channels, u: tap group [QT == 1] or 32-wide q sub-tile [QT == 4])
//   T = KH*KW taps; QT q sub-tiles of 32 per block; each wave accumulates TPW taps of one 32a x 32q tile
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive pixels of one channel from an LDS tile (stride `st` floats between pixels) as a bf16 MFMA fragment
__device__ __forceinline__ wbf16x8 frag8(const float* p, int st) {
    wbf16x8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (__bf16)p[i * st];
    return f;
}

// BF16: operands rounded to bf16 while they are read from the (float32) LDS tiles, v_mfma_f32_32x32x16_bf16 with
// k = 16 consecutive pixels of a row (lane half h takes pixels 8h .. 8h+7), float32 accumulate -- the training step of the
// bf16-operand mode; the float32 form (k = a pixel pair, exact products) is the parity path.
template <int QT, int TPW, bool BF16>
__global__ __launch_bounds__(512) void conv_wgrad_f32(const WgradArgs a) {

    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int buf_sz = 64 * PS + 32 * QT * a.QS;     // [P 64][PS] | [Q 32*QT][QS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, u = wave >> 1;
    const int tg = QT == 1 ? u : 0, qsub = QT == 1 ? 0 : u;
    const int j = lane & 31, h = lane >> 5;
    const int a0 = (blockIdx.x / a.q_tiles) * 64, q0 = (blockIdx.x % a.q_tiles) * (32 * QT);
    const int T = a.KH * a.KW;
    const int t_lo = tg * TPW, t_hi = (t_lo + TPW < T) ? t_lo + TPW : T;     // this wave's taps [t_lo, t_hi)

    f32x16 acc[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;
    int toff[TPW];                                    // patch offset of tap k (row kh, column kw)
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int t = t_lo + k < T ? t_lo + k : T - 1;
        toff[k] = (t / a.KW) * a.PWq + (t % a.KW);
    }

    const size_t cplane = (size_t)a.Hc * a.Wc, fplane = (size_t)a.Hf * a.Wf;
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int prr = lane >> 5, pcc = lane & 31;

    // Staging.  Float32 form: global -> LDS DMA of the next tile into the other buffer.  BF16 form: the DMA's 4 bytes per lane
    // make one wave-instruction per 256 bytes and that issue rate (~80 cycles each, measured: 320 of them per tile) bounded the
    // kernel; the tile is fetched into registers with ordinary loads while the current one is contracted and stored to the other
    // buffer afterwards (same LDS image).
    constexpr int NQW = 32 * QT / 8;                  // q channels per wave
    constexpr int MAXG = QT == 1 ? 8 : 1;             // 64-pixel groups per channel patch the register form holds
    float rp[8], rq[BF16 ? MAXG * NQW : 1];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int r0 = (trem / a.tiles_w) * 2, c0 = (trem % a.tiles_w) * 32;
        {
            const int r = r0 + prr, c = c0 + pcc;
            const bool pok = r < a.Hc && c < a.Wc;
            const float* pb = a.P + ((size_t)b * a.p_ctot + a.p_coff) * cplane + (size_t)(pok ? r : 0) * a.Wc + (pok ? c : 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ch = wave + 8 * i;
                rp[i] = (pok && (a0 + ch) < a.CA) ? pb[(size_t)(a0 + ch) * cplane] : 0.0f;
            }
        }
        const float* qb = a.Q + ((size_t)b * a.q_ctot + a.q_coff) * fplane;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int e = g * 64 + lane;
            const int rr = e / a.PWq, pc = e - rr * a.PWq;
            const int fh = r0 * a.s - a.pad + rr, fw = c0 * a.s - a.pad + pc;
            const bool qok = g < a.QG && e < a.QPIX && fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf;
            const float* src = qb + (size_t)(qok ? fh : 0) * a.Wf + (qok ? fw : 0);
#pragma unroll
            for (int i = 0; i < NQW; ++i) {
                const int ch = wave + 8 * i;
                rq[BF16 ? g * NQW + i : 0] = (qok && (q0 + ch) < a.CQ) ? src[(size_t)(q0 + ch) * fplane] : 0.0f;
            }
        }
    };
    auto stash = [&](float* buf) {
        float* Pt = buf;
        float* Qt = buf + 64 * PS;
#pragma unroll
        for (int i = 0; i < 8; ++i) Pt[(wave + 8 * i) * PS + lane] = rp[i];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            if (g < a.QG && g * 64 + lane < a.QPIX) {
#pragma unroll
                for (int i = 0; i < NQW; ++i) Qt[(wave + 8 * i) * a.QS + g * 64 + lane] = rq[BF16 ? g * NQW + i : 0];
            }
        }
    };
    auto issue = [&](int tile, float* buf) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int r0 = (trem / a.tiles_w) * 2, c0 = (trem % a.tiles_w) * 32;
        float* Pt = buf;
        float* Qt = buf + 64 * PS;
        {   // P: 64 channels x (2 rows x 32 cols), 8 channels per wave
            const int r = r0 + prr, c = c0 + pcc;
            const bool pok = r < a.Hc && c < a.Wc;
            const float* pb = a.P + ((size_t)b * a.p_ctot + a.p_coff) * cplane + (size_t)(pok ? r : 0) * a.Wc + (pok ? c : 0);
            for (int ch = wave; ch < 64; ch += 8) {
                const bool ok = pok && (a0 + ch) < a.CA;
                dma4(ok ? pb + (size_t)(a0 + ch) * cplane : g_zero_wg, Pt + ch * PS);
            }
        }
        {   // Q: 32*QT channels x (QROWS fine rows x PWq cols)
            const float* qb = a.Q + ((size_t)b * a.q_ctot + a.q_coff) * fplane;
            for (int g = 0; g < a.QG; ++g) {
                const int e = g * 64 + lane;
                const int rr = e / a.PWq, pc = e - rr * a.PWq;
                const int fh = r0 * a.s - a.pad + rr, fw = c0 * a.s - a.pad + pc;
                const bool inpatch = e < a.QPIX;
                const bool qok = inpatch && fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf;
                const float* src = qb + (size_t)(qok ? fh : 0) * a.Wf + (qok ? fw : 0);
                if (inpatch) {                                   // lanes beyond the patch must not spill into the next channel
                    for (int ch = wave; ch < 32 * QT; ch += 8) {
                        const bool ok = qok && (q0 + ch) < a.CQ;
                        dma4(ok ? src + (size_t)(q0 + ch) * fplane : g_zero_wg, Qt + ch * a.QS + g * 64);
                    }
                }
            }
        }
    };

    int it = 0;
    if constexpr (BF16) {
        if (blockIdx.z < a.ntiles) fetch(blockIdx.z);
    } else {
        if (blockIdx.z < a.ntiles) issue(blockIdx.z, lds);
    }
    for (int tile = blockIdx.z; tile < a.ntiles; tile += a.nsplit, ++it) {
        float* cur = lds + (it & 1) * buf_sz;
        if constexpr (BF16) {
            stash(cur);                                        // the other buffer may still be read by a slower wave: untouched
            __syncthreads();
            if (tile + a.nsplit < a.ntiles) fetch(tile + a.nsplit);
        } else {
            __syncthreads();                                   // this tile landed; everyone left the other buffer
            if (tile + a.nsplit < a.ntiles) issue(tile + a.nsplit, lds + ((it + 1) & 1) * buf_sz);
        }
        if constexpr (BF16) {
            const float* pa = cur + (wa * 32 + j) * PS;
            const float* qa = cur + 64 * PS + (qsub * 32 + j) * a.QS;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int px = 16 * ks + 8 * h;             // this lane's 8 pixels: one row, columns cc .. cc+7
                const int rr = px >> 5, cc = px & 31;
                const wbf16x8 af = frag8(pa + px, 1);
                const float* qrow = qa + (rr * a.s) * a.PWq + cc * a.s;
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (t_lo + k < t_hi) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, frag8(qrow + toff[k], a.s), acc[k], 0, 0, 0);
            }
            continue;
        }
        const float* pa = cur + (wa * 32 + j) * PS + h;
        const float* qa = cur + 64 * PS + (qsub * 32 + j) * a.QS + h * a.s;
        for (int kk = 0; kk < 32; ++kk) {
            const int px = 2 * kk;                      // pixel of lane half 0; half 1 takes px+1 (same row: 32 is even)
            const int rr = px >> 5, cc = px & 31;
            const float av = pa[px];
            const float* qrow = qa + (rr * a.s) * a.PWq + cc * a.s;
            float bv[TPW];
#pragma unroll
            for (int k = 0; k < TPW; ++k) bv[k] = qrow[toff[k]];
#pragma unroll
            for (int k = 0; k < TPW; ++k)
                if (t_lo + k < t_hi) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[k], acc[k], 0, 0, 0);
        }
    }

    // ---- reduce into the [tap][a][q] workspace (q contiguous: 128-byte segments per half wave)
    const int q = q0 + qsub * 32 + j;
    if (q < a.CQ) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            if (t_lo + k >= t_hi) continue;
            float* wsk = a.ws + (size_t)(t_lo + k) * a.CA * a.CQ;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ai = a0 + wa * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (ai < a.CA) atomicAdd(wsk + (size_t)ai * a.CQ + q, acc[k][e]);
            }
        }
    }
}

// ---- packed variant for layers whose fine-side tensor has few channels (CQ <= 8: the 3-channel image ends of the
// transforms, the 6<->3 pre/after convs).  A 32-wide q tile would be >= 75 % padding there; instead the 32 lanes of the
// B operand enumerate (tap, q) pairs -- column n = t*CQ + q, T*CQ columns in all -- so a 5x5 kernel over 3 channels is
// 75 columns = 3 MFMA column tiles instead of 25.  Wave = (which 32 a-channels, column-tile group); everything else
// (pixel tiles, DMA double buffering, odd channel pitch, atomic reduction into [tap][a][q]) as above.
template <int TPW, bool BF16>
__global__ __launch_bounds__(512) void conv_wgrad_packed_f32(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int buf_sz = 64 * PS + a.CQ * a.QS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, u = wave >> 1;
    const int j = lane & 31, h = lane >> 5;
    const int a0 = blockIdx.x * 64;
    const int T = a.KH * a.KW, NCOL = T * a.CQ;

    f32x16 acc[TPW];
    int coloff[TPW];       // LDS offset (within the Q tile) of this lane's column: q*QS + kh*PWq + kw; -1 beyond NCOL
    int colidx[TPW];       // workspace index t*CA*CQ + q
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;
        const int n = (u + 4 * k) * 32 + j;
        const int t = n / a.CQ, q = n - t * a.CQ;
        coloff[k] = n < NCOL ? q * a.QS + (t / a.KW) * a.PWq + (t % a.KW) : 0;
        colidx[k] = n < NCOL ? t * a.CA * a.CQ + q : -1;
    }

    const size_t cplane = (size_t)a.Hc * a.Wc, fplane = (size_t)a.Hf * a.Wf;
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    const int prr = lane >> 5, pcc = lane & 31;

    auto issue = [&](int tile, float* buf) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int r0 = (trem / a.tiles_w) * 2, c0 = (trem % a.tiles_w) * 32;
        float* Pt = buf;
        float* Qt = buf + 64 * PS;
        {
            const int r = r0 + prr, c = c0 + pcc;
            const bool pok = r < a.Hc && c < a.Wc;
            const float* pb = a.P + ((size_t)b * a.p_ctot + a.p_coff) * cplane + (size_t)(pok ? r : 0) * a.Wc + (pok ? c : 0);
            for (int ch = wave; ch < 64; ch += 8) {
                const bool ok = pok && (a0 + ch) < a.CA;
                dma4(ok ? pb + (size_t)(a0 + ch) * cplane : g_zero_wg, Pt + ch * PS);
            }
        }
        {
            const float* qb = a.Q + ((size_t)b * a.q_ctot + a.q_coff) * fplane;
            // (channel, group) pairs dealt to the 8 waves
            for (int cg = wave; cg < a.CQ * a.QG; cg += 8) {
                const int ch = cg / a.QG, g = cg - ch * a.QG;
                const int e = g * 64 + lane;
                const int rr = e / a.PWq, pc = e - rr * a.PWq;
                const int fh = r0 * a.s - a.pad + rr, fw = c0 * a.s - a.pad + pc;
                const bool inpatch = e < a.QPIX;
                const bool qok = inpatch && fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf;
                const float* src = qb + (size_t)ch * fplane + (size_t)(qok ? fh : 0) * a.Wf + (qok ? fw : 0);
                if (inpatch) dma4(qok ? src : g_zero_wg, Qt + ch * a.QS + g * 64);
            }
        }
    };

    // register-staged form of the same tile (BF16; see conv_wgrad_f32): 8 P channels and up to 8 (channel, group) pairs per wave
    float rp[8], rq[8];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int r0 = (trem / a.tiles_w) * 2, c0 = (trem % a.tiles_w) * 32;
        const int r = r0 + prr, c = c0 + pcc;
        const bool pok = r < a.Hc && c < a.Wc;
        const float* pb = a.P + ((size_t)b * a.p_ctot + a.p_coff) * cplane + (size_t)(pok ? r : 0) * a.Wc + (pok ? c : 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = wave + 8 * i;
            rp[i] = (pok && (a0 + ch) < a.CA) ? pb[(size_t)(a0 + ch) * cplane] : 0.0f;
        }
        const float* qb = a.Q + ((size_t)b * a.q_ctot + a.q_coff) * fplane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cg = wave + 8 * i;
            const int ch = cg / a.QG, g = cg - ch * a.QG;
            const int e = g * 64 + lane;
            const int rr = e / a.PWq, pc = e - rr * a.PWq;
            const int fh = r0 * a.s - a.pad + rr, fw = c0 * a.s - a.pad + pc;
            const bool qok = cg < a.CQ * a.QG && e < a.QPIX && fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf;
            rq[i] = qok ? qb[(size_t)ch * fplane + (size_t)fh * a.Wf + fw] : 0.0f;
        }
    };
    auto stash = [&](float* buf) {
        float* Qt = buf + 64 * PS;
#pragma unroll
        for (int i = 0; i < 8; ++i) buf[(wave + 8 * i) * PS + lane] = rp[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cg = wave + 8 * i;
            const int ch = cg / a.QG, g = cg - ch * a.QG;
            if (cg < a.CQ * a.QG && g * 64 + lane < a.QPIX) Qt[ch * a.QS + g * 64 + lane] = rq[i];
        }
    };

    int it = 0;
    if constexpr (BF16) {
        if (blockIdx.z < a.ntiles) fetch(blockIdx.z);
    } else {
        if (blockIdx.z < a.ntiles) issue(blockIdx.z, lds);
    }
    for (int tile = blockIdx.z; tile < a.ntiles; tile += a.nsplit, ++it) {
        float* cur = lds + (it & 1) * buf_sz;
        if constexpr (BF16) {
            stash(cur);
            __syncthreads();
            if (tile + a.nsplit < a.ntiles) fetch(tile + a.nsplit);
        } else {
            __syncthreads();
            if (tile + a.nsplit < a.ntiles) issue(tile + a.nsplit, lds + ((it + 1) & 1) * buf_sz);
        }
        if constexpr (BF16) {
            const float* pa = cur + (wa * 32 + j) * PS;
            const float* qa = cur + 64 * PS;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int px = 16 * ks + 8 * h;
                const int rr = px >> 5, cc = px & 31;
                const wbf16x8 af = frag8(pa + px, 1);
                const float* qrow = qa + (rr * a.s) * a.PWq + cc * a.s;
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if ((u + 4 * k) * 32 < NCOL) {
                        wbf16x8 bf = frag8(qrow + coloff[k], a.s);
                        if (colidx[k] < 0) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) bf[i] = (__bf16)0.0f;
                        }
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[k], 0, 0, 0);
                    }
            }
            continue;
        }
        const float* pa = cur + (wa * 32 + j) * PS + h;
        const float* qa = cur + 64 * PS + h * a.s;
        for (int kk = 0; kk < 32; ++kk) {
            const int px = 2 * kk;
            const int rr = px >> 5, cc = px & 31;
            const float av = pa[px];
            const float* qrow = qa + (rr * a.s) * a.PWq + cc * a.s;
            float bv[TPW];
#pragma unroll
            for (int k = 0; k < TPW; ++k) bv[k] = colidx[k] >= 0 ? qrow[coloff[k]] : 0.0f;
#pragma unroll
            for (int k = 0; k < TPW; ++k)
                if ((u + 4 * k) * 32 < NCOL) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[k], acc[k], 0, 0, 0);
        }
    }
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        if (colidx[k] < 0) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ai = a0 + wa * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (ai < a.CA) atomicAdd(a.ws + (size_t)colidx[k] + (size_t)ai * a.CQ, acc[k][e]);
        }
    }
}

// ---- 1x1 layers in the bf16-operand mode (the nine-layer entropy-parameter stacks: 18 weight gradients per step):
// a plain GEMM dW[a][q] = sum_px P[a][px] Q[q][px] with k = pixel.  Workgroup (4 waves) = a 128 x 128 block of dW over a
// strided share of the 64-pixel K tiles; wave = 64 x 64 (2 x 2 accumulators).  A thread fetches 16-byte runs (4 pixels of a
// channel row) of the next K tile into registers while the current one is contracted, rounds them to bf16 and lays them
// out [channel][64 px] in LDS with a 144-byte pitch (a ds_read_b128 of 16 channels x 8 pixels is conflict-free).
// v_mfma_f32_32x32x16_bf16, float32 accumulate; the partial blocks are added into dW (zeroed first) with float atomics.
struct Wgrad1x1Args {
    const float* P; const float* Q; float* dw;
    int CA, CQ, p_ctot, p_coff, q_ctot, q_coff, HW, ntk, nsplit, q_tiles;
};

constexpr int W1_PITCH = 144;

__global__ __launch_bounds__(256) void conv_wgrad_1x1_bf16(const Wgrad1x1Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * W1_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = w & 1, wq = w >> 1, j = lane & 31, h = lane >> 5;
    const int a0 = (blockIdx.x / a.q_tiles) * 128, q0 = (blockIdx.x % a.q_tiles) * 128;
    const int lr = tid >> 4, lc = tid & 15;                 // this thread's rows lr + 16 i, pixels 4 lc .. 4 lc + 3
    const int tpi = a.HW >> 6;                              // K tiles per image

    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][0][e] = 0.0f; acc[0][1][e] = 0.0f; acc[1][0][e] = 0.0f; acc[1][1][e] = 0.0f; }

    float4 ra[8], rb[8];
    auto fetch = [&](int kt) {
        const int b = kt / tpi, px = (kt - b * tpi) * 64 + 4 * lc;
        const float* pb = a.P + ((size_t)b * a.p_ctot + a.p_coff) * a.HW + px;
        const float* qb = a.Q + ((size_t)b * a.q_ctot + a.q_coff) * a.HW + px;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ca = a0 + lr + 16 * i, cq = q0 + lr + 16 * i;
            ra[i] = ca < a.CA ? *reinterpret_cast<const float4*>(pb + (size_t)ca * a.HW) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = cq < a.CQ ? *reinterpret_cast<const float4*>(qb + (size_t)cq * a.HW) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bf16x4_t va, vb;
            va[0] = (__bf16)ra[i].x; va[1] = (__bf16)ra[i].y; va[2] = (__bf16)ra[i].z; va[3] = (__bf16)ra[i].w;
            vb[0] = (__bf16)rb[i].x; vb[1] = (__bf16)rb[i].y; vb[2] = (__bf16)rb[i].z; vb[3] = (__bf16)rb[i].w;
            *reinterpret_cast<bf16x4_t*>(lds + (lr + 16 * i) * W1_PITCH + lc * 8) = va;
            *reinterpret_cast<bf16x4_t*>(lds + 128 * W1_PITCH + (lr + 16 * i) * W1_PITCH + lc * 8) = vb;
        }
    };

    int kt = blockIdx.z;
    if (kt < a.ntk) fetch(kt);
    const unsigned char* pa = lds + (64 * wa + j) * W1_PITCH + 16 * h;
    const unsigned char* pq = lds + 128 * W1_PITCH + (64 * wq + j) * W1_PITCH + 16 * h;
    for (; kt < a.ntk; kt += a.nsplit) {
        __syncthreads();                                   // the previous tile's fragment reads are done
        stash();
        __syncthreads();
        if (kt + a.nsplit < a.ntk) fetch(kt + a.nsplit);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wbf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = __builtin_bit_cast(wbf16x8, *reinterpret_cast<const uint4*>(pa + i * 32 * W1_PITCH + 32 * ks));
                fb[i] = __builtin_bit_cast(wbf16x8, *reinterpret_cast<const uint4*>(pq + i * 32 * W1_PITCH + 32 * ks));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int q = q0 + 64 * wq + 32 * ni + j;
            if (q >= a.CQ) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ai = a0 + 64 * wa + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (ai < a.CA) atomicAdd(a.dw + (size_t)ai * a.CQ + q, acc[mi][ni][e]);
            }
        }
}

// ---- 5x5 stride-2 layers in the bf16-operand mode (the analysis / synthesis / hyper transforms: the bulk of the weight-gradient
// time).  Same decomposition as conv_wgrad_f32 -- block = 64 a x 32 q x all 25 taps over a strided share of the 2 x 32 coarse
// pixel tiles, k = pixel -- but the tiles live in LDS as bf16, laid out so that every MFMA operand is ONE aligned ds_read_b128:
//   P  [64 a][64 px], 144-byte pitch.
//   Q  [32 q][7 fine rows][even | odd column plane][48], 1360-byte channel pitch.  Fine column f = 2c + kw - 2: the even taps
//      (kw = 0, 2, 4) read the even plane at c - 1, c, c + 1, the odd taps (kw = 1, 3) the odd plane at c - 1, c.  Planes are
//      stored 7 entries to the right, which puts the run of the middle tap on a 16-byte boundary and the 2-pixel pairs a thread
//      writes on 4-byte ones; the runs shifted by one pixel are funnel-shifted (v_alignbit) out of that read and ONE neighbouring
//      dword -- 5 operand fragments from 2 ds_read_b128 + 3 ds_read_b32, against 40 strided ds_read_b32 in the float32-tile form.
// 8 waves: waves 0-4 = kernel row kh with the three even taps (both 32-channel a blocks: 6 accumulators), waves 5-7 = the two
// odd taps of rows {0,1}, {2,3}, {4} (8, 8, 4 accumulators); a SIMD's two waves carry 10-14 of the 50 accumulators.
// The next tile is fetched with 16-byte loads into registers while this one is contracted, rounded to bf16 on the way into LDS
// (single buffer, two barriers per tile).  Workgroups that share pixel tiles (the 8 (a, q) blocks of a 128 x 128 layer) get
// consecutive ids on ONE XCD, so P and Q cross HBM once.
constexpr int W5_PP = 144, W5_QP = 1360, W5_ROW = 192, W5_PLANE = 96;
constexpr int W5_LDS = 64 * W5_PP + 32 * W5_QP;

// IN16: P and Q are bf16 NCHW tensors (masic_conv2d_wgrad_bf16in: the GDN backward writes dx that way, the saved F16K activation is
// converted that way) -- 8-byte loads of 4 pixels instead of 16-byte ones, no conversion on the way into LDS, half the L2 -> CU stream
// that bounds this kernel
template <bool IN16>
__global__ __launch_bounds__(512) void conv_wgrad_k5s2_bf16(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[W5_LDS];
    unsigned char* const pl = lds;
    unsigned char* const ql = lds + 64 * W5_PP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
    const int j = lane & 31, h = lane >> 5;
    // block -> (split, combo): the combos of one split sit next to each other on one XCD
    const int ncombo = ((a.CA + 63) / 64) * a.q_tiles;
    const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int combo = m % ncombo, split = (m / ncombo) * 8 + xcd;
    const int a0 = (combo / a.q_tiles) * 64, q0 = (combo % a.q_tiles) * 32;

    const size_t cplane = (size_t)a.Hc * a.Wc, fplane = (size_t)a.Hf * a.Wf;
    const int tiles_per_img = a.tiles_w * a.tiles_h;

    // staging roles (tile-invariant, so a tile costs a handful of address instructions per thread):
    //   P: thread = (channel tid>>4 [+32], coarse row, 4-pixel run);  Q: thread = (fine row, 4-column run) of channels grp + 4k
    const int p_ch = tid >> 4, p_rr = (tid >> 3) & 1, p_run = tid & 7;
    const int q_grp = tid / 126, q_rem = tid - q_grp * 126, q_rr = q_rem / 18, q_i = q_rem - q_rr * 18;
    const bool q_thread = tid < 504;
    unsigned char* const p_dst = pl + p_ch * W5_PP + p_rr * 64 + p_run * 8;
    unsigned char* const q_dst = ql + q_grp * W5_QP + q_rr * W5_ROW + (2 * q_i + 6) * 2;       // plane entry e + 7, e = 2i - 1
    typedef typename std::conditional<IN16, uint2, float4>::type ld_t;      // 4 consecutive pixels of one channel row
    typedef typename std::conditional<IN16, unsigned short, float>::type el_t;
    const el_t* const Pp = reinterpret_cast<const el_t*>(a.P);
    const el_t* const Qp = reinterpret_cast<const el_t*>(a.Q);
    ld_t rp[2], rq[8];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int r0 = (trem / a.tiles_w) * 2, c0 = (trem % a.tiles_w) * 32;
        {
            const int r = r0 + p_rr, c = c0 + 4 * p_run;
            const bool ok = r < a.Hc && c < a.Wc;
            const el_t* pb = Pp + ((size_t)b * a.p_ctot + a.p_coff + a0 + p_ch) * cplane + (ok ? (size_t)r * a.Wc + c : 0);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                rp[k] = (ok && a0 + p_ch + 32 * k < a.CA) ? *reinterpret_cast<const ld_t*>(pb + (size_t)(32 * k) * cplane) : ld_t{};
        }
        {
            const int fh = 2 * r0 - 2 + q_rr, fw = 2 * c0 - 4 + 4 * q_i;
            const bool ok = q_thread && fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf;
            const el_t* qb = Qp + ((size_t)b * a.q_ctot + a.q_coff + q0 + q_grp) * fplane + (ok ? (size_t)fh * a.Wf + fw : 0);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                rq[k] = (ok && q0 + q_grp + 4 * k < a.CQ) ? *reinterpret_cast<const ld_t*>(qb + (size_t)(4 * k) * fplane) : ld_t{};
        }
    };
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    auto stash = [&]() {
        if constexpr (IN16) {
#pragma unroll
            for (int k = 0; k < 2; ++k) *reinterpret_cast<uint2*>(p_dst + 32 * k * W5_PP) = rp[k];
            if (q_thread) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {                              // (x, y) = pixels (0, 1), (2, 3): even plane 0, 2; odd plane 1, 3
                    *reinterpret_cast<unsigned*>(q_dst + 4 * k * W5_QP) = (rq[k].x & 0xffffu) | (rq[k].y << 16);
                    *reinterpret_cast<unsigned*>(q_dst + 4 * k * W5_QP + W5_PLANE) = (rq[k].x >> 16) | (rq[k].y & 0xffff0000u);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bf16x4_t v;
                v[0] = (__bf16)rp[k].x; v[1] = (__bf16)rp[k].y; v[2] = (__bf16)rp[k].z; v[3] = (__bf16)rp[k].w;
                *reinterpret_cast<bf16x4_t*>(p_dst + 32 * k * W5_PP) = v;
            }
            if (q_thread) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    bf16x2_t ev, od;                                           // columns 2c0-4+4i .. +3 = even, odd, even, odd
                    ev[0] = (__bf16)rq[k].x; ev[1] = (__bf16)rq[k].z;
                    od[0] = (__bf16)rq[k].y; od[1] = (__bf16)rq[k].w;
                    *reinterpret_cast<bf16x2_t*>(q_dst + 4 * k * W5_QP) = ev;
                    *reinterpret_cast<bf16x2_t*>(q_dst + 4 * k * W5_QP + W5_PLANE) = od;
                }
            }
        }
    };

    const unsigned char* pa = pl + j * W5_PP + 16 * h;
    const int q = q0 + j;
    // one instantiation per wave role (its own accumulator set): EVEN = taps kw 0, 2, 4 of row kh_first; else taps 1, 3 of NROWS rows
    auto role = [&](auto even_c, auto nrows_c, int kh_first) {
        constexpr bool EVEN = decltype(even_c)::value;
        constexpr int NROWS = decltype(nrows_c)::value, NT = EVEN ? 3 : 2, NACC = NROWS * NT * 2;
        f32x16 acc[NACC];                                  // [row kk][tap t][a block i]
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;
        const unsigned char* qa = ql + j * W5_QP + (EVEN ? 0 : W5_PLANE);
        int tile = split;
        if (tile < a.ntiles) fetch(tile);
        for (; tile < a.ntiles; tile += a.nsplit) {
            __syncthreads();                               // the previous tile's fragment reads are done
            stash();
            __syncthreads();
            if (tile + a.nsplit < a.ntiles) fetch(tile + a.nsplit);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int rho = ks >> 1, g = 2 * (ks & 1) + h;   // coarse row of the tile, 8-pixel column group
                const wbf16x8 fa0 = __builtin_bit_cast(wbf16x8, *reinterpret_cast<const uint4*>(pa + 32 * ks));
                const wbf16x8 fa1 = __builtin_bit_cast(wbf16x8, *reinterpret_cast<const uint4*>(pa + 32 * W5_PP + 32 * ks));
#pragma unroll
                for (int kk = 0; kk < NROWS; ++kk) {
                    const unsigned char* qr = qa + (2 * rho + kh_first + kk) * W5_ROW + 16 * (g + 1);
                    const uint4 cur = *reinterpret_cast<const uint4*>(qr);
                    const unsigned prev = *reinterpret_cast<const unsigned*>(qr - 4);
                    const unsigned a1 = __builtin_amdgcn_alignbit(cur.y, cur.x, 16), a2 = __builtin_amdgcn_alignbit(cur.z, cur.y, 16),
                                   a3 = __builtin_amdgcn_alignbit(cur.w, cur.z, 16);
                    const wbf16x8 fm = __builtin_bit_cast(wbf16x8, make_uint4(__builtin_amdgcn_alignbit(cur.x, prev, 16), a1, a2, a3));   // column - 1
                    const wbf16x8 f0 = __builtin_bit_cast(wbf16x8, cur);
                    f32x16* ac = acc + kk * NT * 2;
                    ac[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fm, ac[0], 0, 0, 0);
                    ac[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fm, ac[1], 0, 0, 0);
                    ac[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, f0, ac[2], 0, 0, 0);
                    ac[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, f0, ac[3], 0, 0, 0);
                    if constexpr (EVEN) {
                        const unsigned next = *reinterpret_cast<const unsigned*>(qr + 16);
                        const wbf16x8 fp = __builtin_bit_cast(wbf16x8, make_uint4(a1, a2, a3, __builtin_amdgcn_alignbit(next, cur.w, 16)));     // column + 1
                        ac[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fp, ac[4], 0, 0, 0);
                        ac[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fp, ac[5], 0, 0, 0);
                    }
                }
            }
        }
        if (q < a.CQ) {
#pragma unroll
            for (int kk = 0; kk < NROWS; ++kk)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int kw = EVEN ? 2 * t : 2 * t + 1;
                    float* wsk = a.ws + (size_t)((kh_first + kk) * 5 + kw) * a.CA * a.CQ;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int ai = a0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                            if (ai < a.CA) atomicAdd(wsk + (size_t)ai * a.CQ + q, acc[(kk * NT + t) * 2 + i][e]);
                        }
                }
        }
    };
    typedef std::integral_constant<bool, true> true_c;
    typedef std::integral_constant<bool, false> false_c;
    if (u < 5) role(true_c{}, std::integral_constant<int, 1>{}, u);
    else if (u < 7) role(false_c{}, std::integral_constant<int, 2>{}, 2 * (u - 5));
    else role(false_c{}, std::integral_constant<int, 1>{}, 4);
}

// ---------------------------------------------------------------------------- 5x5 stride-1 layers with a handful of channels
// encoder2.pre_conv (Conv2d 6 -> 3) and decoder2.after_conv (ConvTranspose2d 6 -> 3; MASIC.py:559, :576) on full-resolution pictures:
// 450 weights, 2 M pixels per batch of 8 x 512 x 512 -- 0.94 G multiply-adds and a few planes of traffic.  On the matrix-core
// kernels above 3 (or 6) of the 64 tile rows are real and the tile staging dominates (211 / 260 us).  Here: plain float32 FMAs.
// A workgroup takes NA a-channels x all CQ q-channels and walks 32 x 64 pixel tiles; lane = column, wave = 8 rows; the q tile (2-row
// halo, columns c0-4 .. c0+67 so that every 16-byte DMA chunk is aligned and wholly inside or outside the picture) sits in LDS, a
// thread slides a 5 x 5 register window down its 8 rows and feeds NA x CQ x 25 accumulators.  Each workgroup stores ONE partial per
// weight and a finishing pass adds them in a fixed order (deterministic; float atomics from 500 workgroups on the same 15 cache
// lines cost 56 us of a 220 us launch).  rnd: operands rounded to bf16 as they are staged (the bf16-operand mode: the same operand
// values as the matrix-core kernels of that mode); products and sums are float32 either way.  Needs W % 4 == 0.
__device__ __attribute__((aligned(16))) float g_zero16_wg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
__device__ __forceinline__ float wg_round(float v, int rnd) { return rnd ? (float)(__bf16)v : v; }
__device__ __forceinline__ void dma16w(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int NA, int CQ>
__global__ __launch_bounds__(256, 2) void conv_wgrad_small_s1(const WgradArgs a, float* __restrict__ part, int rnd) {
    constexpr int TR = 32, TC = 64, LR = TR + 4, LC = TC + 8, R = 8, NV = NA * CQ * 25;
    constexpr int NQ = CQ * LR * LC, NC16 = NQ / 4, NCH = (NC16 + 63) / 64;       // 16-byte chunks, wave instructions
    __shared__ __attribute__((aligned(16))) float ql[NCH * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a_base = blockIdx.y * NA;
    const size_t plane = (size_t)a.Hc * a.Wc;
    const int tiles_w = (a.Wc + TC - 1) / TC, tiles_h = (a.Hc + TR - 1) / TR;
    const int tiles_per_img = tiles_w * tiles_h, ntiles = tiles_per_img * a.B;
    float acc[NA][CQ][25];
#pragma unroll
    for (int n = 0; n < NA; ++n)
#pragma unroll
        for (int q = 0; q < CQ; ++q)
#pragma unroll
            for (int t = 0; t < 25; ++t) acc[n][q][t] = 0.0f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, trem = tile - b * tiles_per_img;
        const int r0 = (trem / tiles_w) * TR, c0 = (trem % tiles_w) * TC;
        __syncthreads();                                  // the previous tile has been read
        // the q tile by global -> LDS DMA, 16 bytes per lane: every request of the tile is in flight before the first one is awaited
        const float* qb = a.Q + ((size_t)b * a.q_ctot + a.q_coff) * plane;
        for (int ch = wave; ch < NCH; ch += 4) {
            const int e = ch * 64 + lane;                 // chunk index: (q, row, 18 chunks of 4 columns)
            const int q = e / (LR * (LC / 4)), rem = e - q * (LR * (LC / 4));
            const int rr = rem / (LC / 4), cc = rem - rr * (LC / 4);
            const int fh = r0 - 2 + rr, fw = c0 - 4 + 4 * cc;
            const bool ok = e < NC16 && fh >= 0 && fh < a.Hc && fw >= 0 && fw < a.Wc;
            dma16w(ok ? qb + (size_t)q * plane + (size_t)fh * a.Wc + fw : g_zero16_wg, ql + ch * 256);
        }
        float p[NA][R];
        {
            const int c = c0 + lane;
            const float* pb = a.P + ((size_t)b * a.p_ctot + a.p_coff + a_base) * plane;
#pragma unroll
            for (int n = 0; n < NA; ++n)
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int r = r0 + wave * R + i;
                    p[n][i] = (r < a.Hc && c < a.Wc && a_base + n < a.CA) ? wg_round(pb[(size_t)n * plane + (size_t)r * a.Wc + c], rnd) : 0.0f;
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (rnd) {                                        // bf16-operand mode: the staged tile rounded in place
            for (int e = tid; e < NQ; e += 256) ql[e] = (float)(__bf16)ql[e];
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const float* base = ql + q * (LR * LC) + (wave * R) * LC + lane + 2;
            float win[5][5];                              // row (i + kh) of the halo tile lives in win[(i + kh) % 5]
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) win[kh][kw] = base[kh * LC + kw];
#pragma unroll
            for (int i = 0; i < R; ++i) {
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) win[(i + 4) % 5][kw] = base[(i + 4) * LC + kw];
#pragma unroll
                for (int kh = 0; kh < 5; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw)
#pragma unroll
                        for (int n = 0; n < NA; ++n) acc[n][q][kh * 5 + kw] = fmaf(p[n][i], win[(i + kh) % 5][kw], acc[n][q][kh * 5 + kw]);
            }
        }
    }
    __syncthreads();
    static_assert(4 * NV <= NCH * 256, "reduction scratch fits the tile");
    float* red = ql;                                      // [4][NV]
#pragma unroll
    for (int n = 0; n < NA; ++n)
#pragma unroll
        for (int q = 0; q < CQ; ++q)
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                float v = acc[n][q][t];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) red[wave * NV + (n * CQ + q) * 25 + t] = v;
            }
    __syncthreads();
    float* mine = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * NV;
    for (int k = tid; k < NV; k += 256) mine[k] = (red[k] + red[NV + k]) + (red[2 * NV + k] + red[3 * NV + k]);
}

// dw[o] = sum over the gx workgroups of a column of partials, in index order; the partials are zeroed behind the read (the
// persistent workspace protocol of masic_conv2d_wgrad_ws); one wave per weight
__global__ __launch_bounds__(64) void wgrad_small_finish(float* __restrict__ part, float* __restrict__ dw, int gx, int gy, int NV, int total) {
    const int o = blockIdx.x, lane = threadIdx.x;
    const int y = o / NV, k = o - y * NV;
    float s = 0.0f;
    for (int x0 = 0; x0 < gx; x0 += 256) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + 64 * j + lane;
            float* src = part + ((size_t)(x < gx ? x : 0) * gy + y) * NV + k;
            v[j] = x < gx ? *src : 0.0f;
            if (x < gx) *src = 0.0f;
        }
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0 && o < total) dw[o] = s;
}

// ws [T][CA][CQ] -> dw [CA][CQ][T]
// rezero: every workspace element is read by exactly one thread of this pass, which then puts the zero back -- a persistent workspace
// is clean again for the next weight gradient and needs no fill launch of its own (masic_conv2d_wgrad_ws)
__global__ __launch_bounds__(256) void wgrad_transpose_kernel(float* __restrict__ ws, float* __restrict__ dw, int T, int AQ, int rezero) {
    const size_t total = (size_t)T * AQ;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t aq = i / T;
        const int t = (int)(i - aq * T);
        dw[i] = ws[(size_t)t * AQ + aq];
        if (rezero) ws[(size_t)t * AQ + aq] = 0.0f;
    }
}

}  // namespace

namespace {
constexpr int SMALL_WGS = 512;       // workgroups of conv_wgrad_small_s1 (two per CU), each with one partial of 150 floats in the workspace
// the 5x5 stride-1 layers with 3 / 6 channels on either side (pre_conv, after_conv) take conv_wgrad_small_s1
bool small_path(const masic_conv_desc_t* d) {
    static const bool on = [] { const char* e = getenv("MASIC_WGRAD_SMALL"); return e == nullptr || e[0] != '0'; }();
    if (!on || d->stride != 1 || d->KH != 5 || d->KW != 5 || d->pad != 2 || d->Wi % 4 != 0 || d->Hi != d->Ho || d->Wi != d->Wo) return false;
    const int CA = d->transposed ? d->Cin : d->Cout, CQ = d->transposed ? d->Cout : d->Cin;
    return (CQ == 6 || CQ == 3) && CA <= 8;
}
}  // namespace

extern "C" size_t masic_conv2d_wgrad_workspace_bytes(const masic_conv_desc_t* d) {
    if (!d) return 0;
    const size_t w = (size_t)d->Cin * d->Cout * d->KH * d->KW * sizeof(float);
    const size_t small = small_path(d) ? (size_t)SMALL_WGS * 150 * sizeof(float) : 0;
    return w > small ? w : small;
}

extern "C" int masic_conv2d_wgrad(const float* x, const float* dy, float* dw, void* workspace,
                                  const masic_conv_desc_t* d, void* stream) {
    return masic_conv2d_wgrad_ws(x, dy, dw, workspace, d, 0, stream);
}

// workspace_clean != 0: `workspace` holds zeros on entry (a persistent buffer: zeroed once by its owner) and holds zeros again when the
// call's last kernel has run -- no fill launch per weight gradient (a training step computes ~80 of them; a launch costs the device
// ~10 us whatever it does).  0: contents unknown, zeroed here, left dirty (the behaviour of masic_conv2d_wgrad).
namespace { int wgrad_launch(const void* x, const void* dy, float* dw, void* workspace, const masic_conv_desc_t* d, int workspace_clean, int in16, void* stream); }
extern "C" int masic_conv2d_wgrad_ws(const float* x, const float* dy, float* dw, void* workspace,
                                     const masic_conv_desc_t* d, int workspace_clean, void* stream) {
    return wgrad_launch(x, dy, dw, workspace, d, workspace_clean, 0, stream);
}

// 1 if masic_conv2d_wgrad_bf16in takes this layer: the 5x5 stride-2 layers of the analysis / synthesis / hyper transforms in bf16 mode
extern "C" int masic_conv2d_wgrad_bf16in_supported(const masic_conv_desc_t* d) {
    if (d == nullptr || d->prec != MASIC_PREC_BF16 || d->KH != 5 || d->KW != 5 || d->stride != 2 || d->pad != 2) return 0;
    const int CQ = d->transposed ? d->Cout : d->Cin, Hf = d->transposed ? d->Ho : d->Hi, Wf = d->transposed ? d->Wo : d->Wi;
    const int Hc = d->transposed ? d->Hi : d->Ho, Wc = d->transposed ? d->Wi : d->Wo;
    return CQ > 8 && Wf % 8 == 0 && Wf == 2 * Wc && Hf == 2 * Hc;
}

// x, dy: bf16 NCHW tensors (x: [B][in_ctot][Hi][Wi] with the channel view of d, dy: [B][Cout][Ho][Wo]); only layers for which
// masic_conv2d_wgrad_bf16in_supported(d); workspace as masic_conv2d_wgrad_ws.
extern "C" int masic_conv2d_wgrad_bf16in(const void* x_bf16, const void* dy_bf16, float* dw, void* workspace,
                                         const masic_conv_desc_t* d, int workspace_clean, void* stream) {
    MASIC_REQUIRE(masic_conv2d_wgrad_bf16in_supported(d), MASIC_ERR_UNSUPPORTED, "conv2d_wgrad_bf16in: not a 5x5 stride-2 layer of the bf16 mode");
    return wgrad_launch(x_bf16, dy_bf16, dw, workspace, d, workspace_clean, 1, stream);
}

namespace {
int wgrad_launch(const void* xv, const void* dyv, float* dw, void* workspace, const masic_conv_desc_t* d, int workspace_clean, int in16, void* stream) {
    const float* x = (const float*)xv;
    const float* dy = (const float*)dyv;
    MASIC_REQUIRE(x && dy && dw && workspace && d, MASIC_ERR_ARG, "conv2d_wgrad: null pointer");
    MASIC_REQUIRE(d->KH * d->KW == 1 || d->KH * d->KW <= 28, MASIC_ERR_UNSUPPORTED, "conv2d_wgrad: kernel %dx%d", d->KH, d->KW);
    MASIC_REQUIRE(d->stride == 1 || d->stride == 2, MASIC_ERR_UNSUPPORTED, "conv2d_wgrad: stride %d", d->stride);
    MASIC_REQUIRE(d->in_coff >= 0 && d->in_coff + d->Cin <= d->in_ctot, MASIC_ERR_SHAPE, "conv2d_wgrad: input view out of range");
    hipStream_t st = (hipStream_t)stream;
    const bool bf16 = d->prec == MASIC_PREC_BF16;      // operands rounded to bf16, float32 accumulate (training in the bf16-operand mode)
    WgradArgs a{};
    a.ws = (float*)workspace;
    a.B = d->B; a.s = d->stride; a.pad = d->pad; a.KH = d->KH;
    if (!d->transposed) {   // P = dy (coarse, contiguous), Q = x (fine, view)
        a.P = dy; a.p_ctot = d->Cout; a.p_coff = 0; a.CA = d->Cout; a.Hc = d->Ho; a.Wc = d->Wo;
        a.Q = x; a.q_ctot = d->in_ctot; a.q_coff = d->in_coff; a.CQ = d->Cin; a.Hf = d->Hi; a.Wf = d->Wi;
    } else {                // P = x (coarse, view), Q = dy (fine, contiguous)
        a.P = x; a.p_ctot = d->in_ctot; a.p_coff = d->in_coff; a.CA = d->Cin; a.Hc = d->Hi; a.Wc = d->Wi;
        a.Q = dy; a.q_ctot = d->Cout; a.q_coff = 0; a.CQ = d->Cout; a.Hf = d->Ho; a.Wf = d->Wo;
    }
    a.KW = d->KW;
    a.tiles_w = ceil_div(a.Wc, 32); a.tiles_h = ceil_div(a.Hc, 2);
    a.ntiles = a.tiles_w * a.tiles_h * a.B;
    const int Tt = d->KH * d->KW;
    a.PWq = 31 * a.s + d->KW;
    a.QROWS = a.s + d->KH;
    a.QPIX = a.QROWS * a.PWq;
    a.QG = ceil_div(a.QPIX, 64);
    a.QS = a.QPIX | 1;
    const size_t wbytes = masic_conv2d_wgrad_workspace_bytes(d);
    if (bf16 && Tt == 1 && a.s == 1 && a.pad == 0 && (a.Hc * a.Wc) % 64 == 0) {
        Wgrad1x1Args g{};
        g.P = a.P; g.Q = a.Q; g.dw = dw; g.CA = a.CA; g.CQ = a.CQ;
        g.p_ctot = a.p_ctot; g.p_coff = a.p_coff; g.q_ctot = a.q_ctot; g.q_coff = a.q_coff;
        g.HW = a.Hc * a.Wc; g.ntk = a.B * (g.HW / 64);
        g.q_tiles = ceil_div(a.CQ, 128);
        const int base = ceil_div(a.CA, 128) * g.q_tiles;
        int nsplit = ceil_div(512, base);
        if (nsplit > g.ntk) nsplit = g.ntk;
        g.nsplit = nsplit < 1 ? 1 : nsplit;
        if (masic_zero_async(dw, wbytes, st, 0) != hipSuccess) {
            masic_set_error("conv2d_wgrad: memset failed");
            return MASIC_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(conv_wgrad_1x1_bf16, dim3(base, 1, g.nsplit), dim3(256), 0, st, g);
        return masic_launch_status("conv2d_wgrad");
    }
    if (!workspace_clean && masic_zero_async(workspace, wbytes, st, 1) != hipSuccess) {
        masic_set_error("conv2d_wgrad: workspace memset failed");
        return MASIC_ERR_LAUNCH;
    }
    const int a_tiles = ceil_div(a.CA, 64);
    const int ncol_tiles = ceil_div(Tt * a.CQ, 32);
    if (bf16 && d->KH == 5 && d->KW == 5 && a.s == 2 && a.pad == 2 && a.CQ > 8 && a.Wf % 8 == 0 && a.Wf == 2 * a.Wc && a.Hf == 2 * a.Hc) {
        a.q_tiles = ceil_div(a.CQ, 32);
        const int ncombo = a_tiles * a.q_tiles;
        // pixel splits: a multiple of 8 (dealt round-robin to the XCDs); one workgroup per CU at a time, each ends with
        // 51 200 float atomics (~4 tile times), so: rounds over the 256 CUs x (tiles per workgroup + 4), minimised
        int nsplit = 8;
        long best = -1;
        for (int ns = 8; ns <= 128; ns += 8) {
            if (ns > round_up(a.ntiles, 8)) break;
            const long cost = (long)ceil_div(ncombo * ns, 256) * (ceil_div(a.ntiles, ns) + 4);
            if (best < 0 || cost < best) { best = cost; nsplit = ns; }
        }
        a.nsplit = nsplit;
        if (in16) hipLaunchKernelGGL(conv_wgrad_k5s2_bf16<true>, dim3(ncombo * nsplit), dim3(512), 0, st, a);
        else hipLaunchKernelGGL(conv_wgrad_k5s2_bf16<false>, dim3(ncombo * nsplit), dim3(512), 0, st, a);
        const size_t total5 = (size_t)Tt * a.CA * a.CQ;
        int tb5 = (int)((total5 + 255) / 256);
        if (tb5 > 4096) tb5 = 4096;
        hipLaunchKernelGGL(wgrad_transpose_kernel, dim3(tb5), dim3(256), 0, st, (float*)workspace, dw, Tt, a.CA * a.CQ, workspace_clean);
        return masic_launch_status("conv2d_wgrad");
    }
    MASIC_REQUIRE(!in16, MASIC_ERR_UNSUPPORTED, "conv2d_wgrad_bf16in: layer shape without a bf16-input kernel");
    if (small_path(d)) {
        // pre_conv / after_conv: plain float32 FMAs, one partial per workgroup in the workspace, summed (and zeroed again) by the finishing pass
        const int ntl = ceil_div(a.Wc, 64) * ceil_div(a.Hc, 32) * a.B;
        const int NA = a.CQ == 6 ? 1 : 2, gy = ceil_div(a.CA, NA), NV = NA * a.CQ * 25;
        int gx = SMALL_WGS / gy;
        if (gx > ntl) gx = ntl;
        if (a.CQ == 6) hipLaunchKernelGGL((conv_wgrad_small_s1<1, 6>), dim3(gx, gy), dim3(256), 0, st, a, (float*)workspace, (int)bf16);
        else hipLaunchKernelGGL((conv_wgrad_small_s1<2, 3>), dim3(gx, gy), dim3(256), 0, st, a, (float*)workspace, (int)bf16);
        hipLaunchKernelGGL(wgrad_small_finish, dim3(gy * NV), dim3(64), 0, st, (float*)workspace, dw, gx, gy, NV, a.CA * a.CQ * 25);
        return masic_launch_status("conv2d_wgrad");
    }
    if (a.CQ <= 8 && Tt > 1 && ncol_tiles <= 8) {          // few fine-side channels: (tap, q) pairs packed into the 32 MFMA columns
        a.q_tiles = 1;
        int nsplit = ceil_div(512, a_tiles);
        if (nsplit > a.ntiles) nsplit = a.ntiles;
        a.nsplit = nsplit < 1 ? 1 : nsplit;
        const size_t lds = (size_t)2 * (64 * PS + a.CQ * a.QS) * sizeof(float);
        dim3 grid(a_tiles, 1, a.nsplit);
        if (bf16 && a.CQ * a.QG <= 64) {                    // (the register-staged form holds 8 (channel, group) pairs per wave)
            if (ncol_tiles <= 4) hipLaunchKernelGGL((conv_wgrad_packed_f32<1, true>), grid, dim3(512), lds, st, a);
            else hipLaunchKernelGGL((conv_wgrad_packed_f32<2, true>), grid, dim3(512), lds, st, a);
        } else if (ncol_tiles <= 4) hipLaunchKernelGGL((conv_wgrad_packed_f32<1, false>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((conv_wgrad_packed_f32<2, false>), grid, dim3(512), lds, st, a);
    } else {
        const int QT = Tt == 1 ? 4 : 1;                     // 1x1: four q sub-tiles per block; else four tap groups
        a.q_tiles = ceil_div(a.CQ, 32 * QT);
        const int base = a_tiles * a.q_tiles;
        int nsplit = ceil_div(512, base);
        if (nsplit > a.ntiles) nsplit = a.ntiles;
        a.nsplit = nsplit < 1 ? 1 : nsplit;
        const size_t lds = (size_t)2 * (64 * PS + 32 * QT * a.QS) * sizeof(float);
        MASIC_REQUIRE(lds <= 160 * 1024, MASIC_ERR_UNSUPPORTED, "conv2d_wgrad: tile does not fit LDS");
        dim3 grid(a_tiles * a.q_tiles, 1, a.nsplit);
        if (bf16 && a.QG <= 8) {                            // (the register-staged form holds 8 pixel groups per channel patch)
            if (Tt == 1) hipLaunchKernelGGL((conv_wgrad_f32<4, 1, true>), grid, dim3(512), lds, st, a);
            else if (Tt <= 12) hipLaunchKernelGGL((conv_wgrad_f32<1, 3, true>), grid, dim3(512), lds, st, a);
            else hipLaunchKernelGGL((conv_wgrad_f32<1, 7, true>), grid, dim3(512), lds, st, a);
        } else if (Tt == 1) hipLaunchKernelGGL((conv_wgrad_f32<4, 1, false>), grid, dim3(512), lds, st, a);
        else if (Tt <= 12) hipLaunchKernelGGL((conv_wgrad_f32<1, 3, false>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((conv_wgrad_f32<1, 7, false>), grid, dim3(512), lds, st, a);
    }
    const int T = d->KH * d->KW, AQ = a.CA * a.CQ;
    const size_t total = (size_t)T * AQ;
    int tb = (int)((total + 255) / 256);
    if (tb > 4096) tb = 4096;
    hipLaunchKernelGGL(wgrad_transpose_kernel, dim3(tb), dim3(256), 0, st, (float*)workspace, dw, T, AQ, workspace_clean);
    return masic_launch_status("conv2d_wgrad");
}
}  // namespace
