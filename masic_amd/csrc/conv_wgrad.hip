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
    if (blockIdx.z < a.ntiles) issue(blockIdx.z, lds);
    for (int tile = blockIdx.z; tile < a.ntiles; tile += a.nsplit, ++it) {
        float* cur = lds + (it & 1) * buf_sz;
        __syncthreads();                                   // this tile landed; everyone left the other buffer
        if (tile + a.nsplit < a.ntiles) issue(tile + a.nsplit, lds + ((it + 1) & 1) * buf_sz);
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

    int it = 0;
    if (blockIdx.z < a.ntiles) issue(blockIdx.z, lds);
    for (int tile = blockIdx.z; tile < a.ntiles; tile += a.nsplit, ++it) {
        float* cur = lds + (it & 1) * buf_sz;
        __syncthreads();
        if (tile + a.nsplit < a.ntiles) issue(tile + a.nsplit, lds + ((it + 1) & 1) * buf_sz);
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

// ws [T][CA][CQ] -> dw [CA][CQ][T]
__global__ __launch_bounds__(256) void wgrad_transpose_kernel(const float* __restrict__ ws, float* __restrict__ dw, int T, int AQ) {
    const size_t total = (size_t)T * AQ;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t aq = i / T;
        const int t = (int)(i - aq * T);
        dw[i] = ws[(size_t)t * AQ + aq];
    }
}

}  // namespace

extern "C" size_t masic_conv2d_wgrad_workspace_bytes(const masic_conv_desc_t* d) {
    if (!d) return 0;
    return (size_t)d->Cin * d->Cout * d->KH * d->KW * sizeof(float);
}

extern "C" int masic_conv2d_wgrad(const float* x, const float* dy, float* dw, void* workspace,
                                  const masic_conv_desc_t* d, void* stream) {
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
    if (hipMemsetAsync(workspace, 0, wbytes, st) != hipSuccess) {
        masic_set_error("conv2d_wgrad: workspace memset failed");
        return MASIC_ERR_LAUNCH;
    }
    const int a_tiles = ceil_div(a.CA, 64);
    const int ncol_tiles = ceil_div(Tt * a.CQ, 32);
    if (a.CQ <= 8 && Tt > 1 && ncol_tiles <= 8) {          // few fine-side channels: (tap, q) pairs packed into the 32 MFMA columns
        a.q_tiles = 1;
        int nsplit = ceil_div(512, a_tiles);
        if (nsplit > a.ntiles) nsplit = a.ntiles;
        a.nsplit = nsplit < 1 ? 1 : nsplit;
        const size_t lds = (size_t)2 * (64 * PS + a.CQ * a.QS) * sizeof(float);
        dim3 grid(a_tiles, 1, a.nsplit);
        if (bf16) {
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
        if (bf16) {
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
    hipLaunchKernelGGL(wgrad_transpose_kernel, dim3(tb), dim3(256), 0, st, (const float*)workspace, dw, T, AQ);
    return masic_launch_status("conv2d_wgrad");
}
