// conv.hip -- convolutions of the MASIC hot path for gfx950 (MI355X).
//
// Replaces every nn.Conv2d / nn.ConvTranspose2d / MaskedConv2d the reference issues from
// HSIC.forward (coremasic/mywork/MASIC.py:744-851; factories compressai/models/utils.py:128-146,
// compressai/layers/layers.py:52-83).
//
// One formulation covers all of them: an output "phase plane" pixel (r,c) accumulates, over live
// taps (a,b) and input channels ci,
//     In[ci, r*is + dh0 + a*dsh, c*is + dw0 + b*dsw] * W[co, ci, kh0 + a*khs, kw0 + b*kws]
// and lands at output (r*os + oph, c*os + opw).  A strided Conv2d is one phase (is = stride);
// a ConvTranspose2d with stride s is s*s phases with is = 1, os = s (no multiplies by the zeros of
// the zero-insertion form); a type-'A' masked conv is one phase that keeps only its 12 live taps.
//
// Kernel 1 (implicit GEMM on the matrix cores): the GEMM is  Out[co, pixel] = W[co, k] * Im2col[k, pixel]
// with k = (tap, ci).  NCHW rows are read coalesced into an LDS patch [ci][row][col] (halo
// included, padding zero-filled) once per KC input channels and re-used by every tap; the weights
// arrive pre-packed as [tap][ci][co] so their LDS image is a straight copy.  Each of the 4 waves
// owns a 32(co) x WN*32(pixel) block of the 64 x BN tile and issues v_mfma_f32_32x32x2_f32
// (float32 operands, float32 accumulate: bitwise a k-ordered fmaf chain).  The bf16-operand
// variant of the same structure is selected with desc.prec.
//
// Kernel 2 (direct): Cout <= 8 layers (128->3 synthesis output, 6->3 pre/after convs, the
// mask2weights chain) are HBM/L2-bound, so they skip the matrix cores: one thread per output pixel,
// all output channels in registers, weights through scalar loads.
#include "common.h"
#include "conv_geom.h"

namespace {
struct ConvCfg {
    int direct;         // 1: direct kernel (Cout <= 8); 2: LDS-tiled 5x5 stride-2 transposed conv to <= 4 channels
    int wvm;            // waves along the channel dimension (2, or 1 for narrow layers); 4/wvm along pixels
    int wm;             // 32-channel sub-tiles per wave: block covers BM = 32*wm*wvm output channels
    int wn;             // pixel sub-tiles (of 32) per wave: 4 -> BM x 256 block tile, 1 -> BM x 64
    int buf_sz;         // floats per LDS buffer: patch [KC][PSZ] + weights [round4(taps*KC)][64]
    int vec4;           // 1x1 layers: the patch is staged with 16-byte DMA pieces
    int bf16;           // bf16-operand kernel (desc.prec == MASIC_PREC_BF16 and Cout > 8)
    int pf;             // bf16: small patch -> next chunk's pixels prefetched in registers
    int KC, KClog;      // input channels per LDS chunk
    int TW, TWlog, SR, TH;
    int Cin_pad, Cout_pad;
    int PH, PW, PWp, PSZ;
    int max_taps;
    size_t lds_bytes;
};

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

ConvCfg choose_cfg(const masic_conv_desc_t& d, const ConvGeom* g, int nphase) {
    ConvCfg c{};
    if (d.Cout <= 4 && d.transposed && d.stride == 2 && d.KH == 5 && d.KW == 5 && d.pad == 2 && d.Cin >= 16 &&
        d.in_op == MASIC_INOP_NONE) {
        c.direct = 2; c.Cin_pad = round_up(d.Cin, 8); c.Cout_pad = 4;      // packed [ci][5][5][4]
        return c;
    }
    if (d.Cout <= 8 && d.Cin <= 8 && d.stride == 1 && d.KH == 5 && d.KW == 5 && d.pad == 2 && !d.masked &&
        d.in_op == MASIC_INOP_NONE && d.act != MASIC_ACT_SOFTMAX_C) {
        c.direct = 3; c.Cin_pad = d.Cin; c.Cout_pad = d.Cout <= 4 ? 4 : 8;  // packed [ci][5][5][4 or 8]
        return c;
    }
    if (d.Cout <= 8) {
        c.direct = 1; c.Cin_pad = d.Cin; c.Cout_pad = 8;
        return c;
    }
    int Hp = g[0].Hp, Wp = g[0].Wp, is = g[0].is;
    int span_h = 0, span_w = 0, max_taps = 0;
    for (int p = 0; p < nphase; ++p) {
        span_h = span_h > g[p].nth ? span_h : g[p].nth;
        span_w = span_w > g[p].ntw ? span_w : g[p].ntw;
        max_taps = max_taps > g[p].ntaps ? max_taps : g[p].ntaps;
    }
    c.max_taps = max_taps;
    c.TW = Wp > 16 ? 32 : (Wp > 8 ? 16 : 8);
    c.TWlog = ilog2(c.TW);
    c.SR = 32 / c.TW;
    // 128-channel blocks (2x4 MFMA tiles per wave: 0.75 LDS reads per MFMA) when Cout is a multiple of 128 and the
    // layer is big enough to still fill the chip; 64-channel blocks otherwise
    auto nblocks = [&](int wm, int wn) {
        int TH = c.SR * 2 * wn;
        return (long)ceil_div(Hp, TH) * ceil_div(Wp, c.TW) * (round_up(d.Cout, 64 * wm) / (64 * wm)) * d.B * nphase;
    };
    const bool m128 = round_up(d.Cout, 128) * 10 <= d.Cout * 11;          // <= 10 % padding waste with 128-wide blocks
    c.wvm = 2;
    if (d.Cout <= 32) { c.wvm = 1; c.wm = 1; c.wn = 2; }                    // 32 x 256 tile (CQE 32-channel layers)
    else if (d.Cout > 64 && d.Cout <= 96) { c.wvm = 1; c.wm = 3; c.wn = 2; } // 96 x 256 tile (CQE 96-channel layers)
    else if (m128 && nblocks(2, 4) >= 512) { c.wm = 2; c.wn = 4; }
    // bf16 1x1 layers are bound by operand staging, not MFMA: the widest tile amortises the weight stream and the
    // barriers best, and its small LDS footprint keeps every workgroup of a ~1-wave grid resident at once
    else if (m128 && d.prec == MASIC_PREC_BF16 && d.KH == 1 && d.KW == 1 && nblocks(2, 4) >= 192) { c.wm = 2; c.wn = 4; }
    else if (m128 && nblocks(2, 2) >= 512) { c.wm = 2; c.wn = 2; }
    else { c.wm = 1; c.wn = nblocks(1, 4) >= 768 ? 4 : (nblocks(1, 2) >= 512 ? 2 : 1); }
    const int wvn = 4 / c.wvm;
    const int BM = 32 * c.wm * c.wvm;
    c.Cout_pad = round_up(d.Cout, BM);
    // the per-channel patch must fit 24 DMA wave-instructions (MAXE = 6 per wave)
    while (c.wvm == 2 && c.wn > 1 && ((c.SR * wvn * c.wn - 1) * is + span_h) * ((c.TW - 1) * is + span_w) > 24 * 64) c.wn >>= 1;
    c.TH = c.SR * wvn * c.wn;
    c.PH = (c.TH - 1) * is + span_h;
    c.PW = (c.TW - 1) * is + span_w;
    c.PWp = c.PW;                                       // LDS-DMA writes 64 consecutive floats: no row padding
    if (d.prec == MASIC_PREC_BF16) {
        // [pixel][KC+8] bf16 patch in one LDS buffer (two workgroups per CU overlap staging and MFMA);
        // the whole patch must fit MAXP*256 pixels
        c.bf16 = 1;
        while (c.wn > 1 && c.PH * c.PW > 6 * 256) {
            c.wn >>= 1;
            c.TH = c.SR * wvn * c.wn;
            c.PH = (c.TH - 1) * is + span_h;
        }
        int KC = 32;
        while (KC > 16 && ((size_t)c.PH * c.PW * (KC + 8) * 2 > 64 * 1024 || KC > round_up(d.Cin, 16))) KC >>= 1;
        c.KC = KC; c.KClog = ilog2(KC);
        c.Cin_pad = round_up(d.Cin, KC);
        c.PSZ = c.PH * c.PW;
        c.buf_sz = 0;
        c.lds_bytes = (size_t)c.PH * c.PW * (KC + 8) * 2;
        c.pf = (c.PH * c.PW * (KC / 8) <= 1024) ? 1 : 0;
        return c;
    }
    c.PSZ = round_up(c.PH * c.PW, 64);                  // whole wave-instructions per channel
    const int groups = c.PSZ / 64;                      // DMA wave-instructions per channel
    const int epw = ceil_div(groups, 4);                // ... per wave
    const int cin2 = round_up(d.Cin, 2);
    // 1x1 layers (the GMM heads): rows are contiguous and 16-byte aligned -> dwordx4 DMA, 4x fewer instructions
    c.vec4 = (max_taps == 1 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.Wi % 4 == 0 && d.Cin >= 4 &&
              d.in_op == MASIC_INOP_NONE) ? 1 : 0;
    // two LDS buffers (the next chunk lands by DMA while this one is contracted); keep 2 blocks per CU
    int KC = 32;
    if (c.vec4) {
        while (KC > 4 && (KC * c.PSZ > 8192 || (size_t)2 * KC * (c.PSZ + BM) * 4 > 48 * 1024 || KC > round_up(d.Cin, 4))) KC >>= 1;
    } else {
        while (KC > 2 && ((size_t)2 * KC * (c.PSZ + max_taps * BM) * 4 > 80 * 1024 || KC > cin2 || KC * epw > 48)) KC >>= 1;
    }
    c.KC = KC; c.KClog = ilog2(KC);
    c.Cin_pad = round_up(d.Cin, KC);
    c.buf_sz = KC * c.PSZ + round_up(max_taps * KC, 256 / BM) * BM;
    c.lds_bytes = (size_t)2 * c.buf_sz * 4;
    return c;
}

// ------------------------------------------------------------------------------------------ pack

__global__ void pack_weight_kernel(const PackArgs a) {
    const ConvGeom g = a.gs[blockIdx.y];
    const size_t per_tap = (size_t)a.Cin_pad * a.Cout_pad;
    const size_t total = (size_t)g.ntaps * per_tap;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / per_tap);
        const int rem = (int)(i - (size_t)t * per_tap);
        const int ci = rem / a.Cout_pad, co = rem - ci * a.Cout_pad;
        float v = 0.0f;
        if (ci < a.Cin && co < a.Cout) {
            const int ta = t / g.ntw, tb = t - ta * g.ntw;
            const int kh = g.kh0 + ta * g.khs, kw = g.kw0 + tb * g.kws;
            const size_t src = a.transposed ? (((size_t)ci * a.Cout + co) * a.KH + kh) * a.KW + kw
                                            : (((size_t)co * a.Cin + ci) * a.KH + kh) * a.KW + kw;
            v = a.w[src];
        }
        a.wp[(size_t)g.tap_base * per_tap + i] = v;
    }
}

// ------------------------------------------------------------------------------------------ igemm
struct IgemmArgs {
    const float* x; const float* wp; const float* bias; const float* gate; const float* res1; const float* res2; float* y;
    int Cin, Hi, Wi, in_ctot, in_coff;
    int Cout, Ho, Wo, out_ctot, out_coff;
    int Cin_pad, Cout_pad, KC, KClog;
    int TW, TWlog, SR, TH, tiles_w;
    int PH, PW, PWp, PSZ, buf_sz, vec4;
    int in_op, act, gate_ctot, gate_c;
    GeomParams q;
    int nphase;
};

__device__ __attribute__((aligned(16))) float g_zero_word[4] = {0.0f, 0.0f, 0.0f, 0.0f};

// global -> LDS DMA (no VGPR round trip): per-lane source address, LDS destination = wave-uniform base + lane*size
__device__ __forceinline__ void dma4(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void dma16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- shared epilogue: bias, activation, gate, residual adds, NCHW store (32 consecutive pixels per half-wave).
// Every load (bias, gate, residuals) is issued BEFORE the arithmetic of its tile: a load inside the per-element loop
// would put an s_waitcnt vmcnt(0) in front of every store and serialise the epilogue on memory latency.
template <int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, const ConvGeom& g, f32x16 (&acc)[WM][WN], int b, int m0w,
                                               int r0, int c0, int wn, int jr, int jc, int h) {
    const size_t oplane = (size_t)a.Ho * a.Wo;
    float bv[WM][16];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = m0w + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            bv[m][e] = (a.bias != nullptr) ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.0f;
        }
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int r = r0 + (wn * WN + n) * a.SR + jr, c = c0 + jc;
        const bool pok = r < g.Hp && c < g.Wp;
        const int oh = pok ? r * g.os + g.oph : 0, ow = pok ? c * g.os + g.opw : 0;
        const size_t opix = (size_t)oh * a.Wo + ow;
        const float gv = (a.gate != nullptr) ? a.gate[((size_t)b * a.gate_ctot + a.gate_c) * oplane + opix] : 1.0f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            float rv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) rv[e] = 0.0f;
            if (a.res1 != nullptr) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = m0w + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    rv[e] = a.res1[((size_t)b * a.Cout + (co < a.Cout ? co : a.Cout - 1)) * oplane + opix];
                }
                if (a.res2 != nullptr) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int co = m0w + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        rv[e] += a.res2[((size_t)b * a.Cout + (co < a.Cout ? co : a.Cout - 1)) * oplane + opix];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = m0w + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = apply_act(acc[m][n][e] + bv[m][e], a.act) * gv + rv[e];
                if (pok && co < a.Cout) a.y[((size_t)b * a.out_ctot + a.out_coff + co) * oplane + opix] = v;
            }
        }
    }
}

constexpr int MAXE = 6;   // DMA wave-instructions per wave per channel plane (patch <= 1536 floats)
constexpr int MAXQ = 8;   // 16-byte-DMA wave-instructions per wave per chunk (1x1 layers: KC*PSZ <= 8192 floats)

template <int WVM, int WM, int WN, int INOP, bool VEC4>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const IgemmArgs a) {
    constexpr int BM = 32 * WM * WVM;           // output channels per block: WVM waves x WM x 32
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // two buffers, each: patch [KC][PSZ] (rows of PW floats, flattened) | weights [ntaps][KC][BM]
    const int wts_off = a.KC * a.PSZ;
    const int buf_sz = a.buf_sz;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WVM, wn = wave / WVM;
    const int j = lane & 31, h = lane >> 5;

    const int phase = blockIdx.x % a.nphase, tile = blockIdx.x / a.nphase;
    const ConvGeom g = make_geom(a.q, phase);
    const int tw_i = tile % a.tiles_w, th_i = tile / a.tiles_w;
    const int m0 = blockIdx.y * BM;
    const int b = blockIdx.z;
    const int r0 = th_i * a.TH, c0 = tw_i * a.TW;
    const int ih0 = r0 * g.is + g.dh_min, iw0 = c0 * g.is + g.dw_min;

    // ---- per-lane DMA source offsets, fixed for the whole K loop (-1: zero fill).
    //  dword mode : this wave's groups of 64 consecutive patch floats within one input channel plane
    //  VEC4 (1x1) : 16-byte pieces over the whole [KC][PSZ] chunk; per lane (channel-in-chunk, offset)
    constexpr int NOFF = VEC4 ? MAXQ : MAXE;
    const int groups = a.PSZ >> 6;
    const int nq = (a.KC * a.PSZ) >> 8;
    int goff[NOFF], gci[VEC4 ? MAXQ : 1];
#pragma unroll
    for (int i = 0; i < NOFF; ++i) {
        if (VEC4) {
            const int off = (wave + 4 * i) * 256 + lane * 4;
            const int ci = off / a.PSZ, rem = off - ci * a.PSZ;
            const int pr = rem / a.PW, pc = rem - pr * a.PW;
            const int ih = ih0 + pr, iw = iw0 + pc;
            const bool ok = (pr < a.PH) && ih >= 0 && ih < a.Hi && iw >= 0 && iw + 3 < a.Wi;
            gci[i] = ci;
            goff[i] = ok ? ci * (a.Hi * a.Wi) + ih * a.Wi + iw : -1;
        } else {
            const int e = (wave + 4 * i) * 64 + lane;
            const int pr = e / a.PW, pc = e - pr * a.PW;
            const int ih = ih0 + pr, iw = iw0 + pc;
            const bool ok = (pr < a.PH) && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
            goff[i] = ok ? ih * a.Wi + iw : -1;
        }
    }
    const size_t plane = (size_t)a.Hi * a.Wi;
    const float* xb = a.x + ((size_t)b * a.in_ctot + a.in_coff) * plane;
    // weights: 16-byte pieces, BM/4 lanes per (tap,ci) row of BM co -> 256/BM rows per wave-instruction
    constexpr int LPR = BM / 4, RPI = 64 / LPR;     // BM = 96: 24 lanes per row, 2 rows + 16 idle lanes per instruction
    const int wrows = g.ntaps * a.KC;
    const int wq = lane % LPR, wr = lane / LPR;

    auto issue = [&](int cc, float* buf) {
        if (VEC4) {
            const float* xc = xb + (size_t)cc * plane;
#pragma unroll
            for (int i = 0; i < NOFF; ++i) {
                const int q = wave + 4 * i;
                if (q < nq) {
                    const float* src = (goff[i] >= 0 && cc + gci[VEC4 ? i : 0] < a.Cin) ? xc + goff[i] : g_zero_word;
                    dma16(src, buf + q * 256);
                }
            }
        } else {
            for (int ci = 0; ci < a.KC; ++ci) {
                const bool cok = (cc + ci) < a.Cin;
                const float* xc = xb + (size_t)(cc + ci) * plane;
                float* dst = buf + ci * a.PSZ;
#pragma unroll
                for (int i = 0; i < NOFF; ++i) {
                    const int g = wave + 4 * i;
                    if (g < groups) {
                        const float* src = (cok && goff[i] >= 0) ? xc + goff[i] : g_zero_word;
                        dma4(src, dst + g * 64);
                    }
                }
            }
        }
        float* wdst = buf + wts_off;
        for (int r = wave * RPI; r < wrows; r += 4 * RPI) {   // this wave-instruction covers rows r..r+RPI-1
            int row = r + wr;
            row = row < wrows ? row : wrows - 1;                // tail lanes duplicate the last row into the pad rows
            const int t = row >> a.KClog, ci = row & (a.KC - 1);
            const float* src = a.wp + ((size_t)(g.tap_base + t) * a.Cin_pad + cc + ci) * a.Cout_pad + m0 + wq * 4;
            if (RPI * LPR == 64 || lane < RPI * LPR) dma16(src, wdst + r * BM);
        }
    };

    const int jr = j >> a.TWlog, jc = j & (a.TW - 1);
    // per-lane LDS offsets: A fragment (co = wm*32*WM + mi*32 + j of k-row h), B fragment (pixel j of sub-tile n, channel h)
    const int lane_w = h * BM + wm * (32 * WM) + j;
    int lane_p[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int r = (wn * WN + n) * a.SR + jr;
        lane_p[n] = h * a.PSZ + (r * g.is) * a.PW + jc * g.is;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    const int kh = a.KC >> 1, khlog = a.KClog - 1;
    const int nsteps = g.ntaps << khlog;
    const int tap0_off = (g.dh0 - g.dh_min) * a.PW + (g.dw0 - g.dw_min);
    const int chan_span = 2 * a.PSZ * (kh - 1);                       // back to channel pair 0 of the chunk
    const int row_step = g.dsh * a.PW - g.dsw * (g.ntw - 1);   // first tap of the next kernel row

    const int nchunks = a.Cin_pad >> a.KClog;
    issue(0, lds);
    for (int c = 0; c < nchunks; ++c) {
        float* cur = lds + (c & 1) * buf_sz;
        // own DMAs landed (vmcnt(0)) + everyone finished reading the other buffer -> barrier
        __syncthreads();
        if (c + 1 < nchunks) issue((c + 1) << a.KClog, lds + ((c + 1) & 1) * buf_sz);
        const float* patch = cur;
        const float* wts = cur + wts_off;
        // k-steps (tap, channel pair) of this chunk, software-pipelined with two fragment sets: the LDS reads of
        // step s+1 are issued before the MFMAs of step s.  Offsets advance incrementally: weights are contiguous
        // in step order; the patch offset moves by two channels inside a tap, else to the next (row-major) tap.
        int sw = 0, sp = tap0_off, kk2 = 0, tb = 0;
        auto advance = [&]() {
            sw += 2 * BM;
            if (++kk2 < kh) {
                sp += 2 * a.PSZ;
            } else {
                kk2 = 0;
                sp -= chan_span;
                if (++tb < g.ntw) sp += g.dsw;
                else { tb = 0; sp += row_step; }
            }
        };
        float a0[WM], b0[WN], a1[WM], b1[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) a0[m] = wts[lane_w + m * 32];
#pragma unroll
        for (int n = 0; n < WN; ++n) b0[n] = patch[sp + lane_p[n]];
        for (int st = 0; st < nsteps; st += 2) {
            const bool more1 = st + 1 < nsteps;
            if (more1) {
                advance();
#pragma unroll
                for (int m = 0; m < WM; ++m) a1[m] = wts[sw + lane_w + m * 32];
#pragma unroll
                for (int n = 0; n < WN; ++n) b1[n] = patch[sp + lane_p[n]];
            }
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], apply_inop(b0[n], INOP), acc[m][n], 0, 0, 0);
            if (!more1) break;
            if (st + 2 < nsteps) {
                advance();
#pragma unroll
                for (int m = 0; m < WM; ++m) a0[m] = wts[sw + lane_w + m * 32];
#pragma unroll
                for (int n = 0; n < WN; ++n) b0[n] = patch[sp + lane_p[n]];
            }
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], apply_inop(b1[n], INOP), acc[m][n], 0, 0, 0);
        }
    }

    igemm_epilogue<WM, WN>(a, g, acc, b, m0 + wm * (32 * WM), r0, c0, wn, jr, jc, h);
}

// ------------------------------------------------------------------------------------------ direct
struct DirectArgs {
    const float* x; const float* wp; const float* bias; const float* gate; const float* res1; const float* res2; float* y;
    int Cin, Hi, Wi, in_ctot, in_coff;
    int Cout, Ho, Wo, out_ctot, out_coff;
    int in_op, act, gate_ctot, gate_c;
    GeomParams q;
    int nphase;
};

template <int CO>
__global__ __launch_bounds__(256) void conv_direct_f32(const DirectArgs a) {
    const int phase = blockIdx.x % a.nphase;
    const ConvGeom g = make_geom(a.q, phase);
    const int pix = (blockIdx.x / a.nphase) * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int npix = g.Hp * g.Wp;
    const bool live = pix < npix;
    const int r = live ? pix / g.Wp : 0, c = live ? pix - r * g.Wp : 0;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.0f;
    const size_t plane = (size_t)a.Hi * a.Wi;
    const float* xb = a.x + ((size_t)b * a.in_ctot + a.in_coff) * plane;
    for (int t = 0; t < g.ntaps; ++t) {
        const int ta = t / g.ntw, tb = t - ta * g.ntw;
        const int ih = r * g.is + g.dh0 + ta * g.dsh;
        const int iw = c * g.is + g.dw0 + tb * g.dsw;
        const bool ok = live && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
        const float* xp = xb + (size_t)ih * a.Wi + iw;
        const float* wt = a.wp + (size_t)(g.tap_base + t) * a.Cin * 8;
        for (int ci = 0; ci < a.Cin; ++ci) {
            const float v = ok ? apply_inop(xp[(size_t)ci * plane], a.in_op) : 0.0f;
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] = fmaf(v, wt[ci * 8 + o], acc[o]);
        }
    }
    if (!live) return;
    const int oh = r * g.os + g.oph, ow = c * g.os + g.opw;
    const size_t oplane = (size_t)a.Ho * a.Wo, opix = (size_t)oh * a.Wo + ow;
    float gv = 1.0f;
    if (a.gate) gv = a.gate[((size_t)b * a.gate_ctot + a.gate_c) * oplane + opix];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = (o < a.Cout) ? acc[o] + (a.bias ? a.bias[o] : 0.0f) : 0.0f;
    if (a.act == MASIC_ACT_SOFTMAX_C) {   // mask2weights: softmax over the output channels (MASIC.py:497-502)
        float mx = acc[0], sum = 0.0f;
#pragma unroll
        for (int o = 1; o < CO; ++o) if (o < a.Cout) mx = fmaxf(mx, acc[o]);
#pragma unroll
        for (int o = 0; o < CO; ++o) if (o < a.Cout) { acc[o] = expf(acc[o] - mx); sum += acc[o]; }
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = acc[o] / sum;
    } else {
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = apply_act(acc[o], a.act);
    }
#pragma unroll
    for (int o = 0; o < CO; ++o)
        if (o < a.Cout) {
            float v = a.gate ? acc[o] * gv : acc[o];
            if (a.res1) v += a.res1[((size_t)b * a.Cout + o) * oplane + opix];
            if (a.res2) v += a.res2[((size_t)b * a.Cout + o) * oplane + opix];
            a.y[((size_t)b * a.out_ctot + a.out_coff + o) * oplane + opix] = v;
        }
}

// ------------------------------------------------------------------------------------------ deconv to <=4 channels
// ConvTranspose2d(Cin -> Cout<=4, k=5, s=2, p=2, op=1): the last layer of both synthesis transforms
// (decoder{1,2}.g_s_conv4, MASIC.py:542,596).  Cout = 3 cannot feed a 32x32 MFMA tile, and the layer is bound by
// reading its 128-channel input once, so it runs on the VALU: a 16x16 tile of INPUT pixels per workgroup, the
// input staged through LDS by DMA (8 channels per chunk, halo 1), each thread producing the 2x2 output pixels of
// its input pixel for all output channels (25 taps x Cout FMAs per input channel, weights as scalar operands).
struct Deconv4Args {
    const float* x; const float* wp; const float* bias; float* y;
    int Cin, Cin_pad, Hi, Wi, in_ctot, in_coff;
    int Cout, Ho, Wo, out_ctot, out_coff;
    int act, tiles_w;
};

__global__ __launch_bounds__(256) void deconv5s2_small_cout(const Deconv4Args a, const float* __restrict__ wpk) {
    constexpr int T = 16, LW = T + 2, PLANE = LW * LW, PSZ = 384, KC = 8;   // 324 floats per channel -> 6 DMA groups
    __shared__ __attribute__((aligned(16))) float lds[2 * KC * PSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = tid & 15, ty = tid >> 4;
    const int tile_x = blockIdx.x % a.tiles_w, tile_y = blockIdx.x / a.tiles_w;
    const int b = blockIdx.y;
    const int r = tile_y * T + ty, c = tile_x * T + tx;
    const int ih0 = tile_y * T - 1, iw0 = tile_x * T - 1;

    int goff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = (wave + 4 * i) * 64 + lane;
        const int pr = e / LW, pc = e - pr * LW;
        const int ih = ih0 + pr, iw = iw0 + pc;
        const bool ok = e < PLANE && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
        goff[i] = ok ? ih * a.Wi + iw : -1;
    }
    const size_t plane = (size_t)a.Hi * a.Wi;
    const float* xb = a.x + ((size_t)b * a.in_ctot + a.in_coff) * plane;
    auto issue = [&](int cc, float* buf) {
#pragma unroll
        for (int ci = 0; ci < KC; ++ci) {
            const bool cok = (cc + ci) < a.Cin;
            const float* xc = xb + (size_t)(cc + ci) * plane;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int g = wave + 4 * i;
                if (g < 6) dma4((cok && goff[i] >= 0) ? xc + goff[i] : g_zero_word, buf + ci * PSZ + g * 64);
            }
        }
    };

    float acc[2][2][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) (&acc[0][0][0])[i] = 0.0f;
    const int lbase = (ty + 1) * LW + tx + 1;
    const int nchunks = a.Cin_pad / KC;
    issue(0, lds);
    for (int ch = 0; ch < nchunks; ++ch) {
        const float* cur = lds + (ch & 1) * (KC * PSZ);
        __syncthreads();
        if (ch + 1 < nchunks) issue((ch + 1) * KC, lds + ((ch + 1) & 1) * (KC * PSZ));
#pragma unroll 1
        for (int ci = 0; ci < KC; ++ci) {
            const float* pl = cur + ci * PSZ + lbase;
            float v[3][3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v[dy][dx] = pl[(dy - 1) * LW + dx - 1];
            const float* w = wpk + (ch * KC + ci) * 100;       // [kh][kw][4]; const __restrict__ + uniform -> s_load
            // out(2r+ph, 2c+pw) += in(r+1-a, c+1-b) * W[ph+2a][pw+2b]
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                    for (int ta = 0; ta < 3 - ph; ++ta)
#pragma unroll
                        for (int tb = 0; tb < 3 - pw; ++tb) {
                            const float xv = v[2 - ta][2 - tb];
                            const float* wt = w + ((ph + 2 * ta) * 5 + (pw + 2 * tb)) * 4;
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[ph][pw][o] = fmaf(xv, wt[o], acc[ph][pw][o]);
                        }
        }
    }
    if (r >= a.Hi || c >= a.Wi) return;
    const size_t oplane = (size_t)a.Ho * a.Wo;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o >= a.Cout) break;
        const float bv = a.bias ? a.bias[o] : 0.0f;
        float* yo = a.y + ((size_t)b * a.out_ctot + a.out_coff + o) * oplane;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float2 st;
            st.x = apply_act(acc[ph][0][o] + bv, a.act);
            st.y = apply_act(acc[ph][1][o] + bv, a.act);
            *reinterpret_cast<float2*>(yo + (size_t)(2 * r + ph) * a.Wo + 2 * c) = st;
        }
    }
}

// packed layout for deconv5s2_small_cout: [ci (padded to 8)][kh][kw][4]
__global__ void pack_deconv4_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cin_pad, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin_pad * 100) return;
    const int ci = i / 100, rem = i - ci * 100, tap = rem >> 2, o = rem & 3;
    wp[i] = (ci < Cin && o < Cout) ? w[((size_t)ci * Cout + o) * 25 + tap] : 0.0f;
}

// ------------------------------------------------------------------------------------------ bf16-operand igemm
// Same implicit GEMM with bf16 operands on v_mfma_f32_32x32x16_bf16 (16x the f32 matrix rate), float32 accumulate,
// float32 NCHW tensors in HBM on both sides.  At this rate the contraction is no longer the bound -- operand delivery
// is -- so the staging differs from the f32 kernel:
//   * B (im2col) operand: a lane's fragment is 8 consecutive k = 8 consecutive input channels of one pixel, so the LDS
//     patch is [pixel][channel] bf16 (pixel pitch KC+8 elements: 16-byte pad against bank conflicts) and one
//     ds_read_b128 fetches a fragment.  NCHW float32 rows are read coalesced along W (one pixel per lane, 8 channel
//     planes per task), converted (v_cvt_pk_bf16_f32; |.| / round() fused here) and written with ds_write_b128:
//     the transpose happens in registers.
//   * A (weight) operand: straight from L2 into registers -- the weights are pre-packed as
//     [tap][ci/16][co][16 ci] bf16, so a wave's fragment load is 1 KiB contiguous -- prefetched one k-step ahead.
//     Keeping the 25-tap weights out of LDS is what lets the whole 5x5 patch of a 128 x 256 tile fit.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int MAXP = 6;    // pixel tasks per thread per 8-channel group (patch <= 1536 pixels)

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

template <int WVM, int WM, int WN, int INOP, bool PF>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16(const IgemmArgs a, const unsigned short* __restrict__ wpk) {
    constexpr int BM = 32 * WM * WVM;
    // A-fragment prefetch distance (k-steps): an L2 round trip under load is ~1-2 us, a k-step of a 1-MFMA wave tile
    // 13 ns -- the fewer MFMAs per k-step, the deeper the ring (same 32-64 VGPRs either way)
    constexpr int DEPTH = WM * WN >= 8 ? 4 : (WM * WN >= 2 ? 8 : 16);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned short* patch = reinterpret_cast<unsigned short*>(lds);          // [NP][KCP] bf16
    const int KCP = a.KC + 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WVM, wn = wave / WVM;
    const int j = lane & 31, h = lane >> 5;

    const int phase = blockIdx.x % a.nphase, tile = blockIdx.x / a.nphase;
    const ConvGeom g = make_geom(a.q, phase);
    const int tw_i = tile % a.tiles_w, th_i = tile / a.tiles_w;
    const int m0 = blockIdx.y * BM;
    const int b = blockIdx.z;
    const int r0 = th_i * a.TH, c0 = tw_i * a.TW;
    const int ih0 = r0 * g.is + g.dh_min, iw0 = c0 * g.is + g.dw_min;

    const int NP = a.PH * a.PW;
    const size_t plane = (size_t)a.Hi * a.Wi;
    const float* xb = a.x + ((size_t)b * a.in_ctot + a.in_coff) * plane;
    auto pixel_offset = [&](int p) {     // patch pixel -> offset in an input channel plane (-1: zero fill)
        const int pr = p / a.PW, pc = p - pr * a.PW;
        const int ih = ih0 + pr, iw = iw0 + pc;
        return (ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi) ? ih * a.Wi + iw : -1;
    };
    // ---- staging maps.
    //  !PF: pixel p = tid + 256 i for every 8-channel group in turn (big patches)
    //   PF: (pixel, group) tasks t = tid + 256 i, t < NP * KC/8 <= 1024 (small patches: next chunk prefetched in registers)
    constexpr int NT = PF ? 4 : MAXP;
    int goff[NT], gch_t[PF ? 4 : 1], loff[PF ? 4 : 1];
    const int ngroups = a.KC >> 3;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = tid + 256 * i;
        if (PF) {
            const int gq = t / NP, p = t - gq * NP;
            goff[i] = gq >= ngroups ? -2 : pixel_offset(p);
            gch_t[i] = gq * 8;
            loff[i] = p * KCP + gq * 8;
        } else {
            goff[i] = t >= NP ? -2 : pixel_offset(t);
        }
    }

    const int jr = j >> a.TWlog, jc = j & (a.TW - 1);
    int lane_p[WN];        // LDS element offset of this lane's pixel (tap 0) + its k half
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int r = (wn * WN + n) * a.SR + jr;
        lane_p[n] = ((r * g.is) * a.PW + jc * g.is) * KCP + 8 * h;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    const int ksteps = a.KC >> 4, kslog = ksteps >> 1;              // 16-channel MFMA steps per tap (1 or 2)
    const int nsteps = g.ntaps * ksteps;
    const int tap0_off = ((g.dh0 - g.dh_min) * a.PW + (g.dw0 - g.dw_min)) * KCP;
    const int row_step = (g.dsh * a.PW - g.dsw * (g.ntw - 1)) * KCP;
    const int col_step = g.dsw * KCP;
    const int cin16 = a.Cin_pad >> 4;
    // A fragment address: ((tap * cin16 + c16) * Cout_pad + co) * 16 + 8h   (bf16 elements)
    const size_t a_lane = ((size_t)m0 + wm * (32 * WM) + j) * 16 + 8 * h;
    const size_t a_tap_stride = (size_t)cin16 * a.Cout_pad * 16;
    const size_t a_c16_stride = (size_t)a.Cout_pad * 16;

    float pre[PF ? 4 : 1][8];
    // raw, unconditional loads from clamped (always valid) addresses; the zero-fill select happens at store time so that
    // nothing consumes a loaded value here (a consumer would put an s_waitcnt behind every single load)
    auto load_tasks = [&](int cc) {
#pragma unroll
        for (int i = 0; i < (PF ? 4 : 1); ++i) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int cg = cc + gch_t[i] + c;
                pre[i][c] = xb[(size_t)(cg < a.Cin ? cg : a.Cin - 1) * plane + (goff[i] >= 0 ? goff[i] : 0)];
            }
        }
    };
    auto store_tasks = [&](int cc) {
#pragma unroll
        for (int i = 0; i < (PF ? 4 : 1); ++i) {
            if (goff[i] == -2) continue;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                v[c] = (goff[i] >= 0 && cc + gch_t[i] + c < a.Cin) ? apply_inop(pre[i][c], INOP) : 0.0f;
            uint4 q;
            q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]);
            q.z = pack_bf16x2(v[4], v[5]); q.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4*>(patch + loff[i]) = q;
        }
    };
    if (PF) load_tasks(0);

    for (int cc = 0; cc < a.Cin_pad; cc += a.KC) {
        // ---- stage: NCHW float32 -> [pixel][channel] bf16
        if (PF) {
            store_tasks(cc);
        } else {
            for (int gch = 0; gch < a.KC; gch += 8) {
                // all loads of this 8-channel group first (up to 48 in flight per lane), then convert + write:
                // one memory latency per group instead of one per pixel task
                float v[NT][8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int cg = cc + gch + c;
                    if (cg < a.Cin) {                     // wave-uniform: channels beyond Cin (3-channel image layers) cost no loads
                        const float* xc = xb + (size_t)cg * plane;
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            const float t = xc[goff[i] >= 0 ? goff[i] : 0];
                            v[i][c] = goff[i] >= 0 ? t : 0.0f;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NT; ++i) v[i][c] = 0.0f;
                    }
                }
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (goff[i] == -2) continue;
                    uint4 q;
                    q.x = pack_bf16x2(apply_inop(v[i][0], INOP), apply_inop(v[i][1], INOP));
                    q.y = pack_bf16x2(apply_inop(v[i][2], INOP), apply_inop(v[i][3], INOP));
                    q.z = pack_bf16x2(apply_inop(v[i][4], INOP), apply_inop(v[i][5], INOP));
                    q.w = pack_bf16x2(apply_inop(v[i][6], INOP), apply_inop(v[i][7], INOP));
                    *reinterpret_cast<uint4*>(patch + (size_t)(tid + 256 * i) * KCP + gch) = q;
                }
            }
        }
        __syncthreads();
        if (PF && cc + a.KC < a.Cin_pad) load_tasks(cc + a.KC);        // in flight during the contraction
        // ---- contraction: k-steps (tap, 16 channels); A fragments stream from L2 through a DEPTH-deep register ring
        const unsigned short* wa = wpk + (size_t)g.tap_base * a_tap_stride + (size_t)(cc >> 4) * a_c16_stride + a_lane;
        auto a_offset = [&](int st) { return (size_t)(st >> kslog) * a_tap_stride + (size_t)(st & (ksteps - 1)) * a_c16_stride; };
        uint4 ring[DEPTH][WM];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if (u < nsteps) {
                const unsigned short* pa = wa + a_offset(u);
#pragma unroll
                for (int m = 0; m < WM; ++m) ring[u][m] = *reinterpret_cast<const uint4*>(pa + m * 32 * 16);
            }
        int sp = tap0_off, kk = 0, tb = 0;
        for (int st0 = 0; st0 < nsteps; st0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int st = st0 + u;
                if (st < nsteps) {
                    bf16x8 bfrag[WN], afr[WM];
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        bfrag[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(patch + sp + lane_p[n] + kk * 16));
#pragma unroll
                    for (int m = 0; m < WM; ++m) afr[m] = __builtin_bit_cast(bf16x8, ring[u][m]);
                    if (st + DEPTH < nsteps) {
                        const unsigned short* pa = wa + a_offset(st + DEPTH);
#pragma unroll
                        for (int m = 0; m < WM; ++m) ring[u][m] = *reinterpret_cast<const uint4*>(pa + m * 32 * 16);
                    }
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[m], bfrag[n], acc[m][n], 0, 0, 0);
                    if (++kk == ksteps) {
                        kk = 0;
                        if (++tb < g.ntw) sp += col_step;
                        else { tb = 0; sp += row_step; }
                    }
                }
            }
        }
        __syncthreads();
    }

    igemm_epilogue<WM, WN>(a, g, acc, b, m0 + wm * (32 * WM), r0, c0, wn, jr, jc, h);
}


// ------------------------------------------------------------------------------------------ 5x5 stride-1, few channels
// Conv2d / ConvTranspose2d(k=5, s=1, p=2) with Cin <= 8 and Cout <= CO (4 or 8): encoder2.pre_conv (6->3) and
// decoder2.after_conv (6->3, transposed; MASIC.py:559,600), and with CO = 8 the input gradient of after_conv (3->6).  Full-resolution, HBM-bound (reads Cin, writes Cout
// planes): a 16x16 output tile per workgroup, the (16+4)^2 input tile of every channel staged by DMA, weights packed
// [ci][kh][kw][4] (flipped for the transposed layer, so the kernel is always a correlation) as scalar operands.
struct Conv5s1Args {
    const float* x; const float* bias; const float* gate; const float* res1; float* y;
    int Cin, H, W, in_ctot, in_coff, Cout, out_ctot, out_coff, act, tiles_w, gate_ctot, gate_c;
    // fused form of encoder2.pre_conv + pre_gdn and decoder2.after_gdn + cat + after_conv (MASIC.py:574-576, :617-621):
    const float* xb;          // second source: channels split .. Cin-1 come from xb[B][Cin-split][H][W] (no torch.cat)
    int split;                // channels taken from x (0: everything from x)
    const float* gin_beta; const float* gin_gamma; int gin_inverse;      // 3-channel (I)GDN applied to x's channels on load
    const float* gout_beta; const float* gout_gamma; int gout_inverse;   // 3-channel (I)GDN applied to the output
    float beta_bound, gamma_bound, pedestal;
};

// y_i = x_i * (beta^_i + sum_j gamma^_ij x_j^2)^(-1/2 or +1/2) over 3 channels, same operation order as gdn_generic
__device__ __forceinline__ void gdn3(float (&v)[3], const float* beta, const float* gamma, int inverse, float bb, float gb, float ped) {
    float sq[3], o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) sq[k] = __fmul_rn(v[k], v[k]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float bv = fmaxf(beta[i], bb);
        float n = __fsub_rn(__fmul_rn(bv, bv), ped);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float g = fmaxf(gamma[i * 3 + k], gb);
            s = fmaf(__fsub_rn(__fmul_rn(g, g), ped), sq[k], s);
        }
        n += s;
        const float r = sqrtf(n);
        o[i] = inverse ? v[i] * r : v[i] * (1.0f / r);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = o[i];
}

template <int CO>
__global__ __launch_bounds__(256) void conv5s1_small(const Conv5s1Args a, const float* __restrict__ wpk) {
    constexpr int T = 16, LW = T + 4, PLANE = LW * LW, PSZ = 448;       // 400 floats per channel -> 7 DMA groups
    __shared__ __attribute__((aligned(16))) float lds[8 * PSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a wave takes rows wave, wave+4, wave+8, wave+12 of the tile: with the 20-float row pitch their 16-lane runs start 80 floats
    // = 16 banks apart, so the 64 lanes of a tap read hit 64 different banks (adjacent rows overlap in 12 of them: measured
    // half of the LDS cycles were conflicts)
    const int tx = tid & 15, ty = ((tid >> 4) & 3) * 4 + (tid >> 6);
    const int tile_x = blockIdx.x % a.tiles_w, tile_y = blockIdx.x / a.tiles_w;
    const int b = blockIdx.y;
    const int oy = tile_y * T + ty, ox = tile_x * T + tx;
    const size_t plane = (size_t)a.H * a.W;
    const float* xb = a.x + ((size_t)b * a.in_ctot + a.in_coff) * plane;
    const int split = a.xb != nullptr ? a.split : a.Cin;
    const float* xb2 = a.xb != nullptr ? a.xb + (size_t)b * (a.Cin - split) * plane : nullptr;
    for (int cg = wave; cg < a.Cin * 7; cg += 4) {
        const int ci = cg / 7, g = cg - ci * 7;
        const int e = g * 64 + lane;
        const int pr = e / LW, pc = e - pr * LW;
        const int ih = tile_y * T - 2 + pr, iw = tile_x * T - 2 + pc;
        const bool ok = e < PLANE && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        const float* src = ci < split ? xb + (size_t)ci * plane : xb2 + (size_t)(ci - split) * plane;
        if (e < PLANE) dma4(ok ? src + (size_t)ih * a.W + iw : g_zero_word, lds + ci * PSZ + g * 64);
    }
    __syncthreads();
    if (a.gin_beta != nullptr) {            // (I)GDN of the first three channels, per staged pixel (zero padding stays zero)
        for (int e = tid; e < PLANE; e += 256) {
            float v[3] = {lds[e], lds[PSZ + e], lds[2 * PSZ + e]};
            gdn3(v, a.gin_beta, a.gin_gamma, a.gin_inverse, a.beta_bound, a.gamma_bound, a.pedestal);
            lds[e] = v[0]; lds[PSZ + e] = v[1]; lds[2 * PSZ + e] = v[2];
        }
        __syncthreads();
    }
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.0f;
    const float* base = lds + ty * LW + tx;
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* w = wpk + ci * 25 * CO;
        const float* pl = base + ci * PSZ;
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const float xv = pl[kh * LW + kw];
#pragma unroll
                for (int o = 0; o < CO; ++o) acc[o] = fmaf(xv, w[(kh * 5 + kw) * CO + o], acc[o]);
            }
    }
    if (oy >= a.H || ox >= a.W) return;
    const size_t opix = (size_t)oy * a.W + ox;
    const float gv = a.gate ? a.gate[((size_t)b * a.gate_ctot + a.gate_c) * plane + opix] : 1.0f;
    if (a.gout_beta != nullptr) {           // 3-channel (I)GDN of the result (no activation / gate / residual in this form)
        float v[3] = {acc[0] + (a.bias ? a.bias[0] : 0.0f), acc[1] + (a.bias ? a.bias[1] : 0.0f), acc[2] + (a.bias ? a.bias[2] : 0.0f)};
        gdn3(v, a.gout_beta, a.gout_gamma, a.gout_inverse, a.beta_bound, a.gamma_bound, a.pedestal);
#pragma unroll
        for (int o = 0; o < 3; ++o) a.y[((size_t)b * a.out_ctot + a.out_coff + o) * plane + opix] = v[o];
        return;
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        if (o >= a.Cout) break;
        float v = apply_act(acc[o] + (a.bias ? a.bias[o] : 0.0f), a.act) * gv;
        if (a.res1) v += a.res1[((size_t)b * a.Cout + o) * plane + opix];
        a.y[((size_t)b * a.out_ctot + a.out_coff + o) * plane + opix] = v;
    }
}

// [ci][kh][kw][CO]; the transposed layer's taps are flipped so that the kernel correlates
__global__ void pack_conv5s1_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int transposed, int CO) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * 25 * CO) return;
    const int ci = i / (25 * CO), rem = i - ci * 25 * CO, tap = rem / CO, o = rem - tap * CO;
    float v = 0.0f;
    if (o < Cout) v = transposed ? w[((size_t)ci * Cout + o) * 25 + (24 - tap)] : w[((size_t)o * Cin + ci) * 25 + tap];
    wp[i] = v;
}


}  // namespace

extern "C" size_t masic_conv_packed_bytes(const masic_conv_desc_t* d) {
    if (check_desc(d) != MASIC_OK) return 0;
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const ConvCfg c = choose_cfg(*d, g, np);
    int taps = 0;
    for (int p = 0; p < np; ++p) taps += g[p].ntaps;
    if (c.direct == 2) return (size_t)c.Cin_pad * 100 * sizeof(float);
    if (c.direct == 3) return (size_t)c.Cin_pad * 25 * c.Cout_pad * sizeof(float);
    if (c.bf16) return (size_t)taps * c.Cin_pad * c.Cout_pad * sizeof(unsigned short);
    return (size_t)taps * c.Cin_pad * c.Cout_pad * sizeof(float);
}

extern "C" int masic_conv_variant(const masic_conv_desc_t* d, int* launches) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const ConvCfg c = choose_cfg(*d, g, np);
    if (launches) *launches = 1;   // all phases ride in one launch
    if (c.direct == 2) return 6;
    if (c.direct == 3) return 11;
    if (c.direct) return d->Cout <= 3 ? 0 : 1;
    if (c.bf16) return 10;
    if (c.wvm == 1) return c.wm == 1 ? 8 : 9;
    if (c.wm == 2) return c.wn == 4 ? 5 : 7;
    return c.wn == 4 ? 4 : (c.wn == 2 ? 3 : 2);
}

extern "C" int masic_conv_kernel_name(const masic_conv_desc_t* d, char* buf, size_t n) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    MASIC_REQUIRE(buf && n > 0, MASIC_ERR_ARG, "conv_kernel_name: null buffer");
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const ConvCfg c = choose_cfg(*d, g, np);
    if (c.direct == 2) snprintf(buf, n, "deconv5s2_small_cout");
    else if (c.direct == 3) snprintf(buf, n, "conv5s1_small");
    else if (c.direct) snprintf(buf, n, "conv_direct_f32<%d>", d->Cout <= 3 ? 3 : 8);
    else if (c.bf16) snprintf(buf, n, "conv_igemm_bf16<%d, %d, %d, %d, %s>", c.wvm, c.wm, c.wn, d->in_op, c.pf ? "true" : "false");
    else snprintf(buf, n, "conv_igemm_f32<%d, %d, %d, %d, %s>", c.wvm, c.wm, c.wn, d->in_op, c.vec4 ? "true" : "false");
    return MASIC_OK;
}

extern "C" int masic_conv_pack_weight(const float* w, void* w_packed, const masic_conv_desc_t* d, void* stream) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    MASIC_REQUIRE(w && w_packed, MASIC_ERR_ARG, "conv_pack_weight: null pointer");
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const ConvCfg c = choose_cfg(*d, g, np);
    if (c.direct == 2) {
        hipLaunchKernelGGL(pack_deconv4_kernel, dim3(ceil_div(c.Cin_pad * 100, 256)), dim3(256), 0, (hipStream_t)stream,
                           w, (float*)w_packed, d->Cin, c.Cin_pad, d->Cout);
        return masic_launch_status("conv_pack_weight");
    }
    if (c.direct == 3) {
        hipLaunchKernelGGL(pack_conv5s1_kernel, dim3(ceil_div(d->Cin * 25 * c.Cout_pad, 256)), dim3(256), 0, (hipStream_t)stream,
                           w, (float*)w_packed, d->Cin, d->Cout, d->transposed, c.Cout_pad);
        return masic_launch_status("conv_pack_weight");
    }
    PackArgs a{w, (float*)w_packed, d->Cin, d->Cout, d->KH, d->KW, c.Cin_pad, c.Cout_pad, d->transposed, {}};
    int max_taps = 0;
    for (int p = 0; p < np; ++p) {
        a.gs[p] = g[p];
        max_taps = max_taps > g[p].ntaps ? max_taps : g[p].ntaps;
    }
    const size_t total = (size_t)max_taps * c.Cin_pad * c.Cout_pad;           // the largest phase sizes the grid (grid-stride loops)
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (c.bf16) hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks, np), dim3(256), 0, (hipStream_t)stream, a, (unsigned short*)w_packed);
    else hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks, np), dim3(256), 0, (hipStream_t)stream, a);
    return masic_launch_status("conv_pack_weight");
}

extern "C" int masic_conv2d_fwd_ex(const float* x, const void* w_packed, const float* bias, const float* gate,
                                   const float* res1, const float* res2, float* y, const masic_conv_desc_t* d, void* stream);

extern "C" int masic_conv2d_fwd(const float* x, const void* w_packed, const float* bias, const float* gate,
                                float* y, const masic_conv_desc_t* d, void* stream) {
    return masic_conv2d_fwd_ex(x, w_packed, bias, gate, nullptr, nullptr, y, d, stream);
}

extern "C" int masic_conv2d_fwd_ex(const float* x, const void* w_packed, const float* bias, const float* gate,
                                   const float* res1, const float* res2, float* y, const masic_conv_desc_t* d, void* stream) {
    int rc = check_desc(d);
    if (rc != MASIC_OK) return rc;
    MASIC_REQUIRE(d->prec != MASIC_PREC_FP8, MASIC_ERR_UNSUPPORTED, "conv2d_fwd: fp8 operands exist on the F8K entry points only (masic_conv_f8k_fwd)");
    MASIC_REQUIRE(x && w_packed && y, MASIC_ERR_ARG, "conv2d_fwd: null pointer");
    ConvGeom g[4];
    const int np = build_geoms(*d, g);
    const ConvCfg c = choose_cfg(*d, g, np);
    hipStream_t st = (hipStream_t)stream;
    const GeomParams q = geom_params(*d);
    if (g[0].Hp <= 0 || g[0].Wp <= 0) return MASIC_OK;
    if (c.direct == 2) {
        MASIC_REQUIRE(gate == nullptr && res1 == nullptr && res2 == nullptr, MASIC_ERR_UNSUPPORTED, "conv2d_fwd: gate/residual on the small-Cout deconv path");
        MASIC_REQUIRE(d->Wo % 2 == 0, MASIC_ERR_SHAPE, "conv2d_fwd: odd output width");
        Deconv4Args a{x, (const float*)w_packed, bias, y, d->Cin, c.Cin_pad, d->Hi, d->Wi, d->in_ctot, d->in_coff,
                      d->Cout, d->Ho, d->Wo, d->out_ctot, d->out_coff, d->act, ceil_div(d->Wi, 16)};
        hipLaunchKernelGGL(deconv5s2_small_cout, dim3(ceil_div(d->Wi, 16) * ceil_div(d->Hi, 16), d->B), dim3(256), 0, st, a,
                           (const float*)w_packed);
        return masic_launch_status("conv2d_fwd");
    }
    if (c.direct == 3) {
        MASIC_REQUIRE(res2 == nullptr, MASIC_ERR_UNSUPPORTED, "conv2d_fwd: two residuals on the small 5x5 path");
        Conv5s1Args a{x, bias, gate, res1, y, d->Cin, d->Hi, d->Wi, d->in_ctot, d->in_coff, d->Cout, d->out_ctot, d->out_coff,
                      d->act, ceil_div(d->Wi, 16), d->gate_ctot, d->gate_c, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0.0f, 0.0f, 0.0f};
        if (c.Cout_pad == 4) hipLaunchKernelGGL(conv5s1_small<4>, dim3(ceil_div(d->Wi, 16) * ceil_div(d->Hi, 16), d->B), dim3(256), 0, st, a,
                                                (const float*)w_packed);
        else hipLaunchKernelGGL(conv5s1_small<8>, dim3(ceil_div(d->Wi, 16) * ceil_div(d->Hi, 16), d->B), dim3(256), 0, st, a,
                                (const float*)w_packed);
        return masic_launch_status("conv2d_fwd");
    }
    if (c.direct) {
        DirectArgs a{x, (const float*)w_packed, bias, gate, res1, res2, y,
                     d->Cin, d->Hi, d->Wi, d->in_ctot, d->in_coff,
                     d->Cout, d->Ho, d->Wo, d->out_ctot, d->out_coff,
                     d->in_op, d->act, d->gate_ctot, d->gate_c, q, np};
        dim3 grid(ceil_div(g[0].Hp * g[0].Wp, 256) * np, d->B);
        if (d->Cout <= 3) hipLaunchKernelGGL(conv_direct_f32<3>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(conv_direct_f32<8>, grid, dim3(256), 0, st, a);
        return masic_launch_status("conv2d_fwd");
    }
    {
        IgemmArgs a{x, (const float*)w_packed, bias, gate, res1, res2, y,
                    d->Cin, d->Hi, d->Wi, d->in_ctot, d->in_coff,
                    d->Cout, d->Ho, d->Wo, d->out_ctot, d->out_coff,
                    c.Cin_pad, c.Cout_pad, c.KC, c.KClog,
                    c.TW, c.TWlog, c.SR, c.TH, ceil_div(g[0].Wp, c.TW),
                    c.PH, c.PW, c.PWp, c.PSZ, c.buf_sz, c.vec4,
                    d->in_op, d->act, d->gate_ctot, d->gate_c, q, np};
        dim3 grid(ceil_div(g[0].Wp, c.TW) * ceil_div(g[0].Hp, c.TH) * np, c.Cout_pad / (32 * c.wm * c.wvm), d->B);
        if (c.bf16) {
#define BF16_LAUNCH(WV, WMV, WNV, OPV)                                                                                            \
    do {                                                                                                                          \
        if (c.pf) hipLaunchKernelGGL((conv_igemm_bf16<WV, WMV, WNV, OPV, true>), grid, dim3(256), c.lds_bytes, st, a,             \
                                     (const unsigned short*)w_packed);                                                           \
        else hipLaunchKernelGGL((conv_igemm_bf16<WV, WMV, WNV, OPV, false>), grid, dim3(256), c.lds_bytes, st, a,                 \
                                (const unsigned short*)w_packed);                                                                 \
    } while (0)
#define BF16_BY_OP(WV, WMV, WNV)                                                       \
    do {                                                                               \
        if (d->in_op == MASIC_INOP_ABS) BF16_LAUNCH(WV, WMV, WNV, MASIC_INOP_ABS);     \
        else if (d->in_op == MASIC_INOP_ROUND) BF16_LAUNCH(WV, WMV, WNV, MASIC_INOP_ROUND); \
        else BF16_LAUNCH(WV, WMV, WNV, MASIC_INOP_NONE);                               \
    } while (0)
            if (c.wvm == 1 && c.wm == 1) BF16_BY_OP(1, 1, 2);
            else if (c.wvm == 1) BF16_BY_OP(1, 3, 2);
            else if (c.wm == 2 && c.wn == 4) BF16_BY_OP(2, 2, 4);
            else if (c.wm == 2 && c.wn == 2) BF16_BY_OP(2, 2, 2);
            else if (c.wm == 2) BF16_BY_OP(2, 2, 1);
            else if (c.wn == 4) BF16_BY_OP(2, 1, 4);
            else if (c.wn == 2) BF16_BY_OP(2, 1, 2);
            else BF16_BY_OP(2, 1, 1);
#undef BF16_BY_OP
#undef BF16_LAUNCH
            return masic_launch_status("conv2d_fwd");
        }
#define IGEMM_LAUNCH(WV, WMV, WNV, OPV, V4) \
    hipLaunchKernelGGL((conv_igemm_f32<WV, WMV, WNV, OPV, V4>), grid, dim3(256), c.lds_bytes, st, a)
#define IGEMM_BY_OP(WV, WMV, WNV)                                                               \
    do {                                                                                        \
        if (d->in_op == MASIC_INOP_ABS) IGEMM_LAUNCH(WV, WMV, WNV, MASIC_INOP_ABS, false);      \
        else if (d->in_op == MASIC_INOP_ROUND) IGEMM_LAUNCH(WV, WMV, WNV, MASIC_INOP_ROUND, false); \
        else if (c.vec4) IGEMM_LAUNCH(WV, WMV, WNV, MASIC_INOP_NONE, true);                     \
        else IGEMM_LAUNCH(WV, WMV, WNV, MASIC_INOP_NONE, false);                                \
    } while (0)
        if (c.wvm == 1 && c.wm == 1) IGEMM_BY_OP(1, 1, 2);
        else if (c.wvm == 1) IGEMM_BY_OP(1, 3, 2);
        else if (c.wm == 2 && c.wn == 4) IGEMM_BY_OP(2, 2, 4);
        else if (c.wm == 2) IGEMM_BY_OP(2, 2, 2);
        else if (c.wn == 4) IGEMM_BY_OP(2, 1, 4);
        else if (c.wn == 2) IGEMM_BY_OP(2, 1, 2);
        else IGEMM_BY_OP(2, 1, 1);
#undef IGEMM_BY_OP
#undef IGEMM_LAUNCH
    }
    return masic_launch_status("conv2d_fwd");
}

// encoder2.pre_conv + pre_gdn and decoder2.after_gdn + cat + after_conv (MASIC.py:559-560, :574-576, :599-600, :617-621) as one
// launch each: Conv2d / ConvTranspose2d(6 -> 3, k5, s1, p2) whose six input channels come from two 3-channel tensors (no
// concat buffer), with an optional 3-channel (I)GDN on the first source while it is staged and / or on the result.
extern "C" int masic_conv5s1_pair_fwd(const float* xa, const float* xb, const float* w_packed, const float* bias,
                                      const float* gin_beta, const float* gin_gamma, int gin_inverse,
                                      const float* gout_beta, const float* gout_gamma, int gout_inverse, double beta_min,
                                      float* y, int B, int H, int W, void* stream) {
    MASIC_REQUIRE(xa && xb && w_packed && y, MASIC_ERR_ARG, "conv5s1_pair_fwd: null pointer");
    MASIC_REQUIRE(B > 0 && H > 0 && W > 0, MASIC_ERR_SHAPE, "conv5s1_pair_fwd: non-positive dimension");
    MASIC_REQUIRE((gin_beta == nullptr) == (gin_gamma == nullptr) && (gout_beta == nullptr) == (gout_gamma == nullptr), MASIC_ERR_ARG,
                  "conv5s1_pair_fwd: a GDN needs both beta and gamma");
    const double ped = 0x1p-36;
    Conv5s1Args a{xa, bias, nullptr, nullptr, y, 6, H, W, 3, 0, 3, 3, 0, MASIC_ACT_NONE, ceil_div(W, 16), 0, 0,
                  xb, 3, gin_beta, gin_gamma, gin_inverse, gout_beta, gout_gamma, gout_inverse,
                  (float)__builtin_sqrt(beta_min + ped), (float)__builtin_sqrt(ped), (float)ped};
    hipLaunchKernelGGL(conv5s1_small<4>, dim3(ceil_div(W, 16) * ceil_div(H, 16), B), dim3(256), 0, (hipStream_t)stream, a, (const float*)w_packed);
    return masic_launch_status("conv5s1_pair_fwd");
}
