// conv_geom.h -- phase geometry of the convolution family (shared by conv.hip and conv_f16k.hip); see conv.hip header.
#pragma once
#include "common.h"

namespace {


struct ConvGeom {
    int ntaps, nth, ntw;
    int dh0, dsh, dw0, dsw;   // input offset of tap (a,b): ih = r*is + dh0 + a*dsh
    int kh0, khs, kw0, kws;   // kernel index of tap (a,b): kh = kh0 + a*khs
    int is;                   // input step per phase-plane pixel
    int os, oph, opw;         // output coordinate: oh = r*os + oph
    int Hp, Wp;               // phase-plane size
    int tap_base;             // first tap of this phase in the packed weights
    int dh_min, dw_min;       // smallest tap offsets (patch origin)
};

// What is needed to derive a phase's geometry; small enough to ride in the kernel arguments so that one launch
// covers every phase (blockIdx.x % nphase) and the phases of a tile run next to each other (their interleaved
// output columns merge in L2).
struct GeomParams {
    int transposed, masked, KH, KW, stride, pad, Ho, Wo;
};

__host__ __device__ inline void phase_axis(int p, int K, int s, int pad, int& k0, int& nt, int& d0) {
    k0 = (p + pad) % s;              // first kernel index of this output parity
    nt = (K - k0 + s - 1) / s;       // taps along this axis
    d0 = (p + pad - k0) / s;         // input offset of tap 0 (then -1 per tap)
}

__host__ __device__ inline ConvGeom make_geom(const GeomParams& q, int phase) {
    ConvGeom g;
    if (!q.transposed) {
        g.nth = q.KH; g.ntw = q.KW; g.ntaps = q.KH * q.KW;
        if (q.masked) {                                   // layers.py:69-75, mask type 'A'
            g.ntaps = (q.KH / 2) * q.KW + q.KW / 2;
            g.nth = q.KH / 2 + 1;
        }
        g.dh0 = -q.pad; g.dsh = 1; g.dw0 = -q.pad; g.dsw = 1;
        g.kh0 = 0; g.khs = 1; g.kw0 = 0; g.kws = 1;
        g.is = q.stride; g.os = 1; g.oph = 0; g.opw = 0;
        g.Hp = q.Ho; g.Wp = q.Wo; g.tap_base = 0;
        g.dh_min = -q.pad; g.dw_min = -q.pad;
        return g;
    }
    const int s = q.stride;
    const int ph = phase / s, pw = phase - ph * s;
    phase_axis(ph, q.KH, s, q.pad, g.kh0, g.nth, g.dh0);
    phase_axis(pw, q.KW, s, q.pad, g.kw0, g.ntw, g.dw0);
    g.khs = s; g.kws = s; g.dsh = -1; g.dsw = -1;
    g.ntaps = g.nth * g.ntw;
    g.is = 1; g.os = s; g.oph = ph; g.opw = pw;
    g.Hp = (q.Ho - ph + s - 1) / s; g.Wp = (q.Wo - pw + s - 1) / s;
    g.dh_min = g.dh0 - (g.nth - 1); g.dw_min = g.dw0 - (g.ntw - 1);
    int base = 0;
    for (int p = 0; p < phase; ++p) {
        int k0, nh, nw, d0;
        phase_axis(p / s, q.KH, s, q.pad, k0, nh, d0);
        phase_axis(p % s, q.KW, s, q.pad, k0, nw, d0);
        base += nh * nw;
    }
    g.tap_base = base;
    return g;
}

GeomParams geom_params(const masic_conv_desc_t& d) {
    return GeomParams{d.transposed, d.masked, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo};
}

// Phases of a layer (host side). Returns the number of phases (1 for Conv2d, s*s for transposed).
int build_geoms(const masic_conv_desc_t& d, ConvGeom* g) {
    const GeomParams q = geom_params(d);
    const int n = d.transposed ? d.stride * d.stride : 1;
    for (int p = 0; p < n; ++p) g[p] = make_geom(q, p);
    return n;
}

struct PackArgs {
    const float* w; float* wp;
    int Cin, Cout, KH, KW, Cin_pad, Cout_pad, transposed;
    ConvGeom gs[4];           // one launch packs every phase: blockIdx.y = phase (a training step re-packs every weight every iteration)
};

// packed bf16 weights: [phase-tap][ci/16][co (padded to BM)][16 ci]
__global__ void pack_weight_bf16_kernel(const PackArgs a, unsigned short* __restrict__ wp) {
    const ConvGeom g = a.gs[blockIdx.y];
    const size_t per_tap = (size_t)a.Cin_pad * a.Cout_pad;
    const size_t total = (size_t)g.ntaps * per_tap;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / per_tap);
        const size_t rem = i - (size_t)t * per_tap;
        const int c16 = (int)(rem / ((size_t)a.Cout_pad * 16));
        const int rem2 = (int)(rem - (size_t)c16 * a.Cout_pad * 16);
        const int co = rem2 >> 4, ci = c16 * 16 + (rem2 & 15);
        float v = 0.0f;
        if (ci < a.Cin && co < a.Cout) {
            const int ta = t / g.ntw, tb = t - ta * g.ntw;
            const int kh = g.kh0 + ta * g.khs, kw = g.kw0 + tb * g.kws;
            const size_t src = a.transposed ? (((size_t)ci * a.Cout + co) * a.KH + kh) * a.KW + kw
                                            : (((size_t)co * a.Cin + ci) * a.KH + kh) * a.KW + kw;
            v = a.w[src];
        }
        const __bf16 bv = (__bf16)v;
        wp[(size_t)g.tap_base * per_tap + i] = __builtin_bit_cast(unsigned short, bv);
    }
}

inline int check_desc(const masic_conv_desc_t* d) {
    MASIC_REQUIRE(d != nullptr, MASIC_ERR_ARG, "conv: null descriptor");
    MASIC_REQUIRE(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->Hi > 0 && d->Wi > 0, MASIC_ERR_SHAPE,
                  "conv: non-positive dimension");
    MASIC_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH <= 5 && d->KW <= 5, MASIC_ERR_UNSUPPORTED, "conv: kernel size %dx%d", d->KH, d->KW);
    MASIC_REQUIRE(d->stride == 1 || d->stride == 2, MASIC_ERR_UNSUPPORTED, "conv: stride %d", d->stride);
    MASIC_REQUIRE(d->in_coff >= 0 && d->in_coff + d->Cin <= d->in_ctot, MASIC_ERR_SHAPE, "conv: input channel view out of range");
    MASIC_REQUIRE(d->out_coff >= 0 && d->out_coff + d->Cout <= d->out_ctot, MASIC_ERR_SHAPE, "conv: output channel view out of range");
    MASIC_REQUIRE(!(d->masked && d->transposed), MASIC_ERR_UNSUPPORTED, "conv: masked transposed conv");
    int ho, wo;
    if (!d->transposed) {
        ho = (d->Hi + 2 * d->pad - d->KH) / d->stride + 1;
        wo = (d->Wi + 2 * d->pad - d->KW) / d->stride + 1;
    } else {
        ho = (d->Hi - 1) * d->stride - 2 * d->pad + d->KH + (d->stride - 1);
        wo = (d->Wi - 1) * d->stride - 2 * d->pad + d->KW + (d->stride - 1);
    }
    MASIC_REQUIRE(ho == d->Ho && wo == d->Wo, MASIC_ERR_SHAPE, "conv: output size %dx%d given, %dx%d expected", d->Ho, d->Wo, ho, wo);
    MASIC_REQUIRE(d->act != MASIC_ACT_SOFTMAX_C || d->Cout <= 8, MASIC_ERR_UNSUPPORTED, "conv: channel softmax needs Cout <= 8");
    MASIC_REQUIRE(d->prec == MASIC_PREC_F32 || d->prec == MASIC_PREC_BF16 || d->prec == MASIC_PREC_FP8, MASIC_ERR_UNSUPPORTED, "conv: precision %d not built", d->prec);
    return MASIC_OK;
}

}  // namespace
