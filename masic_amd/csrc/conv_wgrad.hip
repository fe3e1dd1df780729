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
// A workgroup owns a 64(a) x 64(q) tile of dW for ONE kernel row kh and a strided share of the pixel tiles (2 coarse
// rows x 32 columns).  Per pixel tile the P rows and the two Q rows that this kernel row touches are staged into LDS
// by global->LDS DMA (channel pitch odd, so that the 32 lanes of an MFMA operand -- 32 different channels -- hit 32
// different banks); each wave issues v_mfma_f32_32x32x2_f32 with k = a pixel pair, one accumulator per kw.  Partial
// tiles of all workgroups are reduced with float atomics into a [kh][kw][a][q] workspace (128-byte segments per
// half wave = full atomic rate) and a final pass transposes it into the weight layout.
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
    int s, pad, KH;
    int tiles_w, tiles_h, ntiles, nsplit;
    int q_tiles;
    int PWq, QG, QS;      // fine columns per row of the patch, DMA groups per channel, channel pitch (odd)
};

constexpr int PS = 65;    // P-tile channel pitch (64 pixels + 1)

template <int KW>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Pt = lds;                 // [64][PS]
    float* Qt = lds + 64 * PS;       // [64][QS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wq = wave >> 1;
    const int j = lane & 31, h = lane >> 5;
    const int a0 = (blockIdx.x / a.q_tiles) * 64, q0 = (blockIdx.x % a.q_tiles) * 64;
    const int kh = blockIdx.y;

    f32x16 acc[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;

    const size_t cplane = (size_t)a.Hc * a.Wc, fplane = (size_t)a.Hf * a.Wf;
    const int tiles_per_img = a.tiles_w * a.tiles_h;
    // lane -> (row, col) inside the P tile and inside each Q DMA group
    const int prr = lane >> 5, pcc = lane & 31;

    for (int tile = blockIdx.z; tile < a.ntiles; tile += a.nsplit) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int r0 = (trem / a.tiles_w) * 2, c0 = (trem % a.tiles_w) * 32;
        // ---- stage P: 64 channels x (2 rows x 32 cols)
        {
            const int r = r0 + prr, c = c0 + pcc;
            const bool pok = r < a.Hc && c < a.Wc;
            const float* pb = a.P + ((size_t)b * a.p_ctot + a.p_coff) * cplane + (size_t)r * a.Wc + c;
            for (int ch = wave; ch < 64; ch += 4) {
                const bool ok = pok && (a0 + ch) < a.CA;
                dma4(ok ? pb + (size_t)(a0 + ch) * cplane : g_zero_wg, Pt + ch * PS);
            }
        }
        // ---- stage Q: 64 channels x (2 fine rows of this kernel row x PWq cols)
        {
            const float* qb = a.Q + ((size_t)b * a.q_ctot + a.q_coff) * fplane;
            for (int g = 0; g < a.QG; ++g) {
                const int e = g * 64 + lane;
                const int rr = e / a.PWq, pc = e - rr * a.PWq;
                const int fh = (r0 + rr) * a.s + kh - a.pad, fw = c0 * a.s - a.pad + pc;
                const bool qok = rr < 2 && (r0 + rr) < a.Hc && fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf;
                const float* src = qb + (size_t)fh * a.Wf + fw;
                for (int ch = wave; ch < 64; ch += 4) {
                    const bool ok = qok && (q0 + ch) < a.CQ;
                    dma4(ok ? src + (size_t)(q0 + ch) * fplane : g_zero_wg, Qt + ch * a.QS + g * 64);
                }
            }
        }
        __syncthreads();     // vmcnt(0) + barrier: the tile has landed
        // ---- contraction over the 64 pixels of the tile (k = pixel pair)
        const float* pa = Pt + (wa * 32 + j) * PS + h;
        const float* qa = Qt + (wq * 32 + j) * a.QS + h * a.s;
#pragma unroll 2
        for (int kk = 0; kk < 32; ++kk) {
            const int px = 2 * kk;                      // pixel of lane half 0; half 1 takes px+1 (same row: 32 is even)
            const int rr = px >> 5, cc = px & 31;
            const float av = pa[px];
            const float* qrow = qa + rr * a.PWq + cc * a.s;
            float bv[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) bv[k] = qrow[k];
#pragma unroll
            for (int k = 0; k < KW; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[k], acc[k], 0, 0, 0);
        }
        __syncthreads();     // everyone done reading before the next tile overwrites
    }

    // ---- reduce into the [kh][kw][a][q] workspace (q contiguous: 128-byte segments per half wave)
    const int q = q0 + wq * 32 + j;
    if (q < a.CQ) {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            float* wsk = a.ws + (size_t)(kh * KW + k) * a.CA * a.CQ;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ai = a0 + wa * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (ai < a.CA) atomicAdd(wsk + (size_t)ai * a.CQ + q, acc[k][e]);
            }
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
    MASIC_REQUIRE(d->KW == 1 || d->KW == 3 || d->KW == 5, MASIC_ERR_UNSUPPORTED, "conv2d_wgrad: kernel width %d", d->KW);
    MASIC_REQUIRE(d->KH >= 1 && d->KH <= 5, MASIC_ERR_UNSUPPORTED, "conv2d_wgrad: kernel height %d", d->KH);
    MASIC_REQUIRE(d->stride == 1 || d->stride == 2, MASIC_ERR_UNSUPPORTED, "conv2d_wgrad: stride %d", d->stride);
    MASIC_REQUIRE(d->in_coff >= 0 && d->in_coff + d->Cin <= d->in_ctot, MASIC_ERR_SHAPE, "conv2d_wgrad: input view out of range");
    hipStream_t st = (hipStream_t)stream;
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
    a.tiles_w = ceil_div(a.Wc, 32); a.tiles_h = ceil_div(a.Hc, 2);
    a.ntiles = a.tiles_w * a.tiles_h * a.B;
    const int a_tiles = ceil_div(a.CA, 64);
    a.q_tiles = ceil_div(a.CQ, 64);
    const int base = a_tiles * a.q_tiles * a.KH;
    int nsplit = ceil_div(1024, base);
    if (nsplit > a.ntiles) nsplit = a.ntiles;
    if (nsplit < 1) nsplit = 1;
    a.nsplit = nsplit;
    a.PWq = 31 * a.s + d->KW;
    a.QG = ceil_div(2 * a.PWq, 64);
    a.QS = a.QG * 64 + 1;
    const size_t lds = (size_t)(64 * PS + 64 * a.QS) * sizeof(float);
    const size_t wbytes = masic_conv2d_wgrad_workspace_bytes(d);
    if (hipMemsetAsync(workspace, 0, wbytes, st) != hipSuccess) {
        masic_set_error("conv2d_wgrad: workspace memset failed");
        return MASIC_ERR_LAUNCH;
    }
    dim3 grid(a_tiles * a.q_tiles, a.KH, nsplit);
    if (d->KW == 5) hipLaunchKernelGGL(conv_wgrad_f32<5>, grid, dim3(256), lds, st, a);
    else if (d->KW == 3) hipLaunchKernelGGL(conv_wgrad_f32<3>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(conv_wgrad_f32<1>, grid, dim3(256), lds, st, a);
    const int T = d->KH * d->KW, AQ = a.CA * a.CQ;
    const size_t total = (size_t)T * AQ;
    int tb = (int)((total + 255) / 256);
    if (tb > 4096) tb = 4096;
    hipLaunchKernelGGL(wgrad_transpose_kernel, dim3(tb), dim3(256), 0, st, (const float*)workspace, dw, T, AQ);
    return masic_launch_status("conv2d_wgrad");
}
