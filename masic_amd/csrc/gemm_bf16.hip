// gemm_bf16.hip -- the 1x1-convolution stacks of the GMM parameter heads as register-streamed bf16 GEMMs (gfx950).
//
// Reference: coremasic/mywork/MASIC.py:330-468 (`gmm_hyper_y{1,2}_same_resolution`): per view three 3-layer stacks of
// 1x1 (transposed) convolutions, 768/960 -> 1152 -> 768|960 -> 960 channels at latent resolution; 17 of the 80.6 GMAC
// of a forward.  With bf16 operands these GEMMs are bound by operand delivery, so between the layers of a stack the
// activations stay on the device in the layout the matrix cores want:
//     F16K: [B][C/16][H*W][16] bf16   (k-chunk major, 16 consecutive channels of one pixel = one 32-byte record)
// which is exactly the layout of the packed weights ([ci/16][co][16]).  A lane's MFMA fragment (8 consecutive k of one
// row) is then one 16-byte load, a wave's fragment load is 1 KiB contiguous for BOTH operands, and the kernel needs no
// LDS and no barrier: each wave streams its A (weights) and B (pixels) fragments from L2 through a register ring and
// issues v_mfma_f32_32x32x16_bf16 into a 64(co) x 128(pixel) accumulator tile; the 4 waves of a workgroup share
// operands through L1.  Epilogue: bias + ReLU/LeakyReLU, then either F16K bf16 for the next layer of the stack or
// float32 NCHW for the consumer (the GMM likelihood kernel).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// float32 NCHW (channel view) -> F16K bf16
__global__ __launch_bounds__(256) void nchw_to_f16k_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                                           int C, int Cpad, int HW, int ctot, int coff, int op) {
    // one thread per (pixel, 16-channel record): 16 coalesced plane reads, all issued before the first use, and ONE whole 32-byte record
    // written (2 KiB contiguous per wave).  (Round 2's form gave a thread half a record: every wave wrote 16-byte pieces at a 32-byte
    // stride and the other half came from another workgroup -- 4.0 TB/s; tools/bench_elementwise.py.)
    const int b = blockIdx.z, c16 = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* xb = x + ((size_t)b * ctot + coff) * HW + p;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c16 * 16 + i;
        v[i] = xb[(size_t)(c < C ? c : C - 1) * HW];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = c16 * 16 + i < C ? apply_inop(v[i], op) : 0.0f;
    uint4 q0, q1;
    q0.x = pack2(v[0], v[1]); q0.y = pack2(v[2], v[3]); q0.z = pack2(v[4], v[5]); q0.w = pack2(v[6], v[7]);
    q1.x = pack2(v[8], v[9]); q1.y = pack2(v[10], v[11]); q1.z = pack2(v[12], v[13]); q1.w = pack2(v[14], v[15]);
    uint4* yb = reinterpret_cast<uint4*>(y + (((size_t)b * (Cpad >> 4) + c16) * HW + p) * 16);
    yb[0] = q0;
    yb[1] = q1;
}

struct GemmArgs {
    const unsigned short* x;    // F16K [B][Cin_pad/16][HW][16]
    const unsigned short* w;    // packed [Cin_pad/16][Cout_pad][16]
    const float* bias;
    unsigned short* y16;        // F16K output [B][Cout_pad16/16][HW][16]   (or null)
    float* y32;                 // float32 NCHW output view                  (or null)
    int Cin_pad, Cout, Cout_pad, Cout_pad16, HW, out_ctot, out_coff, act;
};

// block = 4 waves (2 x 2): 128 co x 256 pixels; wave = 64 co x 128 px (WM = 2, WN = 4)
__global__ __launch_bounds__(256, 2) void gemm1x1_bf16_kernel(const GemmArgs a) {
    constexpr int WM = 2, WN = 4, DEPTH = 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * 128 + wm * 64;
    const int p0 = blockIdx.x * 256 + wn * 128;
    const int nk = a.Cin_pad >> 4;

    // per-lane element offsets inside one k-chunk slab
    unsigned aoff[WM], boff[WN];
#pragma unroll
    for (int m = 0; m < WM; ++m) aoff[m] = (unsigned)(m0 + m * 32 + j) * 16u + 8u * h;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        int p = p0 + n * 32 + j;
        p = p < a.HW ? p : a.HW - 1;                       // clamp: out-of-range pixels are computed and dropped
        boff[n] = (unsigned)p * 16u + 8u * h;
    }
    const unsigned short* wb = a.w;
    const unsigned short* xb = a.x + (size_t)b * nk * a.HW * 16;
    const unsigned wslab = (unsigned)a.Cout_pad * 16u, xslab = (unsigned)a.HW * 16u;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    uint4 ra[DEPTH][WM], rb[DEPTH][WN];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
        if (u < nk) {
#pragma unroll
            for (int m = 0; m < WM; ++m) ra[u][m] = *reinterpret_cast<const uint4*>(wb + u * wslab + aoff[m]);
#pragma unroll
            for (int n = 0; n < WN; ++n) rb[u][n] = *reinterpret_cast<const uint4*>(xb + u * xslab + boff[n]);
        }
    for (int k0 = 0; k0 < nk; k0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int k = k0 + u;
            if (k < nk) {
                bf16x8 af[WM], bfr[WN];
#pragma unroll
                for (int m = 0; m < WM; ++m) af[m] = __builtin_bit_cast(bf16x8, ra[u][m]);
#pragma unroll
                for (int n = 0; n < WN; ++n) bfr[n] = __builtin_bit_cast(bf16x8, rb[u][n]);
                if (k + DEPTH < nk) {
#pragma unroll
                    for (int m = 0; m < WM; ++m) ra[u][m] = *reinterpret_cast<const uint4*>(wb + (unsigned)(k + DEPTH) * wslab + aoff[m]);
#pragma unroll
                    for (int n = 0; n < WN; ++n) rb[u][n] = *reinterpret_cast<const uint4*>(xb + (unsigned)(k + DEPTH) * xslab + boff[n]);
                }
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---- epilogue.  acc[m][n][e]: channel m0 + 32m + (e&3) + 8(e>>2) + 4h, pixel p0 + 32n + j
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        float bvm[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = m0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            bvm[e] = (a.bias != nullptr) ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.0f;
        }
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = apply_act(acc[m][n][e] + bvm[e], a.act);
    }
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int p = p0 + n * 32 + j;
        if (p >= a.HW) continue;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            if (a.y16 != nullptr) {
                // 4 consecutive channels (e&3) of this lane -> 8 bytes of the pixel's 32-byte record
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c0 = m0 + m * 32 + 8 * g4 + 4 * h;      // multiple of 4
                    if (c0 < a.Cout_pad16) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float t = acc[m][n][4 * g4 + i];
                            v[i] = (c0 + i) < a.Cout ? t : 0.0f;
                        }
                        uint2 q;
                        q.x = pack2(v[0], v[1]); q.y = pack2(v[2], v[3]);
                        unsigned short* dst = a.y16 + (((size_t)b * (a.Cout_pad16 >> 4) + (c0 >> 4)) * a.HW + p) * 16 + (c0 & 15);
                        *reinterpret_cast<uint2*>(dst) = q;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = m0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (co < a.Cout)
                        a.y32[((size_t)b * a.out_ctot + a.out_coff + co) * a.HW + p] = acc[m][n][e];
                }
            }
        }
    }
}

// packed weights for the GEMM: [ci/16][co (padded to 128)][16] bf16 from a 1x1 Conv2d [Cout,Cin] or ConvTranspose2d [Cin,Cout]
__global__ void pack_gemm_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout,
                                        int Cin_pad, int Cout_pad, int transposed) {
    const size_t total = (size_t)Cin_pad * Cout_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = (int)(i / ((size_t)Cout_pad * 16));
        const int rem = (int)(i - (size_t)c16 * Cout_pad * 16);
        const int co = rem >> 4, ci = c16 * 16 + (rem & 15);
        float v = 0.0f;
        if (ci < Cin && co < Cout) v = transposed ? w[(size_t)ci * Cout + co] : w[(size_t)co * Cin + ci];
        const __bf16 bv = (__bf16)v;
        wp[i] = __builtin_bit_cast(unsigned short, bv);
    }
}

}  // namespace

extern "C" size_t masic_f16k_bytes(int B, int C, int HW) { return (size_t)B * round_up(C, 16) * HW * sizeof(unsigned short); }

extern "C" int masic_nchw_to_f16k_op(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int in_op, void* stream);
extern "C" int masic_nchw_to_f16k(const float* x, void* y, int B, int C, int HW, int ctot, int coff, void* stream) {
    return masic_nchw_to_f16k_op(x, y, B, C, HW, ctot, coff, MASIC_INOP_NONE, stream);
}
extern "C" int masic_nchw_to_f16k_op(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int in_op, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "nchw_to_f16k: null pointer");
    MASIC_REQUIRE(coff >= 0 && coff + C <= ctot, MASIC_ERR_SHAPE, "nchw_to_f16k: view out of range");
    const int Cpad = round_up(C, 16);
    hipLaunchKernelGGL(nchw_to_f16k_kernel, dim3(ceil_div(HW, 256), Cpad / 16, B), dim3(256), 0, (hipStream_t)stream,
                       x, (unsigned short*)y, C, Cpad, HW, ctot, coff, in_op);
    return masic_launch_status("nchw_to_f16k");
}

extern "C" size_t masic_gemm1x1_packed_bytes(int Cin, int Cout) {
    return (size_t)round_up(Cin, 16) * round_up(Cout, 128) * sizeof(unsigned short);
}

extern "C" int masic_gemm1x1_pack_weight(const float* w, void* wp, int Cin, int Cout, int transposed, void* stream) {
    MASIC_REQUIRE(w && wp, MASIC_ERR_ARG, "gemm1x1_pack_weight: null pointer");
    const int Cin_pad = round_up(Cin, 16), Cout_pad = round_up(Cout, 128);
    const size_t total = (size_t)Cin_pad * Cout_pad;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_gemm_weight_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wp, Cin, Cout,
                       Cin_pad, Cout_pad, transposed);
    return masic_launch_status("gemm1x1_pack_weight");
}

extern "C" int masic_gemm1x1_bf16_fwd(const void* x_f16k, const void* w_packed, const float* bias, void* y_f16k, float* y_nchw,
                                      int B, int Cin, int Cout, int HW, int out_ctot, int out_coff, int act, void* stream) {
    MASIC_REQUIRE(x_f16k && w_packed, MASIC_ERR_ARG, "gemm1x1_bf16_fwd: null pointer");
    MASIC_REQUIRE((y_f16k != nullptr) != (y_nchw != nullptr), MASIC_ERR_ARG, "gemm1x1_bf16_fwd: exactly one output format");
    MASIC_REQUIRE(act == MASIC_ACT_NONE || act == MASIC_ACT_RELU || act == MASIC_ACT_LEAKY, MASIC_ERR_ARG, "gemm1x1_bf16_fwd: activation");
    if (y_nchw) MASIC_REQUIRE(out_coff >= 0 && out_coff + Cout <= out_ctot, MASIC_ERR_SHAPE, "gemm1x1_bf16_fwd: output view out of range");
    GemmArgs a{(const unsigned short*)x_f16k, (const unsigned short*)w_packed, bias, (unsigned short*)y_f16k, y_nchw,
               round_up(Cin, 16), Cout, round_up(Cout, 128), round_up(Cout, 16), HW, out_ctot, out_coff, act};
    dim3 grid(ceil_div(HW, 256), a.Cout_pad / 128, B);
    hipLaunchKernelGGL(gemm1x1_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return masic_launch_status("gemm1x1_bf16_fwd");
}
