// f16k_ops.hip -- layout-side helpers of the F16K ([B][C/16][H*W][16] bf16) chains of Independent_EN (gfx950 / MI355X).
//
// The CQE network (reference coremasic/mywork/MASIC.py:1436-1501) works at full picture resolution on 32 / 64 / 96 channels;
// with bf16 operands its 3x3 convolutions run on conv_f16k.hip with the activations in F16K between them.  What the reference
// writes as `torch.cat((a * w[:, 1:2], warp(b) * w[:, 0:1]), dim=-3)` (:1470-1482) is done here by writing each operand
// straight into its channel slice of the concatenated F16K buffer:
//   f16k_gate_kernel   dst[:, coff : coff+C] = (warp?)(src) * gate[:, gc]    (src F16K; gate float32 [B, G, H, W] or none;
//                      warp = kornia warp_perspective with the sampling arithmetic of warp.hip, 4 taps x one 32-byte record
//                      per 16-channel block: whole records, no strided gathers)
//   nchw_to_f16k_view  float32 NCHW -> a channel slice of an F16K buffer
//   f16k_to_nchw       F16K channel slice -> float32 NCHW (for the 96 -> 3 output layer, which runs on the NCHW kernels)
// All HBM-bound: bytes in + bytes out.
#include "common.h"

namespace {

__device__ __forceinline__ float bf_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// one thread per (pixel, 8-channel half record)
__global__ __launch_bounds__(256) void f16k_gate_kernel(const unsigned short* __restrict__ src, const float* __restrict__ gate,
                                                        const float* __restrict__ minv, unsigned short* __restrict__ dst,
                                                        int C, int H, int W, int dst_ctot, int dst_coff, int gate_ctot, int gate_c, int align_corners) {
    const int b = blockIdx.z, c8 = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (pix >= HW) return;
    const float g = gate != nullptr ? gate[((size_t)b * gate_ctot + gate_c) * HW + pix] : 1.0f;
    const uint4* sb = reinterpret_cast<const uint4*>(src + (((size_t)b * (C >> 4) + (c8 >> 1)) * HW) * 16) + (c8 & 1);   // + 2 * pixel
    float v[8];
    if (minv == nullptr) {
        const uint4 q = sb[2 * (size_t)pix];
        v[0] = bf_lo(q.x); v[1] = bf_hi(q.x); v[2] = bf_lo(q.y); v[3] = bf_hi(q.y);
        v[4] = bf_lo(q.z); v[5] = bf_hi(q.z); v[6] = bf_lo(q.w); v[7] = bf_hi(q.w);
    } else {
        // the sampling location: warp.hip's float32 arithmetic, operation for operation (no FMA contraction)
        const int oy = pix / W, ox = pix - oy * W;
        const float* m = minv + b * 9;
        const float gx = __fmul_rn(__fsub_rn(__fdiv_rn((float)ox, (float)(W - 1)), 0.5f), 2.0f);
        const float gy = __fmul_rn(__fsub_rn(__fdiv_rn((float)oy, (float)(H - 1)), 0.5f), 2.0f);
        const float X = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[0]), __fmul_rn(gy, m[1])), m[2]);
        const float Y = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[3]), __fmul_rn(gy, m[4])), m[5]);
        const float Z = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[6]), __fmul_rn(gy, m[7])), m[8]);
        const float scale = fabsf(Z) > 1e-8f ? __fdiv_rn(1.0f, __fadd_rn(Z, 1e-8f)) : 1.0f;
        const float nx = __fmul_rn(X, scale), ny = __fmul_rn(Y, scale);
        const float fx = masic_grid_unnormalize(nx, W, align_corners);
        const float fy = masic_grid_unnormalize(ny, H, align_corners);
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float wx = __fsub_rn(fx, x0f), wy = __fsub_rn(fy, y0f);
        const float ex = __fsub_rn(1.0f, wx), ey = __fsub_rn(1.0f, wy);
        const float wt[4] = {__fmul_rn(ex, ey), __fmul_rn(wx, ey), __fmul_rn(ex, wy), __fmul_rn(wx, wy)};
        const float lim = 1.0e9f;
        const int x0 = (int)fminf(fmaxf(x0f, -lim), lim), y0 = (int)fminf(fmaxf(y0f, -lim), lim);
        const bool finite = (fx == fx) && (fy == fy);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {          // nw, ne, sw, se in warp.hip's accumulation order
            const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
            if (finite && xx >= 0 && xx < W && yy >= 0 && yy < H) {
                const uint4 q = sb[2 * ((size_t)yy * W + xx)];
                const float s[8] = {bf_lo(q.x), bf_hi(q.x), bf_lo(q.y), bf_hi(q.y), bf_lo(q.z), bf_hi(q.z), bf_lo(q.w), bf_hi(q.w)};
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __fadd_rn(v[i], __fmul_rn(s[i], wt[t]));
            }
        }
    }
    uint4 o;
    o.x = pack2(v[0] * g, v[1] * g); o.y = pack2(v[2] * g, v[3] * g); o.z = pack2(v[4] * g, v[5] * g); o.w = pack2(v[6] * g, v[7] * g);
    const int dc8 = (dst_coff >> 3) + c8;
    reinterpret_cast<uint4*>(dst + (((size_t)b * (dst_ctot >> 4) + (dc8 >> 1)) * HW) * 16)[2 * (size_t)pix + (dc8 & 1)] = o;
}

// in_op: |x| / round(x) on the way; gate: multiply by gate[b][gate_c][pixel] (float32 [B][gate_ctot][HW]) before the rounding to bf16 --
// round(y) * gate written straight into its slice of an F16K concat buffer is what `quantize(..., out=cat, gate=...)` + a conversion
// of the whole buffer computed in two passes (MASIC.py:827: the gated concat in front of the right view's entropy-parameter heads)
__global__ __launch_bounds__(256) void nchw_to_f16k_view_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int C, int HW,
                                                                int ctot, int coff, int dst_ctot, int dst_coff, int in_op,
                                                                const float* __restrict__ gate, int gate_ctot, int gate_c) {
    const int b = blockIdx.z, c8 = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* xb = x + ((size_t)b * ctot + coff) * HW + p;
    const float gv = gate != nullptr ? gate[((size_t)b * gate_ctot + gate_c) * HW + p] : 1.0f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        v[i] = c < C ? apply_inop(xb[(size_t)c * HW], in_op) * gv : 0.0f;
    }
    uint4 q;
    q.x = pack2(v[0], v[1]); q.y = pack2(v[2], v[3]); q.z = pack2(v[4], v[5]); q.w = pack2(v[6], v[7]);
    const int dc8 = (dst_coff >> 3) + c8;
    reinterpret_cast<uint4*>(y + (((size_t)b * (dst_ctot >> 4) + (dc8 >> 1)) * HW) * 16)[2 * (size_t)p + (dc8 & 1)] = q;
}

// OUT16: the result stays bf16 (NCHW): what the bf16-input form of the 5x5 stride-2 weight-gradient kernel reads (conv_wgrad.hip) --
// half the bytes written here and read there
template <bool OUT16>
__global__ __launch_bounds__(256) void f16k_to_nchw_kernel(const unsigned short* __restrict__ x, void* __restrict__ yv, int C, int HW,
                                                           int src_ctot, int src_coff, int ctot, int coff) {
    const int b = blockIdx.z, c8 = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int sc8 = (src_coff >> 3) + c8;
    const uint4 q = reinterpret_cast<const uint4*>(x + (((size_t)b * (src_ctot >> 4) + (sc8 >> 1)) * HW) * 16)[2 * (size_t)p + (sc8 & 1)];
    if constexpr (OUT16) {
        const unsigned short v[8] = {(unsigned short)(q.x & 0xffff), (unsigned short)(q.x >> 16), (unsigned short)(q.y & 0xffff), (unsigned short)(q.y >> 16),
                                     (unsigned short)(q.z & 0xffff), (unsigned short)(q.z >> 16), (unsigned short)(q.w & 0xffff), (unsigned short)(q.w >> 16)};
        unsigned short* yb = (unsigned short*)yv + ((size_t)b * ctot + coff) * HW + p;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c8 * 8 + i < C) yb[(size_t)(c8 * 8 + i) * HW] = v[i];
    } else {
        const float v[8] = {bf_lo(q.x), bf_hi(q.x), bf_lo(q.y), bf_hi(q.y), bf_lo(q.z), bf_hi(q.z), bf_lo(q.w), bf_hi(q.w)};
        float* yb = (float*)yv + ((size_t)b * ctot + coff) * HW + p;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c8 * 8 + i < C) yb[(size_t)(c8 * 8 + i) * HW] = v[i];
    }
}

// g' = g * act'(y) on F16K buffers (y = the forward activation's output: its sign is the pre-activation's)
__global__ __launch_bounds__(256) void f16k_act_bwd_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y, uint4* __restrict__ out, size_t n16, float slope) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 a = g[i], m = y[i];
        const unsigned av[4] = {a.x, a.y, a.z, a.w}, mv[4] = {m.x, m.y, m.z, m.w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = bf_lo(av[k]) * (bf_lo(mv[k]) > 0.0f ? 1.0f : slope);
            const float hi = bf_hi(av[k]) * (bf_hi(mv[k]) > 0.0f ? 1.0f : slope);
            o[k] = pack2(lo, hi);
        }
        out[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// per-channel sums of an F16K tensor: block (pixel chunk, 16-channel block, image) -> float32 partials [B * chunks][C], then a reduce
__global__ __launch_bounds__(256) void f16k_channel_sum_stage1(const uint4* __restrict__ x, float* __restrict__ partial, int C16, int HW, int chunks) {
    const int chunk = blockIdx.x, cb = blockIdx.y, b = blockIdx.z;
    const int per = (HW + chunks - 1) / chunks;
    const int lo = chunk * per, hi = lo + per < HW ? lo + per : HW;
    const uint4* p = x + ((size_t)b * C16 + cb) * HW * 2;          // 2 uint4 per record
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int half = threadIdx.x & 1;                                // which 8 channels of the record
    for (int px = lo + (threadIdx.x >> 1); px < hi; px += 128) {
        const uint4 q = p[2 * (size_t)px + half];
        acc[0] += bf_lo(q.x); acc[1] += bf_hi(q.x); acc[2] += bf_lo(q.y); acc[3] += bf_hi(q.y);
        acc[4] += bf_lo(q.z); acc[5] += bf_hi(q.z); acc[6] += bf_lo(q.w); acc[7] += bf_hi(q.w);
    }
    __shared__ float red[256][9];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 16) {                                          // channel c of the block: half c >> 3, slot c & 7
        const int c = threadIdx.x;
        float s = 0.0f;
        for (int t = (c >> 3); t < 256; t += 2) s += red[t][c & 7];
        partial[((size_t)(b * chunks + chunk)) * (C16 * 16) + cb * 16 + c] = s;
    }
}
// one block per 16 channels: 256 threads = 16 channels x 16 row lanes, then a tree over the row lanes (double: order-independent result)
__global__ __launch_bounds__(256) void f16k_channel_sum_stage2(const float* __restrict__ partial, float* __restrict__ out, int C, int n) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), lane_r = threadIdx.x >> 4;
    double s = 0.0;
    for (int i = lane_r; i < n; i += 16) s += (double)partial[(size_t)i * C + c];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k >= 16; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x < 16) out[c] = (float)red[threadIdx.x];
}

// g' = g * act'(y) AND the per-channel sums of g' (bf16-rounded, as stored): f16k_act_bwd_kernel + f16k_channel_sum_stage1 in one pass --
// the block layout of the latter (pixel chunk, 16-channel block, image), two loads in flight per lane and operand
__global__ __launch_bounds__(256) void f16k_act_bwd_sum_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y, uint4* __restrict__ out, float* __restrict__ partial,
                                                               int C16, int HW, int chunks, float slope) {
    const int chunk = blockIdx.x, cb = blockIdx.y, b = blockIdx.z;
    const int per = (HW + chunks - 1) / chunks;
    const int lo = chunk * per, hi = lo + per < HW ? lo + per : HW;
    const size_t base = ((size_t)b * C16 + cb) * HW * 2;          // 2 uint4 per record
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int half = threadIdx.x & 1;
    for (int px = lo + (threadIdx.x >> 1); px < hi; px += 256) {
        const bool two = px + 128 < hi;
        const size_t i0 = base + 2 * (size_t)px + half, i1 = i0 + 256;
        const uint4 a0 = g[i0], m0 = y[i0];
        uint4 a1 = make_uint4(0u, 0u, 0u, 0u), m1 = a1;
        if (two) { a1 = g[i1]; m1 = y[i1]; }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !two) break;
            const uint4 a = r ? a1 : a0, m = r ? m1 : m0;
            const unsigned av[4] = {a.x, a.y, a.z, a.w}, mv[4] = {m.x, m.y, m.z, m.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = pack2(bf_lo(av[k]) * (bf_lo(mv[k]) > 0.0f ? 1.0f : slope), bf_hi(av[k]) * (bf_hi(mv[k]) > 0.0f ? 1.0f : slope));
                acc[2 * k] += bf_lo(o[k]);
                acc[2 * k + 1] += bf_hi(o[k]);
            }
            out[r ? i1 : i0] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __shared__ float red[256][9];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 16) {
        const int c = threadIdx.x;
        float s = 0.0f;
        for (int t = (c >> 3); t < 256; t += 2) s += red[t][c & 7];
        partial[((size_t)(b * chunks + chunk)) * (C16 * 16) + cb * 16 + c] = s;
    }
}

}  // namespace

// out = g * act'(y), all F16K of n bf16 elements (n % 8 == 0); slope: 0.01 LeakyReLU, 0 ReLU
extern "C" int masic_f16k_act_bwd(const void* g, const void* y, void* out, size_t n, float slope, void* stream) {
    MASIC_REQUIRE(g && y && out && n % 8 == 0, MASIC_ERR_ARG, "f16k_act_bwd: null pointer or n %% 8 != 0");
    const size_t n16 = n / 8;
    int nb = (int)((n16 + 255) / 256);
    if (nb > 8192) nb = 8192;
    if (n16) hipLaunchKernelGGL(f16k_act_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const uint4*)g, (const uint4*)y, (uint4*)out, n16, slope);
    return masic_launch_status("f16k_act_bwd");
}

constexpr int F16K_CS_CHUNKS = 32;
extern "C" size_t masic_f16k_channel_sum_workspace_bytes(int B, int C) { return (size_t)B * F16K_CS_CHUNKS * round_up(C, 16) * sizeof(float); }
// out[c] = sum over (b, pixel) of an F16K tensor [B][C/16][HW][16] (the bias gradient of a layer whose dy is kept in F16K)
extern "C" int masic_f16k_channel_sum(const void* x, float* out, void* workspace, int B, int C, int HW, void* stream) {
    MASIC_REQUIRE(x && out && workspace && B > 0 && C > 0 && C % 16 == 0 && HW > 0, MASIC_ERR_ARG, "f16k_channel_sum: bad argument");
    hipLaunchKernelGGL(f16k_channel_sum_stage1, dim3(F16K_CS_CHUNKS, C / 16, B), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (float*)workspace, C / 16, HW, F16K_CS_CHUNKS);
    hipLaunchKernelGGL(f16k_channel_sum_stage2, dim3(C / 16), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, C, B * F16K_CS_CHUNKS);
    return masic_launch_status("f16k_channel_sum");
}

// out = g * act'(y) and sums[c] = sum over (b, pixel) of out, one pass over the tensors + the finishing reduce (workspace:
// masic_f16k_channel_sum_workspace_bytes(B, C)): the elementwise pass and the bias-gradient pass of a residual block's backward in one
extern "C" int masic_f16k_act_bwd_sum(const void* g, const void* y, void* out, float* sums, void* workspace, int B, int C, int HW, float slope, void* stream) {
    MASIC_REQUIRE(g && y && out && sums && workspace && B > 0 && C > 0 && C % 16 == 0 && HW > 0, MASIC_ERR_ARG, "f16k_act_bwd_sum: bad argument");
    hipLaunchKernelGGL(f16k_act_bwd_sum_kernel, dim3(F16K_CS_CHUNKS, C / 16, B), dim3(256), 0, (hipStream_t)stream, (const uint4*)g, (const uint4*)y, (uint4*)out,
                       (float*)workspace, C / 16, HW, F16K_CS_CHUNKS, slope);
    hipLaunchKernelGGL(f16k_channel_sum_stage2, dim3(C / 16), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, sums, C, B * F16K_CS_CHUNKS);
    return masic_launch_status("f16k_act_bwd_sum");
}

// dst[:, dst_coff : dst_coff + C] = (minv ? warp(src, minv) : src) * (gate ? gate[:, gate_c] : 1); src: F16K of exactly C channels,
// C % 8 == 0 and dst_coff % 8 == 0; minv: the [B,3,3] normalised sampling matrices of masic_warp_matrix / masic_amd.homography.
extern "C" int masic_f16k_gate(const void* src, const float* gate, const float* minv, void* dst, int B, int C, int H, int W,
                               int dst_ctot, int dst_coff, int gate_ctot, int gate_c, void* stream) {
    MASIC_REQUIRE(src && dst, MASIC_ERR_ARG, "f16k_gate: null pointer");
    MASIC_REQUIRE(B > 0 && C > 0 && C % 16 == 0 && H > 0 && W > 0 && dst_ctot % 16 == 0 && dst_coff % 8 == 0 && dst_coff >= 0 && dst_coff + C <= dst_ctot,
                  MASIC_ERR_SHAPE, "f16k_gate: channel views (C %% 16, dst_coff %% 8, inside dst_ctot)");
    MASIC_REQUIRE(gate == nullptr || (gate_c >= 0 && gate_c < gate_ctot), MASIC_ERR_SHAPE, "f16k_gate: gate channel out of range");
    hipLaunchKernelGGL(f16k_gate_kernel, dim3(ceil_div(H * W, 256), C / 8, B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, gate, minv, (unsigned short*)dst, C, H, W, dst_ctot, dst_coff, gate_ctot, gate_c, masic_warp_align_corners_value());
    return masic_launch_status("f16k_gate");
}

extern "C" int masic_nchw_to_f16k_view(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int dst_ctot, int dst_coff, void* stream) {
    return masic_nchw_to_f16k_view_op(x, y, B, C, HW, ctot, coff, dst_ctot, dst_coff, MASIC_INOP_NONE, nullptr, 0, 0, stream);
}

// ... with |x| / round(x) applied and an optional per-pixel gate (see the kernel)
extern "C" int masic_nchw_to_f16k_view_op(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int dst_ctot, int dst_coff, int in_op,
                                          const float* gate, int gate_ctot, int gate_c, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "nchw_to_f16k_view: null pointer");
    MASIC_REQUIRE(coff >= 0 && coff + C <= ctot && dst_ctot % 16 == 0 && dst_coff % 8 == 0 && dst_coff >= 0 && dst_coff + round_up(C, 8) <= dst_ctot,
                  MASIC_ERR_SHAPE, "nchw_to_f16k_view: view out of range");
    MASIC_REQUIRE(gate == nullptr || (gate_c >= 0 && gate_c < gate_ctot), MASIC_ERR_SHAPE, "nchw_to_f16k_view: gate channel out of range");
    hipLaunchKernelGGL(nchw_to_f16k_view_kernel, dim3(ceil_div(HW, 256), ceil_div(C, 8), B), dim3(256), 0, (hipStream_t)stream,
                       x, (unsigned short*)y, C, HW, ctot, coff, dst_ctot, dst_coff, in_op, gate, gate_ctot, gate_c);
    return masic_launch_status("nchw_to_f16k_view");
}

extern "C" int masic_f16k_to_nchw(const void* x, float* y, int B, int C, int HW, int src_ctot, int src_coff, int ctot, int coff, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "f16k_to_nchw: null pointer");
    MASIC_REQUIRE(coff >= 0 && coff + C <= ctot && src_ctot % 16 == 0 && src_coff % 8 == 0 && src_coff >= 0 && src_coff + C <= src_ctot,
                  MASIC_ERR_SHAPE, "f16k_to_nchw: view out of range");
    hipLaunchKernelGGL(f16k_to_nchw_kernel<false>, dim3(ceil_div(HW, 256), ceil_div(C, 8), B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, y, C, HW, src_ctot, src_coff, ctot, coff);
    return masic_launch_status("f16k_to_nchw");
}

// the same with a bf16 NCHW result (y: [B][ctot][HW] bf16)
extern "C" int masic_f16k_to_nchw_bf16(const void* x, void* y, int B, int C, int HW, int src_ctot, int src_coff, int ctot, int coff, void* stream) {
    MASIC_REQUIRE(x && y, MASIC_ERR_ARG, "f16k_to_nchw_bf16: null pointer");
    MASIC_REQUIRE(coff >= 0 && coff + C <= ctot && src_ctot % 16 == 0 && src_coff % 8 == 0 && src_coff >= 0 && src_coff + C <= src_ctot,
                  MASIC_ERR_SHAPE, "f16k_to_nchw_bf16: view out of range");
    hipLaunchKernelGGL(f16k_to_nchw_kernel<true>, dim3(ceil_div(HW, 256), ceil_div(C, 8), B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, y, C, HW, src_ctot, src_coff, ctot, coff);
    return masic_launch_status("f16k_to_nchw_bf16");
}
