// m2w.hip -- mask2weights_EN of the CQE network in one kernel (gfx950 / MI355X).
//
// Reference: coremasic/mywork/MASIC.py:1411-1434 -- four 3x3 stride-1 convolutions 1 -> 2 -> 4 -> 4 -> 2 with ReLUs on a
// full-resolution homography mask, then a softmax over the two output channels; called twice per Independent_EN forward (mask_R,
// mask_L, :1466-1467).  As four launches of the small-channel direct convolution (conv.hip: conv_direct_f32) it costs 76 + 101 +
// 101 + 76 us per mask at 8 x 512^2 -- 6 % of the bf16 CQE forward -- against ~350 MACs per pixel of arithmetic.
//
// Here a workgroup produces a 32 x 32 tile of gates from the 40 x 40 mask patch around it with every intermediate activation in
// LDS (57 KiB).  A thread computes ALL output channels of a position from one pass over its 3 x 3 x Cin neighbourhood; weights are
// uniform (scalar) loads.  Positions of an intermediate layer that lie outside the picture are zeros (each layer pads its own
// input), not values computed from the extended patch.  The accumulation order per output -- taps outer, input channels inner,
// fmaf, bias added last -- is conv_direct_f32's, so the result is bit-identical to the four-launch form (the float32 parity path
// uses this kernel too).
#include "common.h"

namespace {

constexpr int T = 32;

template <int CIN, int COUT, int SW_IN, int SW_OUT, bool LAST>
__device__ __forceinline__ void m2w_layer(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ w, const float* __restrict__ bias,
                                          int oy0, int ox0, int H, int W, float* __restrict__ out, size_t out_plane, int tid) {
    // src: [CIN][SW_IN][SW_IN] in LDS, dst: [COUT][SW_OUT][SW_OUT] (SW_OUT = SW_IN - 2); (oy0, ox0): picture position of dst[.][0][0]
    for (int idx = tid; idx < SW_OUT * SW_OUT; idx += 256) {
        const int yy = idx / SW_OUT, xx = idx - yy * SW_OUT;
        const int gy = oy0 + yy, gx = ox0 + xx;
        const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
        if (inside) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float v = src[(ci * SW_IN + yy + t / 3) * SW_IN + xx + t % 3];
#pragma unroll
                    for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v, w[(o * CIN + ci) * 9 + t], acc[o]);
                }
            }
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) dst[(o * SW_OUT + yy) * SW_OUT + xx] = inside ? fmaxf(acc[o] + bias[o], 0.0f) : 0.0f;
        } else if (inside) {
            // softmax over the output channels (conv_direct_f32's MASIC_ACT_SOFTMAX_C epilogue)
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[o] = acc[o] + bias[o];
            float mx = acc[0], sum = 0.0f;
#pragma unroll
            for (int o = 1; o < COUT; ++o) mx = fmaxf(mx, acc[o]);
#pragma unroll
            for (int o = 0; o < COUT; ++o) { acc[o] = expf(acc[o] - mx); sum += acc[o]; }
#pragma unroll
            for (int o = 0; o < COUT; ++o) out[(size_t)o * out_plane + (size_t)gy * W + gx] = acc[o] / sum;
        }
    }
}

struct M2wArgs {
    const float* mask; float* gates;
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
    int H, W, tiles_w;
};

__global__ __launch_bounds__(256) void m2w_en_kernel(const M2wArgs a) {
    __shared__ float s0[(T + 8) * (T + 8)];
    __shared__ float s1[2 * (T + 6) * (T + 6)];
    __shared__ float s2[4 * (T + 4) * (T + 4)];
    __shared__ float s3[4 * (T + 2) * (T + 2)];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_w, tx = blockIdx.x - ty * a.tiles_w;
    const int r0 = ty * T, c0 = tx * T;
    const size_t plane = (size_t)a.H * a.W;
    const float* m = a.mask + (size_t)b * plane;
    for (int idx = tid; idx < (T + 8) * (T + 8); idx += 256) {
        const int yy = idx / (T + 8), xx = idx - yy * (T + 8);
        const int gy = r0 - 4 + yy, gx = c0 - 4 + xx;
        s0[idx] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? m[(size_t)gy * a.W + gx] : 0.0f;
    }
    __syncthreads();
    m2w_layer<1, 2, T + 8, T + 6, false>(s0, s1, a.w1, a.b1, r0 - 3, c0 - 3, a.H, a.W, nullptr, 0, tid);
    __syncthreads();
    m2w_layer<2, 4, T + 6, T + 4, false>(s1, s2, a.w2, a.b2, r0 - 2, c0 - 2, a.H, a.W, nullptr, 0, tid);
    __syncthreads();
    m2w_layer<4, 4, T + 4, T + 2, false>(s2, s3, a.w3, a.b3, r0 - 1, c0 - 1, a.H, a.W, nullptr, 0, tid);
    __syncthreads();
    m2w_layer<4, 2, T + 2, T, true>(s3, nullptr, a.w4, a.b4, r0, c0, a.H, a.W, a.gates + (size_t)b * 2 * plane, plane, tid);
}

}  // namespace

// gates [B, 2, H, W] = softmax_c(conv3x3_4(relu(conv3x3_3(relu(conv3x3_2(relu(conv3x3_1(mask))))))))  -- mask2weights_EN with Kw = 2
// (MASIC.py:1411-1434); w_i / b_i: the four Conv2d weights [Cout][Cin][3][3] (1 -> 2 -> 4 -> 4 -> 2) and biases, float32 on the device.
extern "C" int masic_mask2weights_en_fwd(const float* mask, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                         const float* b3, const float* w4, const float* b4, float* gates, int B, int H, int W, void* stream) {
    MASIC_REQUIRE(mask && gates && w1 && b1 && w2 && b2 && w3 && b3 && w4 && b4, MASIC_ERR_ARG, "mask2weights_en_fwd: null pointer");
    MASIC_REQUIRE(B > 0 && H > 0 && W > 0, MASIC_ERR_SHAPE, "mask2weights_en_fwd: bad shape");
    const int tiles_w = ceil_div(W, T), tiles_h = ceil_div(H, T);
    M2wArgs a{mask, gates, w1, b1, w2, b2, w3, b3, w4, b4, H, W, tiles_w};
    hipLaunchKernelGGL(m2w_en_kernel, dim3(tiles_w * tiles_h, B), dim3(256), 0, (hipStream_t)stream, a);
    return masic_launch_status("mask2weights_en_fwd");
}
