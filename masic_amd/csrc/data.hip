// data.hip -- the per-item preparation of the reference's paired dataset on the device (SURVEY.md 8(f)-4;
// compressai/datasets/utils.py:207-285): the paired crop as float32 CHW / 255 (ToTensor), and the grey 128 x 128 patch the
// homography network reads -- cv2.resize of the crop to 256 x 256 (uint8, INTER_LINEAR), ToTensor, Normalize with the scalar
// MEAN / STD of utils.py:26-27, mean over the channels, patch at (x, y).  The reference does this per picture in DataLoader
// workers (cv2 on the host); here the decoded uint8 picture is uploaded once and both outputs come out of two launches.
// The resize follows OpenCV's uint8 algorithm bit for bit (resize.cpp: pixel-centre mapping, 11-bit fixed-point coefficients,
// int32 horizontal pass, (b*(t>>4))>>16 vertical pass, +2 >>2; an exact 2x decimation runs as the rounded 2x2 box mean), and
// only the patch window of the resized picture is ever computed.  cv2 itself is absent: parity is pinned by the numpy
// restatement oracle/data_oracle.py.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void crop_to_tensor_kernel(const unsigned char* __restrict__ img, int W, int h0, int w0, int ph, int pw,
                                                             float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ph * pw) return;
    const int r = i / pw, c = i - r * pw;
    const unsigned char* p = img + ((size_t)(h0 + r) * W + (w0 + c)) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out[(size_t)ch * ph * pw + i] = (float)p[ch] / 255.0f;
}

__device__ __forceinline__ void lin_coeff(int d, double scale, int src, int& s, int& c0, int& c1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    if (s >= src - 1) { s = src - 1; f = 0.0f; }
    c0 = (int)rintf((1.0f - f) * 2048.0f);
    c1 = (int)rintf(f * 2048.0f);
}

__global__ __launch_bounds__(256) void homo_patch_kernel(const unsigned char* __restrict__ img, int W, int h0, int w0, int ph, int pw,
                                                         int S, int px0, int py0, int patch, float mean, float stdv,
                                                         float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= patch * patch) return;
    const int dy = py0 + i / patch, dx = px0 + i % patch;              // pixel of the S x S resized crop
    int v[3];
    if (ph == 2 * S && pw == 2 * S) {
        const unsigned char* p = img + ((size_t)(h0 + 2 * dy) * W + (w0 + 2 * dx)) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) v[ch] = (p[ch] + p[3 + ch] + p[(size_t)W * 3 + ch] + p[(size_t)W * 3 + 3 + ch] + 2) >> 2;
    } else {
        int sx, a0, a1, sy, b0, b1;
        lin_coeff(dx, (double)pw / S, pw, sx, a0, a1);
        lin_coeff(dy, (double)ph / S, ph, sy, b0, b1);
        const int sx1 = min(sx + 1, pw - 1), sy1 = min(sy + 1, ph - 1);
        const unsigned char* r0 = img + ((size_t)(h0 + sy) * W + w0) * 3;
        const unsigned char* r1 = img + ((size_t)(h0 + sy1) * W + w0) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int t0 = r0[sx * 3 + ch] * a0 + r0[sx1 * 3 + ch] * a1;
            const int t1 = r1[sx * 3 + ch] * a0 + r1[sx1 * 3 + ch] * a1;
            const int o = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
            v[ch] = min(max(o, 0), 255);
        }
    }
    float n[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) n[ch] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v[ch], 255.0f), mean), stdv);
    out[i] = __fdiv_rn(__fadd_rn(__fadd_rn(n[0], n[1]), n[2]), 3.0f);
}

}  // namespace

extern "C" int masic_pair_prep(const uint8_t* img_hwc, int H, int W, int start_h, int start_w, int ph, int pw, float* pic_chw,
                               int homopic, int patch_x, int patch_y, int homopatch, float* homo_patch, void* stream) {
    MASIC_REQUIRE(img_hwc && (pic_chw || homo_patch), MASIC_ERR_ARG, "pair_prep: null pointer");
    MASIC_REQUIRE(ph > 0 && pw > 0 && start_h >= 0 && start_w >= 0 && start_h + ph <= H && start_w + pw <= W, MASIC_ERR_SHAPE,
                  "pair_prep: crop [%d:%d, %d:%d] of a %d x %d picture", start_h, start_h + ph, start_w, start_w + pw, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (pic_chw) hipLaunchKernelGGL(crop_to_tensor_kernel, dim3(ceil_div(ph * pw, 256)), dim3(256), 0, st, img_hwc, W, start_h, start_w, ph, pw, pic_chw);
    if (homo_patch) {
        MASIC_REQUIRE(homopic > 0 && homopatch > 0 && patch_x >= 0 && patch_y >= 0 && patch_x + homopatch <= homopic && patch_y + homopatch <= homopic,
                      MASIC_ERR_SHAPE, "pair_prep: patch [%d:%d, %d:%d] of the %d x %d resized crop", patch_y, patch_y + homopatch, patch_x,
                      patch_x + homopatch, homopic, homopic);
        const float mean = (0.485f + 0.456f + 0.406f) / 3.0f, stdv = (0.229f + 0.224f + 0.225f) / 3.0f;    // utils.py:26-27
        hipLaunchKernelGGL(homo_patch_kernel, dim3(ceil_div(homopatch * homopatch, 256)), dim3(256), 0, st, img_hwc, W, start_h, start_w, ph, pw,
                           homopic, patch_x, patch_y, homopatch, mean, stdv, homo_patch);
    }
    return masic_launch_status("pair_prep");
}
