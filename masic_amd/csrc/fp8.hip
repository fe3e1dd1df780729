// fp8.hip -- layout / calibration helpers of the fp8 (OCP e4m3fn) operand path (BASELINE configs[4]; gfx950 / MI355X).
//
// F8K = [B][C/32][H*W][32] fp8: a pixel's 32 channels are one 32-byte record -- the same record size as F16K
// ([B][C/16][H*W][16] bf16), so the DMA-staged kernels of conv_f16k.hip move both formats with the same code and one
// v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64) consumes two records per pixel.  A tensor x is stored as fp8(x / scale) with ONE
// scale per tensor, fixed at calibration (masic_amd/fp8.py: scale = recorded max|x| x margin / 448); weights carry one scale
// per output channel (conv_f16k.hip: wscale_*).  No reference counterpart: the reference computes in float32; what this
// path costs in rate / PSNR / symbols is declared and gated against the oracle in tests/test_gpu_fp8.py.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned pack4(float a, float b, float c, float d, float inv) {
    a = __builtin_amdgcn_fmed3f(a * inv, -448.0f, 448.0f);
    b = __builtin_amdgcn_fmed3f(b * inv, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c * inv, -448.0f, 448.0f);
    d = __builtin_amdgcn_fmed3f(d * inv, -448.0f, 448.0f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

// float32 NCHW (channel view) -> F8K: one thread per (pixel, 16-channel half record): 16 coalesced plane reads, one 16-byte write
__global__ __launch_bounds__(256) void nchw_to_f8k_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int C, int Cpad, int HW,
                                                          int ctot, int coff, int op, float inv) {
    const int b = blockIdx.z, c16 = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* xb = x + ((size_t)b * ctot + coff) * HW + p;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c16 * 16 + i;
        const float t = xb[(size_t)(c < C ? c : C - 1) * HW];
        v[i] = c < C ? apply_inop(t, op) : 0.0f;
    }
    uint4 q;
    q.x = pack4(v[0], v[1], v[2], v[3], inv); q.y = pack4(v[4], v[5], v[6], v[7], inv);
    q.z = pack4(v[8], v[9], v[10], v[11], inv); q.w = pack4(v[12], v[13], v[14], v[15], inv);
    *reinterpret_cast<uint4*>(y + (((size_t)b * (Cpad >> 5) + (c16 >> 1)) * HW + p) * 32 + (c16 & 1) * 16) = q;
}

// max |x| of a float32 or bf16 buffer (calibration): non-negative floats order like their bit patterns, so one atomicMax
// on the unsigned view per workgroup.  `out` must be zeroed by the caller.
template <bool BF16>
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ x, size_t n, unsigned* __restrict__ out) {
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v;
        if (BF16) v = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(x)[i] << 16);
        else v = reinterpret_cast<const float*>(x)[i];
        v = fabsf(v);
        m = v > m ? v : m;           // NaN never wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

}  // namespace

extern "C" int masic_nchw_to_f8k(const float* x, void* y, int B, int C, int HW, int ctot, int coff, int in_op, float inv_scale, void* stream) {
    MASIC_REQUIRE(x && y && inv_scale > 0.0f, MASIC_ERR_ARG, "nchw_to_f8k: null pointer or non-positive scale");
    MASIC_REQUIRE(B > 0 && C > 0 && HW > 0 && coff >= 0 && coff + C <= ctot, MASIC_ERR_SHAPE, "nchw_to_f8k: view out of range");
    const int Cpad = round_up(C, 32);
    hipLaunchKernelGGL(nchw_to_f8k_kernel, dim3(ceil_div(HW, 256), Cpad / 16, B), dim3(256), 0, (hipStream_t)stream,
                       x, (unsigned char*)y, C, Cpad, HW, ctot, coff, in_op, inv_scale);
    return masic_launch_status("nchw_to_f8k");
}

// *out (a device float, zeroed by the caller before the first call) = max(*out, max |x|); x: n float32 (bf16 = 0) or bf16 values
extern "C" int masic_absmax(const void* x, size_t n, int bf16, float* out, void* stream) {
    MASIC_REQUIRE(x && out, MASIC_ERR_ARG, "absmax: null pointer");
    if (n == 0) return MASIC_OK;
    int nb = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (nb > 2048) nb = 2048;
    if (bf16) hipLaunchKernelGGL(absmax_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned*)out);
    else hipLaunchKernelGGL(absmax_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned*)out);
    return masic_launch_status("absmax");
}
