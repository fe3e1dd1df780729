// skinny.hip -- the entropy-parameter layers of HSIC.compress / decompress evaluated on a SHORT LIST OF PIXELS (SURVEY.md 8(f)-1;
// reference MASIC.py:986-1044, :1262-1296: the masked 5x5 context convolution on a crop and the nine 1x1 head layers, per symbol).
//
// The decoder's loop advances one coding wavefront (at most one pixel per latent row: <= 32 pixels of a 32 x 32 latent) per step, and
// only those pixels' entropy parameters are new.  Round 2 re-ran the full-latent kernels every step: ~140 us of GEMMs over 1 024 pixels
// to use 32 of them.  These kernels compute one 32-pixel MFMA column block:
//   * workgroup = one 32-channel output tile x one 32-pixel tile, NW waves splitting the K loop NW ways (k-step s goes to wave s % NW):
//     the whole job is weight streaming (16 MB of head weights per step), so the point is to have MANY short dependent chains in flight --
//     108 ... 324 workgroups x NW waves, every wave's loads issued eight k-steps at a time, two batches deep;
//   * both MFMA operands come STRAIGHT FROM GLOBAL MEMORY: the weight fragment of (co tile, k-step) is 2 x 512 contiguous bytes of the
//     GEMM pack the full-size kernels use ([cb][k16][hh][co 128][8]) or 1 KiB of the context model's fragment pack (below); the
//     activation fragment is the lane's own pixel record half (F16K: 16 bytes at ((k16 * HW + pixel) * 32 + 16 h)) -- for the context
//     model the record of the pixel shifted by the tap, or an out-of-range offset (reads as zeros) outside the latent;
//   * the NW partial accumulators are summed through LDS in wave order (deterministic), then bias, activation, optional per-pixel gate,
//     and a scatter of the 32 pixels' records (F16K) or planes (float32 NCHW) into FULL-SIZE buffers, so that the next layer, the table
//     kernel (masic_gmm_cdf_rows) and the encoder's one-pass form (the same kernels over all pixels, 32 per workgroup) address pixels
//     the same way.  A pixel's result depends on that pixel's operands and the fixed k order only: encoder (all pixels at once) and
//     decoder (a wavefront at a time) get bit-identical parameters, which is what makes their coding tables agree.
// The pixel list of a launch is pix[(*step) * list_stride + i], i < npix (step: device int or null) -- the decoder replays one captured
// graph per coding step and the step counter lives on the device (codec.hip: masic_rans_decode_step).
#include "common.h"

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

struct SkGroup {
    const unsigned short* x;      // F16K [Cin16][HW][16] of one image
    const unsigned short* w;      // GEMM pack [cb][nk16][hh][co 128][8]; context model: [co tile][step][lane][8] (pack_ctx_skinny_kernel)
    const float* bias;            // [Cout] or null
    unsigned short* y16;          // F16K [out_ctot / 16][HW][16] or null
    float* y32;                   // float32 [out_ctot][HW] or null
    int Cin16, nsteps, Cout, out_ctot, out_coff, act;
};
constexpr int SK_MAXG = 3;
struct SkArgs {
    SkGroup g[SK_MAXG];
    int tile_end[SK_MAXG];        // first co tile (blockIdx.x) past group i
    int n;
    const int* pix;
    const int* step;
    int list_stride, npix, HW, h, w;
    const float* gate;            // [gate_ctot][HW] float32 or null: multiplied after the activation (mask2weights gates of the right view)
    int gate_c;
};

constexpr int SK_CH = 8;          // k-steps per load batch and wave
constexpr unsigned SK_OOR = 0xC0000000u;

// NW waves per workgroup (K split); CTX: the masked 5x5 context model (12 live taps) instead of a 1x1 layer
template <int NW, bool CTX>
__global__ __launch_bounds__(NW * 64) void skinny_f16k(const SkArgs a) {
    __shared__ float red[NW][16][64];
    SkGroup g = a.g[0];
    int t0 = 0;
    if (a.n > 1 && (int)blockIdx.x >= a.tile_end[0]) { g = a.g[1]; t0 = a.tile_end[0]; }
    if (a.n > 2 && (int)blockIdx.x >= a.tile_end[1]) { g = a.g[2]; t0 = a.tile_end[1]; }
    const int ct = blockIdx.x - t0;                                   // 32-channel output tile inside the group
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int* pix = a.pix + (a.step != nullptr ? (size_t)(*a.step) * a.list_stride : 0);
    const int pi = blockIdx.y * 32 + j;
    const int p = pi < a.npix ? pix[pi] : -1;
    const int nsteps = g.nsteps;
    size_t wbytes;
    if (CTX) wbytes = (size_t)((g.Cout + 31) / 32) * nsteps * 1024;
    else wbytes = (size_t)((g.Cout + 127) / 128) * nsteps * 4096;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, (int)wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)g.x, 0, g.Cin16 * a.HW * 32, 0x00020000);
    const unsigned wlane = CTX ? (unsigned)(ct * nsteps) * 1024u + lane * 16u
                               : (unsigned)((ct >> 2) * nsteps) * 4096u + h * 2048u + ((ct & 3) * 32 + j) * 16u;
    const unsigned wstep = CTX ? 1024u : 4096u;
    const int pr = p >= 0 ? p / a.w : 0, pc = p >= 0 ? p - pr * a.w : 0;
    auto boff = [&](int s) -> unsigned {                              // the lane's activation operand of k-step s
        if (p < 0 || s >= nsteps) return SK_OOR;
        if (!CTX) return s < g.Cin16 ? (unsigned)(s * a.HW + p) * 32u + 16u * h : SK_OOR;
        const int tap = s / g.Cin16, k16 = s - tap * g.Cin16;         // live taps of the type-A 5x5 mask: rows -2, -1 (5 columns each), row 0 columns -2, -1
        const int dy = tap < 10 ? tap / 5 - 2 : 0, dx = tap < 10 ? tap % 5 - 2 : tap - 12;
        const int r = pr + dy, c = pc + dx;
        return (r >= 0 && c >= 0 && c < a.w) ? (unsigned)(k16 * a.HW + r * a.w + c) * 32u + 16u * h : SK_OOR;
    };
    v4u A[2][SK_CH], Bq[2][SK_CH];
    auto load = [&](auto PB, int base) {
        constexpr int pb = decltype(PB)::value;
#pragma unroll
        for (int i = 0; i < SK_CH; ++i) {
            const int s = wv + NW * (base + i);
            A[pb][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, s < nsteps ? wlane + (unsigned)s * wstep : SK_OOR, 0, 0);
            Bq[pb][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, boff(s), 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    auto mma = [&](auto PB) {
        constexpr int pb = decltype(PB)::value;
#pragma unroll
        for (int i = 0; i < SK_CH; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[pb][i]), __builtin_bit_cast(bf16x8, Bq[pb][i]), acc, 0, 0, 0);
    };
    const int nper = (nsteps + NW - 1) / NW;                           // k-steps of this wave (the tail reads zeros)
    load(std::integral_constant<int, 0>{}, 0);
    int base = 0;
    for (; base + SK_CH < nper; base += 2 * SK_CH) {
        load(std::integral_constant<int, 1>{}, base + SK_CH);
        mma(std::integral_constant<int, 0>{});
        if (base + 2 * SK_CH < nper) load(std::integral_constant<int, 0>{}, base + 2 * SK_CH);
        mma(std::integral_constant<int, 1>{});
    }
    if (base < nper) mma(std::integral_constant<int, 0>{});
    // ---- reduction over the K split, in wave order
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wv][e][lane] = acc[e];
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float s = red[0][e][lane];
#pragma unroll
        for (int k = 1; k < NW; ++k) s += red[k][e][lane];
        acc[e] = s;
    }
    const int c0 = ct * 32;                                            // first channel of this tile; lane (j, h) holds channels c0 + 8 q + 4 h + i in acc[4 q + i]
    if (g.bias != nullptr) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += g.bias[c0 + 4 * h + (e & 3) + 8 * (e >> 2)];
    }
    float gv = 1.0f;
    if (a.gate != nullptr && p >= 0) gv = a.gate[(size_t)a.gate_c * a.HW + p];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = apply_act(acc[e], g.act) * gv;
    if (g.y16 != nullptr) {
        // F16K: the two halves of the wave swap quarters so that lane (j, h) holds channels 8 h .. 8 h + 7 of each 16-channel record
        unsigned short* rec = g.y16 + (((size_t)((g.out_coff + c0) >> 4)) * a.HW + (p >= 0 ? p : 0)) * 16 + 8 * h;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned a0 = pack2bf(acc[8 * r + 0], acc[8 * r + 1]), a1 = pack2bf(acc[8 * r + 2], acc[8 * r + 3]);
            const unsigned b0 = pack2bf(acc[8 * r + 4], acc[8 * r + 5]), b1 = pack2bf(acc[8 * r + 6], acc[8 * r + 7]);
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            uint4 st;
            st.x = s0[0]; st.y = s1[0]; st.z = s0[1]; st.w = s1[1];
            if (p >= 0) *reinterpret_cast<uint4*>(rec + (size_t)r * a.HW * 16) = st;
        }
    } else if (p >= 0) {
        float* yb = g.y32 + (size_t)(g.out_coff + c0 + 4 * h) * a.HW + p;
#pragma unroll
        for (int e = 0; e < 16; ++e) yb[(size_t)((e & 3) + 8 * (e >> 2)) * a.HW] = acc[e];
    }
}

// context-model weights [Cout][Cin][5][5] (masked taps already zero) -> bf16 fragments [co tile][step = tap * Cin16 + k16][lane (j, h)][8]:
// element e of lane (j, h) = W[32 tile + j][16 k16 + 8 h + e][kh][kw], (kh, kw) = the live tap's position
__global__ void pack_ctx_skinny_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cin, int Cout, int Cin16) {
    const int nsteps = 12 * Cin16;
    const size_t total = (size_t)((Cout + 31) / 32) * nsteps * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = (int)(r & 7); r >>= 3;
        const int lane = (int)(r & 63); r >>= 6;
        const int s = (int)(r % nsteps);
        const int tile = (int)(r / nsteps);
        const int j = lane & 31, h = lane >> 5, tap = s / Cin16, k16 = s - tap * Cin16;
        const int kh = tap < 10 ? tap / 5 : 2, kw = tap < 10 ? tap % 5 : tap - 10;
        const int co = tile * 32 + j, ci = k16 * 16 + 8 * h + e;
        const float v = (co < Cout && ci < Cin) ? w[(((size_t)co * Cin + ci) * 5 + kh) * 5 + kw] : 0.0f;
        const __bf16 b = (__bf16)v;
        out[i] = __builtin_bit_cast(unsigned short, b);
    }
}

}  // namespace

extern "C" size_t masic_skinny_ctx_packed_bytes(int Cin, int Cout) { return (size_t)ceil_div(Cout, 32) * 12 * ceil_div(Cin, 16) * 1024; }

extern "C" int masic_skinny_ctx_pack_weight(const float* w, void* w_packed, int Cin, int Cout, void* stream) {
    MASIC_REQUIRE(w && w_packed && Cin > 0 && Cout > 0, MASIC_ERR_ARG, "skinny_ctx_pack_weight: bad argument");
    const size_t total = masic_skinny_ctx_packed_bytes(Cin, Cout) / 2;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_ctx_skinny_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)w_packed, Cin, Cout, ceil_div(Cin, 16));
    return masic_launch_status("skinny_ctx_pack_weight");
}

// Up to three 1x1 layers (w_packed from masic_gemm_f16k_pack_weight) or ONE 5x5 type-A masked convolution (ctx != 0; w_packed from
// masic_skinny_ctx_pack_weight, latent h x w) on the pixels pix[(*step) * list_stride + i], i < npix, of one image.  x / y_f16k / y_nchw are
// FULL-SIZE buffers (HW pixels); only the listed pixels are read (plus, for the masked convolution, their causal neighbours) and written.
extern "C" int masic_skinny_group_fwd(const masic_gemm_group_t* groups, int ngroups, int ctx, const int32_t* pix, const int32_t* step, int list_stride,
                                      int npix, int h, int w, const float* gate, int gate_c, void* stream) {
    MASIC_REQUIRE(groups && pix && ngroups >= 1 && ngroups <= SK_MAXG && (!ctx || ngroups == 1), MASIC_ERR_ARG, "skinny_group_fwd: bad argument");
    MASIC_REQUIRE(npix >= 1 && h >= 1 && w >= 1, MASIC_ERR_SHAPE, "skinny_group_fwd: bad shape");
    SkArgs a{};
    int tiles = 0;
    for (int i = 0; i < ngroups; ++i) {
        const masic_gemm_group_t& g = groups[i];
        MASIC_REQUIRE(g.x && g.w_packed && g.wscale == nullptr && g.y_f8k == nullptr && ((g.y_f16k != nullptr) != (g.y_nchw != nullptr)), MASIC_ERR_ARG,
                      "skinny_group_fwd: bf16 operands, exactly one of an F16K and a float32 output");
        MASIC_REQUIRE(g.Cin > 0 && g.Cin % 16 == 0 && g.Cout > 0 && g.Cout % 32 == 0, MASIC_ERR_UNSUPPORTED, "skinny_group_fwd: needs Cin %% 16 == 0 and Cout %% 32 == 0");
        MASIC_REQUIRE(g.out_coff >= 0 && g.out_coff + g.Cout <= g.out_ctot && (g.y_nchw != nullptr || (g.out_ctot % 16 == 0 && g.out_coff % 16 == 0)), MASIC_ERR_SHAPE,
                      "skinny_group_fwd: output channel view");
        MASIC_REQUIRE((long)(g.Cin / 16) * h * w * 32 < (1l << 30), MASIC_ERR_UNSUPPORTED, "skinny_group_fwd: activation plane too large for 32-bit offsets");
        const int cin16 = g.Cin / 16;
        a.g[i] = SkGroup{(const unsigned short*)g.x, (const unsigned short*)g.w_packed, g.bias, (unsigned short*)g.y_f16k, g.y_nchw, cin16,
                         ctx ? 12 * cin16 : round_up(cin16, 4), g.Cout, g.out_ctot, g.out_coff, g.act};
        tiles += g.Cout / 32;
        a.tile_end[i] = tiles;
    }
    a.n = ngroups;
    a.pix = pix; a.step = step; a.list_stride = list_stride; a.npix = npix; a.HW = h * w; a.h = h; a.w = w;
    a.gate = gate; a.gate_c = gate_c;
    const dim3 grid(tiles, ceil_div(npix, 32));
    if (ctx) hipLaunchKernelGGL((skinny_f16k<8, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((skinny_f16k<4, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return masic_launch_status("skinny_group_fwd");
}
