// How fast does this device take the OUTPUT of an F16K-writing kernel, by store shape and by how the stores are paced?
// The first analysis layer (conv_a_gdn_f16k) writes 134 MB per launch at 8 x 512 x 512 and ran at 60 us with its stores, 15 us without
// them (timing-only ablation) -- 3 TB/s of stores against the guide's 6.0-6.2 TB/s for plain streaming stores.  This probe replays the
// kernel's exact address stream (tile = 8 rows x 32 pixels of a 256 x 256 x 128-channel F16K tensor per workgroup iteration, wave = row,
// 8 stores of 1 KiB per wave and tile) with nothing but the stores, in variants:
//   shape 0: lane (j, h) -> record j, half h      (what store_f16k_tile does: consecutive lanes 32 bytes apart, half-waves interleave)
//   shape 1: lane l -> bytes [16 l, 16 l + 16)    (lane-linear 1 KiB)
//   tile  0: 8 rows x 32 px per workgroup-iteration;  1: 1 row x 256 px (8 KiB runs per channel block)
//   pace  0: free running;  1: s_waitcnt vmcnt(0) + s_barrier after every tile (the kernel's loop);  2: vmcnt(0) only
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_shape.hip -o tools/micro/store_shape && tools/micro/store_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int SHAPE, int TILE, int PACE>
__global__ __launch_bounds__(512, 1) void store_k(unsigned short* __restrict__ y, int B, int Ho, int Wo, int spin) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const size_t oplane = (size_t)Ho * Wo;
    const int tiles_w = TILE == 0 ? Wo / 32 : Wo / 256, tiles_per_img = TILE == 0 ? tiles_w * (Ho / 8) : tiles_w * Ho, ntiles = tiles_per_img * B;
    v4u v = {(unsigned)tid, (unsigned)blockIdx.x, 0x3f803f80u, 0x40004000u};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t = tile - b * tiles_per_img;
        int oh, ow0;
        if (TILE == 0) { oh = (t / tiles_w) * 8 + wave; ow0 = (t % tiles_w) * 32; }
        else { oh = t / tiles_w; ow0 = (t % tiles_w) * 256 + wave * 32; }
        // busy work standing in for the tile's MFMAs (spin iterations of dependent VALU)
        for (int s = 0; s < spin; ++s) v.x = v.x * 1664525u + 1013904223u;
        unsigned char* base = (unsigned char*)y + (((size_t)b * 8) * oplane + (size_t)oh * Wo + ow0) * 32;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            unsigned char* p = base + (size_t)blk * oplane * 32 + (SHAPE == 0 ? j * 32 + h * 16 : lane * 16);
            if (PACE != 3 || v.x == 0x12345u) *reinterpret_cast<v4u*>(p) = v;      // PACE 3: the spin loop alone (the store never executes)
        }
        if (PACE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (PACE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int SHAPE, int TILE, int PACE>
void run(unsigned short* y, int B, int Ho, int Wo, int spin) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_k<SHAPE, TILE, PACE>), dim3(256), dim3(512), 0, 0, y, B, Ho, Wo, spin);
    hipEventRecord(e0);
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((store_k<SHAPE, TILE, PACE>), dim3(256), dim3(512), 0, 0, y, B, Ho, Wo, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)B * 128 * Ho * Wo * 2;
    printf("shape %d tile %d pace %d spin %5d: %7.1f us per launch, %.2f TB/s\n", SHAPE, TILE, PACE, spin, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
}

int main() {
    const int B = 8, Ho = 256, Wo = 256;
    unsigned short* y;
    hipMalloc(&y, (size_t)B * 128 * Ho * Wo * 2);
    for (int spin : {0, 100, 200, 400}) {
        run<0, 0, 0>(y, B, Ho, Wo, spin); run<0, 0, 1>(y, B, Ho, Wo, spin); run<0, 0, 2>(y, B, Ho, Wo, spin);
        run<1, 0, 0>(y, B, Ho, Wo, spin); run<1, 0, 1>(y, B, Ho, Wo, spin);
        run<0, 1, 0>(y, B, Ho, Wo, spin); run<0, 1, 1>(y, B, Ho, Wo, spin);
        run<1, 1, 0>(y, B, Ho, Wo, spin); run<1, 1, 1>(y, B, Ho, Wo, spin);
        run<0, 0, 3>(y, B, Ho, Wo, spin);
    }
    hipFree(y);
    return 0;
}
