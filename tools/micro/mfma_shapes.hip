// (1) What does s_memrealtime count on this device?  tools/f16k_stamps.py and mfma_peak.hip assume 100 MHz; a kernel's duration measured with
//     HIP events against the same kernel's s_memrealtime delta answers it.
// (2) v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 (lever (ii) of VERDICT round 2, item 5): sustained rate of a bare loop on
//     operands that change every instruction, same number of accumulator registers in flight, one and two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__global__ void tick_kernel(unsigned long long* st, int spins) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned v = threadIdx.x;
    for (int i = 0; i < spins; ++i) v = v * 1664525u + 1013904223u;
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { st[0] = r1 - r0; st[1] = c1 - c0; st[2] = v; }
}

// SHAPE 0: 32x32x16 (NACC accumulators of 16 registers); SHAPE 1: 16x16x32 (4 * NACC accumulators of 4 registers: the same register count,
// and the same FLOPs per outer iteration: one 32x32x16 = two 16x16x32)
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void loop(float* out, int iters, unsigned seed) {
    bf16x8 a[8], b[8];
    unsigned h = seed ^ (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            a[r][i] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 0.001f);
            h = h * 1664525u + 1013904223u;
            b[r][i] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 0.001f);
        }
    float s = 0.0f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[NACC];
        for (int k = 0; k < NACC; ++k)
            for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(r + k) & 7], b[(r + 3 * k) & 7], acc[k], 0, 0, 0);
        }
        for (int k = 0; k < NACC; ++k)
            for (int i = 0; i < 16; ++i) s += acc[k][i];
    } else {
        f32x4 acc[4 * NACC];
        for (int k = 0; k < 4 * NACC; ++k)
            for (int i = 0; i < 4; ++i) acc[k][i] = 0.0f;
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 4 * NACC; ++k) {
                    if ((k & 1) == (r & 1) || true)         // 2 of the 4 accumulators per 32x32-equivalent and iteration: same FLOPs as SHAPE 0
                        if (k % 2 == 0 || k % 2 == 1) {}
                }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 2 * NACC; ++k) {
                    const int kk = 2 * k + (r & 1);
                    acc[kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(r + k) & 7], b[(r + 3 * k) & 7], acc[kk], 0, 0, 0);
                }
        }
        for (int k = 0; k < 4 * NACC; ++k)
            for (int i = 0; i < 4; ++i) s += acc[k][i];
    }
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NACC>
static void run(int wgs_per_cu, int ncu, int iters, float* out) {
    const int grid = wgs_per_cu * ncu, reps = 10;
    hipLaunchKernelGGL((loop<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, iters / 10, 7u);
    hipDeviceSynchronize();
    hipEvent_t ev[reps + 1];
    for (int r = 0; r <= reps; ++r) hipEventCreate(&ev[r]);
    hipEventRecord(ev[0], 0);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL((loop<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 7u + r);
        hipEventRecord(ev[r + 1], 0);
    }
    hipEventSynchronize(ev[reps]);
    float best = 1e30f, sum = 0.0f;
    for (int r = 0; r < reps; ++r) {
        float ms; hipEventElapsedTime(&ms, ev[r], ev[r + 1]);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flops = (double)grid * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("{\"mfma\": \"%s\", \"accumulator_registers\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"ms_best\": %.3f, \"ms_avg\": %.3f, \"tflops_best\": %.1f, \"tflops_avg\": %.1f}\n",
           SHAPE == 0 ? "32x32x16_bf16" : "16x16x32_bf16", 16 * NACC, wgs_per_cu, iters, best, sum / reps, flops / best / 1e9, flops / (sum / reps) / 1e9);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    unsigned long long* st; hipMalloc(&st, 32);
    float* out; hipMalloc(&out, (size_t)ncu * 8 * 256 * sizeof(float));
    for (int spins : {200000, 2000000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, 0, st, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, 0, st, spins);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hs[3]; hipMemcpy(hs, st, 24, hipMemcpyDeviceToHost);
        printf("{\"kernel_ms_by_events\": %.3f, \"s_memrealtime_ticks\": %llu, \"s_memrealtime_mhz\": %.2f, \"s_memtime_ticks\": %llu, \"s_memtime_mhz\": %.1f}\n",
               ms, hs[0], hs[0] / (ms * 1e3), hs[1], hs[1] / (ms * 1e3));
    }
    for (int w = 1; w <= 2; ++w) {
        run<0, 4>(w, ncu, 20000, out);
        run<1, 4>(w, ncu, 20000, out);
        run<0, 8>(w, ncu, 20000, out);
        run<1, 8>(w, ncu, 20000, out);
    }
    return 0;
}
