// Sustained rate of v_mfma_f32_32x32x16_bf16 on this device with nothing else in the loop: the ceiling any MFMA-bound kernel
// of this library can be compared with besides the datasheet peak (MI355X_MICROARCH.md: 2.5 PFLOP/s dense bf16 at 2.4 GHz).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak && tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed - i * 0.5f); }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the loop alive
}

// the same loop on pseudo-random operands that change with every instruction (8 A and 8 B fragments cycled): the multiplier array
// toggles as it does on real activations, which is what the power limit sees
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_random(float* out, int iters, unsigned seed, unsigned long long* stamps) {
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
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
    const bool st = stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    if (st) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[1] = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(r + k) & 7], b[(r + 3 * k) & 7], acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (st) { stamps[2] = __builtin_amdgcn_s_memtime(); stamps[3] = __builtin_amdgcn_s_memrealtime(); }
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run_random(int wgs_per_cu, int ncu, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = wgs_per_cu * ncu;
    unsigned long long* stamps; hipMalloc(&stamps, 32);
    hipLaunchKernelGGL(mfma_loop_random<NACC>, dim3(grid), dim3(256), 0, 0, out, iters / 10, 7u, stamps);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.0f;
    const int reps = 20;                      // back to back, as the layers of a forward are
    hipEvent_t ev[reps + 1];
    for (int r = 0; r <= reps; ++r) hipEventCreate(&ev[r]);
    hipEventRecord(ev[0], 0);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(mfma_loop_random<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 7u + r, stamps);
        hipEventRecord(ev[r + 1], 0);
    }
    hipEventSynchronize(ev[reps]);
    for (int r = 0; r < reps; ++r) {
        float ms; hipEventElapsedTime(&ms, ev[r], ev[r + 1]);
        best = ms < best ? ms : best; sum += ms;
    }
    unsigned long long hs[4];
    hipMemcpy(hs, stamps, 32, hipMemcpyDeviceToHost);
    const double mhz = (double)(hs[2] - hs[0]) / (double)(hs[3] - hs[1]) * 100.0;      // s_memtime: core clock; s_memrealtime: 100 MHz
    const double clk_per_mfma = (double)(hs[2] - hs[0]) / ((double)iters * NACC * wgs_per_cu);
    const double flops = (double)grid * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("{\"shader_clock_mhz_in_loop\": %.0f, \"core_clk_per_mfma_per_simd\": %.1f, ", mhz, clk_per_mfma);
    printf("\"operands\": \"random, changing every instruction\", \"accumulators\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"ms_best\": %.3f, "
           "\"ms_avg\": %.3f, \"tflops_best\": %.1f, \"tflops_avg\": %.1f}\n", NACC, wgs_per_cu, iters, best, sum / reps, flops / best / 1e9,
           flops / (sum / reps) / 1e9);
}

template <int NACC>
static void run(int wgs_per_cu, int ncu, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = wgs_per_cu * ncu;
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters / 10, 1.0f);       // warm-up
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.0f;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flops = (double)grid * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("{\"accumulators\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"ms_best\": %.3f, \"ms_avg\": %.3f, \"tflops_best\": %.1f, \"tflops_avg\": %.1f, "
           "\"mfma_issue_clk_at_2.4GHz\": %.1f}\n", NACC, wgs_per_cu, iters, best, sum / reps, flops / best / 1e9, flops / (sum / reps) / 1e9,
           best * 1e-3 * 2.4e9 / ((double)iters * NACC * wgs_per_cu));
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out; hipMalloc(&out, (size_t)ncu * 8 * 256 * sizeof(float));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.name, ncu, p.clockRate / 1000);
    for (int w = 1; w <= 2; ++w) {
        run<1>(w, ncu, iters, out);
        run<2>(w, ncu, iters, out);
        run<4>(w, ncu, iters, out);
        run<8>(w, ncu, iters, out);
    }
    // a long run: the clock the chip settles to once the power limit acts
    run<4>(2, ncu, iters * 20, out);
    for (int rep = 0; rep < 2; ++rep)
        for (int it : {800, 4000, 20000, 100000}) {
            run_random<4>(1, ncu, it, out);
            run_random<4>(2, ncu, it, out);
        }
    return 0;
}
