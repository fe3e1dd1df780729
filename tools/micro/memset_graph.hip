// memset_graph.hip -- do hipMemsetAsync nodes keep their place in a captured single-stream graph on this ROCm?
//
// Background (masic_amd/graph.py: GraphedTrainStep): in the replayed graph of the training step, weight-gradient
// workspaces that were zeroed with hipMemsetAsync / torch.zeros read back with garbage, different tensors from run to run; zero fills as
// kernel nodes cured it.  This program is the pattern without torch: on ONE capturing stream, per round i
//     memset(ws, 0)  ->  accumulate(ws += i + 1, four atomics per element)  ->  consume(out[i] = ws; ws = dirty)
// with `rounds` rounds per graph, several buffer sizes, replayed `replays` times.  Every out[i][e] must be 4 (i + 1).  Variants:
//   A  memset nodes from hipMemsetAsync under capture (what torch.zeros becomes)
//   B  the same with a zero-fill KERNEL instead (the shipped workaround)
//   C  as A, but the buffers come from hipMallocAsync inside the capture (torch's private pool allocates outside: cudaMalloc'd blocks)
//   D  as A with a second, forked stream doing independent work between memset and accumulate (cross-stream edges around the memset node)
//   E  as C (hipMallocAsync inside the capture) with the fill KERNEL
//   F  as A, but the buffer is hipMalloc'ed WHILE the capture is running (what torch's caching allocator does for its private pool),
//      a new buffer every round
//   G  as F with the fill KERNEL
//   H  as F the way torch does the rest: non-blocking streams, capture mode Global (Relaxed around the allocation),
//      hipGraphInstantiateWithFlags(AutoFreeOnLaunch), launched on ANOTHER stream than the captured one
//   I  as H with the fill KERNEL
//   J  as A, but the graph does not START with the memset node: a kernel that dirties the workspace is captured first
//      (the shape of a captured training step: the first zero fill comes after the first kernels)
//   K  as J with the fill KERNEL
//   L  as J, launched on the NULL stream (torch's default stream is the null stream: graph.replay() of a script that never
//      changes streams launches there)
//   M  as L with the fill KERNEL
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/memset_graph.hip -o /tmp/memset_graph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void accumulate(float* ws, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        for (int k = 0; k < 4; ++k) atomicAdd(ws + i, v);
}
__global__ void consume(float* ws, float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        out[i] = ws[i];
        ws[i] = 12345.0f;          // leaves the workspace dirty: only a fill that ran in its place gives the next round a clean one
    }
}
__global__ void zero_kernel(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0f;
}
__global__ void busy(float* p, size_t n, int it) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = p[i];
        for (int k = 0; k < it; ++k) v = v * 1.0001f + 0.5f;
        p[i] = v;
    }
}

static long run(char variant, size_t n, int rounds, int replays, size_t off = 0) {
    const bool torchlike = variant == 'H' || variant == 'I';
    hipStream_t st, side, launch;
    if (torchlike) {
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&launch, hipStreamNonBlocking));
    } else {
        CK(hipStreamCreate(&st));
        CK(hipStreamCreate(&side));
        launch = (variant == 'L' || variant == 'M') ? (hipStream_t) nullptr : st;
    }
    float *ws = nullptr, *out = nullptr, *other = nullptr;
    float* ws_base = nullptr;
    const bool graph_mem = variant == 'C' || variant == 'E', late = variant == 'F' || variant == 'G' || torchlike, kfill = variant == 'B' || variant == 'E' || variant == 'G' || variant == 'I' || variant == 'K' || variant == 'M';
    std::vector<float*> late_bufs;
    if (!graph_mem && !late) { CK(hipMalloc(&ws_base, (n + off) * 4)); ws = ws_base + off; }
    CK(hipMalloc(&out, (size_t)rounds * n * 4));
    CK(hipMalloc(&other, 1 << 22));
    CK(hipMemset(other, 0, 1 << 22));
    const unsigned nb = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipStreamBeginCapture(st, torchlike ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
    if (graph_mem) CK(hipMallocAsync((void**)&ws, n * 4, st));
    if (variant == 'J' || variant == 'K' || variant == 'L' || variant == 'M') hipLaunchKernelGGL(consume, dim3(nb), dim3(256), 0, st, ws, other, n < ((size_t)1 << 20) ? n : ((size_t)1 << 20));
    for (int i = 0; i < rounds; ++i) {
        if (late) {
            hipStreamCaptureMode m = hipStreamCaptureModeRelaxed;
            CK(hipThreadExchangeStreamCaptureMode(&m));
            CK(hipMalloc(&ws, n * 4));
            CK(hipThreadExchangeStreamCaptureMode(&m));
            late_bufs.push_back(ws);
        }
        if (kfill) hipLaunchKernelGGL(zero_kernel, dim3(nb), dim3(256), 0, st, ws, n);
        else CK(hipMemsetAsync(ws, 0, n * 4, st));
        if (variant == 'D') {
            CK(hipEventRecord(fork, st));
            CK(hipStreamWaitEvent(side, fork, 0));
            hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, side, other, (size_t)1 << 20, 64);
            CK(hipEventRecord(join, side));
        }
        hipLaunchKernelGGL(accumulate, dim3(nb), dim3(256), 0, st, ws, n, (float)(i + 1));
        if (variant == 'D') CK(hipStreamWaitEvent(st, join, 0));
        hipLaunchKernelGGL(consume, dim3(nb), dim3(256), 0, st, ws, out + (size_t)i * n, n);
    }
    if (graph_mem) CK(hipFreeAsync(ws, st));
    CK(hipStreamEndCapture(st, &graph));
    if (torchlike) CK(hipGraphInstantiateWithFlags(&exec, graph, hipGraphInstantiateFlagAutoFreeOnLaunch));
    else CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    std::vector<float> h((size_t)rounds * n);
    long bad = 0;
    long bad_round[4] = {0, 0, 0, 0}, bad_first_replay = 0;
    for (int r = 0; r < replays; ++r) {
        CK(hipMemsetAsync(out, 0xff, (size_t)rounds * n * 4, launch));
        CK(hipGraphLaunch(exec, launch));
        CK(hipMemcpyAsync(h.data(), out, (size_t)rounds * n * 4, hipMemcpyDeviceToHost, launch));
        CK(hipStreamSynchronize(launch));
        for (int i = 0; i < rounds; ++i)
            for (size_t e = 0; e < n; ++e)
                if (h[(size_t)i * n + e] != 4.0f * (i + 1)) { ++bad; ++bad_round[i < 3 ? i : 3]; if (r == 0) ++bad_first_replay; }
    }
    if (bad) printf("    wrong elements by round: round 0 %ld, round 1 %ld, round 2 %ld, later %ld; in the first launch %ld\n", bad_round[0], bad_round[1], bad_round[2], bad_round[3], bad_first_replay);
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
    if (!graph_mem && !late) CK(hipFree(ws_base));
    for (float* p : late_bufs) CK(hipFree(p));
    CK(hipFree(out));
    CK(hipFree(other));
    CK(hipStreamDestroy(st));
    CK(hipStreamDestroy(side));
    if (torchlike) CK(hipStreamDestroy(launch));
    return bad;
}

int main() {
    const size_t sizes[] = {64, 4096, 9600, 409600, 3686400};       // floats: bias sums ... the 5x5 128 -> 128 weight gradient and its split-K partials
    const char variants[] = {'A', 'B', 'D', 'F', 'G', 'H', 'I', 'J', 'K', 'L', 'M'};      // (C / E: hipMallocAsync inside the capture fails with either fill -- another matter, not torch's path)
    for (char v : variants)
        for (size_t n : sizes) {
            const long bad = run(v, n, 40, 20);
            printf("variant %c  n = %8zu floats  40 rounds x 20 replays: %ld wrong elements\n", v, n, bad);
            fflush(stdout);
        }
    // small and odd sizes at 4-byte-aligned offsets inside an allocation (what a caching allocator hands out is 512-byte aligned, but
    // the library zeroes slices of its workspaces)
    const size_t odd[] = {1, 3, 17, 100, 1000, 4097};
    const size_t offs[] = {0, 1, 3, 127};
    for (char v : {'A', 'D'})
        for (size_t n : odd)
            for (size_t o : offs) {
                const long bad = run(v, n, 40, 20, o);
                if (bad) printf("variant %c  n = %8zu floats at offset %zu: %ld wrong elements\n", v, n, o, bad);
            }
    printf("odd sizes / offsets done\n");
    return 0;
}
