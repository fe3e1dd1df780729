// Probe: buffer_load_dwordx4 ... lds on gfx950 -- placement, out-of-range zero fill, whether soffset takes part in the range check.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* x, unsigned* y, int nbytes, int soff, int huge_lane) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)lds)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    int voff = threadIdx.x * 16;
    if ((int)threadIdx.x == huge_lane) voff = 0x7ffffff0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 1024), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) y[i] = ((unsigned*)lds)[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i + 1;
    unsigned *x, *y;
    hipMalloc(&x, 16384); hipMalloc(&y, 4096);
    hipMemcpy(x, h.data(), 16384, hipMemcpyHostToDevice);
    struct { int nbytes, soff, huge; } cases[] = {{16384, 0, -1}, {16384, 4096, 5}, {512, 0, -1}, {6144, 4096, -1}, {6144, 8192, -1}};
    for (auto c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, x, y, c.nbytes, c.soff, c.huge);
        std::vector<unsigned> o(1024);
        hipMemcpy(o.data(), y, 4096, hipMemcpyDeviceToHost);
        printf("nbytes %d soff %d huge_lane %d: lds[0]=%x  dst words:", c.nbytes, c.soff, c.huge);
        for (int l : {0, 1, 5, 31, 32, 33, 63}) printf(" L%d=%u", l, o[256 + l * 4]);
        printf("  after=%x\n", o[256 + 256]);
    }
    return 0;
}
