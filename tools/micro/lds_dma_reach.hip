// Does LDS-DMA (global_load_lds_dwordx4) reach every byte of a 160 KB dynamic LDS allocation?
// One workgroup of 256 threads copies SIZE bytes of a known pattern into LDS by LDS-DMA, 1 KB per
// wave-instruction, reads all of it back with ds_read_b128 and writes it to a second global buffer;
// the host compares.  Prints the first 1 KB block that differs, if any.
//   hipcc --offload-arch=gfx950 -O2 -I../../hybrid_rag_system_amd/csrc -I../../include lds_dma_reach.hip -o lds_dma_reach
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hrs_internal.h"

__global__ __launch_bounds__(256) void k_reach(const char* __restrict__ src, char* __restrict__ dst, int size) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = lds_addr_of(smem);
    const int pieces = size / 1024;
    for (int p = wv; p < pieces; p += 4) {
        const char* base = uniform_ptr(src + (size_t)p * 1024);
        const uint32_t d = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)p * 1024);
        glds16_asm(base, lane * 16, d);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int o = threadIdx.x * 16; o < size; o += 256 * 16) *(uint4*)(dst + o) = *(const uint4*)(smem + o);
}

int main() {
    const int sizes[] = {64 * 1024, 128 * 1024, 144 * 1024, 160 * 1024};
    for (int size : sizes) {
        std::vector<unsigned> h(size / 4), back(size / 4, 0);
        for (int i = 0; i < size / 4; ++i) h[i] = 0x9E3779B1u * (unsigned)(i + 1);
        char *s, *d;
        if (hipMalloc(&s, size) != hipSuccess || hipMalloc(&d, size) != hipSuccess) return 1;
        hipMemcpy(s, h.data(), size, hipMemcpyHostToDevice);
        hipMemset(d, 0, size);
        hipError_t e = hipFuncSetAttribute((const void*)k_reach, hipFuncAttributeMaxDynamicSharedMemorySize, size);
        if (e != hipSuccess) {
            printf("size %d KB: attribute refused: %s\n", size / 1024, hipGetErrorString(e));
            continue;
        }
        hipLaunchKernelGGL(k_reach, dim3(1), dim3(256), size, 0, s, d, size);
        e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            printf("size %d KB: launch failed: %s\n", size / 1024, hipGetErrorString(e));
            return 2;
        }
        hipMemcpy(back.data(), d, size, hipMemcpyDeviceToHost);
        int first = -1, bad = 0;
        for (int b = 0; b < size / 1024; ++b) {
            bool ok = true;
            for (int i = 0; i < 256; ++i) ok = ok && back[b * 256 + i] == h[b * 256 + i];
            if (!ok) {
                ++bad;
                if (first < 0) first = b;
            }
        }
        printf("size %d KB: %d of %d 1-KB blocks differ%s", size / 1024, bad, size / 1024, bad ? "" : "\n");
        if (bad) printf(" (first at %d KB)\n", first);
        hipFree(s);
        hipFree(d);
    }
    return 0;
}
