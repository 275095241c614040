// Issue-rate microbenchmarks for v_mfma_f32_32x32x16_bf16 on gfx950 (one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// V: 0 builtin, 2 chains | 1 builtin, 1 chain | 2 asm VGPR-acc AGPR-B 2 chains | 3 = 2 + s_nop 1 |
//    4 builtin 2 chains + one ds_read_b128 per 2 MFMAs | 5 builtin 1 chain + ds_read per MFMA | 6 builtin 4 chains
template <int V, bool RND = false>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)(float)(lane - j);
        if (RND) {
            unsigned hsh = (threadIdx.x * 131 + j * 31 + blockIdx.x * 7919) * 2654435761u;
            a[j] = (__bf16)(((int)(hsh >> 16) - 32768) / 327680.0f);
            b[j] = (__bf16)(((int)(hsh & 0xffff) - 32768) / 327680.0f);
        }
    }
    for (int j = threadIdx.x; j < 65536 / 4; j += blockDim.x) ((float*)lds)[j] = j;
    __syncthreads();
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    typedef const __attribute__((address_space(3))) bf16x8* lp;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + (threadIdx.x >> 6) * 4096;
    bf16x8 bb = b;
    if (V == 2 || V == 3) asm volatile("" : "+a"(bb));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (V == 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            } else if (V == 1) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            } else if (V == 2) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "a"(bb));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "a"(bb));
            } else if (V == 3) {
                asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "a"(bb));
                asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "a"(bb));
            } else if (V == 4) {
                bf16x8 t = *(lp)(size_t)(base + (u & 3) * 1024);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                a[0] += t[0];
            } else if (V == 5) {
                bf16x8 t = *(lp)(size_t)(base + (u & 3) * 1024);
                bf16x8 t2 = *(lp)(size_t)(base + (u & 3) * 1024 + 512);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                a[0] += t[0] + t2[0];
            } else if (V == 6) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// Staged variants: one "stage" = KS k-steps, each one ds_read_b128 feeding two MFMAs (AGPR B operands).
//   S=0 reads only | S=1 + s_barrier per stage | S=2 + LDS-DMA of the stage (24 KB, L2 resident) with counted vmcnt
__device__ __forceinline__ void glds16(const char* sbase, int voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
template <int S, int NW, bool I8 = false>
__global__ __launch_bounds__(NW * 64) void kst(float* out, const char* src, int stages) {
    constexpr int KS = 24, PF = 4, STAGE = 24576, NBUF = 6, IPW = STAGE / (NW * 1024);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int j = threadIdx.x; j < NBUF * STAGE / 4; j += blockDim.x) ((float*)lds)[j] = 0.001f * (j & 255);
    __syncthreads();
    bf16x8 q0[KS], q1[KS];
    for (int s = 0; s < KS; ++s)
        for (int j = 0; j < 8; ++j) {
            unsigned hsh = (lane * 131 + j * 31 + s * 7919) * 2654435761u;
            q0[s][j] = (__bf16)(((int)(hsh >> 16) - 32768) / 327680.0f);
            q1[s][j] = (__bf16)(((int)(hsh & 0xffff) - 32768) / 327680.0f);
        }
    for (int s = 0; s < KS; ++s) { asm volatile("" : "+a"(q0[s])); asm volatile("" : "+a"(q1[s])); }
    typedef const __attribute__((address_space(3))) bf16x8* lp;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int i = lane & 31, h = lane >> 5;
    f32x16 c0, c1;
    float acc = 0.f;
    const char* base = src;
    for (int g = 0; g < stages; ++g) {
        const unsigned sb = lds0 + (g % NBUF) * STAGE + i * 768;
        if (S >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
        if (S >= 1) __builtin_amdgcn_s_barrier();
        if (S >= 2) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + ((g + NBUF - 1) % NBUF) * STAGE + wv * IPW * 1024);
            const char* b2 = base + (size_t)((blockIdx.x * 64 + (g & 63)) % 4096) * STAGE;
            const char* b3 = (const char*)(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((size_t)b2 >> 32)) << 32) | (size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)b2));
#pragma unroll
            for (int ii = 0; ii < IPW; ++ii) glds16(b3, (wv * IPW + ii) * 1024 + lane * 16, dst + ii * 1024);
        }
        bf16x8 av[KS];
#pragma unroll
        for (int s = 0; s < PF; ++s) av[s] = *(lp)(size_t)(sb + ((2 * s + h) ^ (i & 15)) * 16);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + PF < KS) av[s + PF] = *(lp)(size_t)(sb + (((2 * (s + PF) + h) & 15) ^ (i & 15)) * 16 + ((s + PF) >> 3) * 256);
            if (I8) {
                if (s == 0) {
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(c0) : "v"(av[s]), "a"(q0[s]));
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(c1) : "v"(av[s]), "a"(q1[s]));
                } else {
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c0) : "v"(av[s]), "a"(q0[s]));
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c1) : "v"(av[s]), "a"(q1[s]));
                }
            } else if (s == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c0) : "v"(av[s]), "a"(q0[s]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c1) : "v"(av[s]), "a"(q1[s]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(av[s]), "a"(q0[s]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(av[s]), "a"(q1[s]));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c0), "+v"(c1));
        float m = c0[0];
        for (int r = 1; r < 16; ++r) m = fmaxf(m, fmaxf(c0[r], c1[r]));
        acc = fmaxf(acc, m);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// The same stage with the 16x16 MFMA shapes at the same output tile per wave (32 rows x 64 queries): fragment s is
// (k-step s >> 1, row half s & 1) and feeds four MFMAs, one per group of 16 queries (v_mfma_i32_16x16x64_i8 /
// v_mfma_f32_16x16x32_bf16: half the cycles of the 32x32 forms, a quarter of the output).  Same LDS bytes, same
// operand registers (48 query fragments), same matrix-pipe cycles per stage.
typedef __attribute__((ext_vector_type(4))) float f32x4m;
template <int S, int NW, bool I8 = false>
__global__ __launch_bounds__(NW * 64) void kst16(float* out, const char* src, int stages) {
    constexpr int KS = 24, PF = 4, STAGE = 24576, NBUF = 6, IPW = STAGE / (NW * 1024);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int j = threadIdx.x; j < NBUF * STAGE / 4; j += blockDim.x) ((float*)lds)[j] = 0.001f * (j & 255);
    __syncthreads();
    bf16x8 q[4][KS / 2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < KS / 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned hsh = (lane * 131 + j * 31 + s * 7919 + g * 104729) * 2654435761u;
                q[g][s][j] = (__bf16)(((int)(hsh >> 16) - 32768) / 327680.0f);
            }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < KS / 2; ++s) asm volatile("" : "+a"(q[g][s]));
    typedef const __attribute__((address_space(3))) bf16x8* lp;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int i = lane & 15, kg = lane >> 4;
    f32x4m c[2][4];
    float acc = 0.f;
    const char* base = src;
    for (int g = 0; g < stages; ++g) {
        const unsigned sb = lds0 + (g % NBUF) * STAGE + i * 768;
        if (S >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
        if (S >= 1) __builtin_amdgcn_s_barrier();
        if (S >= 2) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + ((g + NBUF - 1) % NBUF) * STAGE + wv * IPW * 1024);
            const char* b2 = base + (size_t)((blockIdx.x * 64 + (g & 63)) % 4096) * STAGE;
            const char* b3 = (const char*)(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((size_t)b2 >> 32)) << 32) | (size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)b2));
#pragma unroll
            for (int ii = 0; ii < IPW; ++ii) glds16(b3, (wv * IPW + ii) * 1024 + lane * 16, dst + ii * 1024);
        }
        // fragment s: rows 16 (s & 1) + i, 16-byte piece 4 (s >> 1) + kg of the row (48 pieces), XOR-permuted as the scan does
        auto addr = [&](int s) { return sb + (s & 1) * 16 * 768 + ((((4 * (s >> 1) + kg) & 15) ^ (i & 15)) * 16) + ((4 * (s >> 1)) >> 4) * 256; };
        bf16x8 av[KS];
#pragma unroll
        for (int s = 0; s < PF; ++s) av[s] = *(lp)(size_t)addr(s);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + PF < KS) av[s + PF] = *(lp)(size_t)addr(s + PF);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (I8) {
                    if (s < 2)
                        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(c[s & 1][gq]) : "v"(av[s]), "a"(q[gq][s >> 1]));
                    else
                        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[s & 1][gq]) : "v"(av[s]), "a"(q[gq][s >> 1]));
                } else {
                    if (s < 2)
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c[s & 1][gq]) : "v"(av[s]), "a"(q[gq][s >> 1]));
                    else
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[s & 1][gq]) : "v"(av[s]), "a"(q[gq][s >> 1]));
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[0][3]), "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[1][2]), "+v"(c[1][3]));
        float m = c[0][0][0];
        for (int hh = 0; hh < 2; ++hh)
            for (int gq = 0; gq < 4; ++gq)
                for (int r = 0; r < 4; ++r) m = fmaxf(m, c[hh][gq][r]);
        acc = fmaxf(acc, m);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int S, int NW, bool I8 = false>
void runst16(const char* name, float* out, const char* src) {
    const int stages = 5000, lds = 6 * 24576;
    CHECK(hipFuncSetAttribute((const void*)kst16<S, NW, I8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kst16<S, NW, I8>), dim3(256), dim3(NW * 64), lds, 0, out, src, 50);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((kst16<S, NW, I8>), dim3(256), dim3(NW * 64), lds, 0, out, src, stages);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)stages * 96 * NW * 256;  // 24 fragments x 4 MFMAs, each a quarter of a 32x32 one's work
    printf("%-46s waves=%d  %.3f ms  %.0f T(FL)OP/s\n", name, NW, ms, mfma * (I8 ? 32768 : 16384) / (ms * 1e-3) / 1e12);
}

template <int S, int NW, bool I8 = false>
void runst(const char* name, float* out, const char* src) {
    const int stages = 5000, lds = 6 * 24576;
    CHECK(hipFuncSetAttribute((const void*)kst<S, NW, I8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kst<S, NW, I8>), dim3(256), dim3(NW * 64), lds, 0, out, src, 50);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((kst<S, NW, I8>), dim3(256), dim3(NW * 64), lds, 0, out, src, stages);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)stages * 48 * NW * 256;
    printf("%-46s waves=%d  %.3f ms  %.0f T(FL)OP/s\n", name, NW, ms, mfma * (I8 ? 65536 : 32768) / (ms * 1e-3) / 1e12);
}

template <int V, bool RND = false>
void run(const char* name, int waves, float* out, int per_iter) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<V, RND>), dim3(256), dim3(waves * 64), 0, 0, out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V, RND>), dim3(256), dim3(waves * 64), 0, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)iters * 16 * per_iter * waves * 256;
    printf("%-46s waves=%d  %.3f ms  %.0f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, waves, ms, mfma * 32768 / (ms * 1e-3) / 1e12,
           ms * 1e6 / ((double)iters * 16 * per_iter * (waves > 4 ? waves / 4 : 1)));
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 512 * 4));
    for (int w : {4, 8}) {
        run<0>("builtin 2 chains", w, out, 2);
        run<0, true>("builtin 2 chains, random operands", w, out, 2);
        run<2, true>("asm agpr-B 2 chains, random operands", w, out, 2);
        run<5, true>("1 chain + 2 ds_read / 2 MFMA, random operands", w, out, 2);
        run<1>("builtin 1 chain", w, out, 2);
        run<6>("builtin 4 chains", w, out, 4);
        run<2>("asm vgpr-acc agpr-B 2 chains", w, out, 2);
        run<3>("asm + s_nop 1", w, out, 2);
        run<4>("builtin 2 chains + ds_read_b128 / 2 MFMA", w, out, 2);
        run<5>("builtin 1 chain + 2 ds_read_b128 / 2 MFMA", w, out, 2);
    }
    char* src; CHECK(hipMalloc(&src, (size_t)4096 * 24576 + 65536)); CHECK(hipMemset(src, 0, (size_t)4096 * 24576));
    runst<2, 4>("staged + barrier + LDS-DMA (zeros)", out, src);
    {   // random bf16 in (-0.1, 0.1): realistic operand toggling
        size_t n = (size_t)4096 * 24576 / 2;
        unsigned short* hbuf = (unsigned short*)malloc(n * 2);
        unsigned long long x = 88172645463325252ull;
        for (size_t j = 0; j < n; ++j) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            float f = ((int)(x & 0xffff) - 32768) / 327680.0f;
            unsigned u; memcpy(&u, &f, 4);
            hbuf[j] = (unsigned short)(u >> 16);
        }
        CHECK(hipMemcpy(src, hbuf, n * 2, hipMemcpyHostToDevice));
        free(hbuf);
    }
    runst<0, 4>("staged: ds_read -> 2 MFMA", out, src);
    runst<1, 4>("staged + barrier", out, src);
    runst<2, 4>("staged + barrier + LDS-DMA", out, src);
    runst<0, 4, true>("i8 staged: ds_read -> 2 MFMA", out, src);
    runst<2, 4, true>("i8 staged + barrier + LDS-DMA", out, src);
    runst<0, 8, true>("i8 staged: ds_read -> 2 MFMA", out, src);
    runst<2, 8, true>("i8 staged + barrier + LDS-DMA", out, src);
    // the 16x16 shapes at the same output tile per wave, alternating with the 32x32 ones on the same data
    for (int rep = 0; rep < 2; ++rep) {
        runst<2, 4, false>("bf16 32x32x16 staged + barrier + LDS-DMA", out, src);
        runst16<2, 4, false>("bf16 16x16x32 staged + barrier + LDS-DMA", out, src);
        runst<2, 4, true>("i8 32x32x32 staged + barrier + LDS-DMA", out, src);
        runst16<2, 4, true>("i8 16x16x64 staged + barrier + LDS-DMA", out, src);
        runst<2, 8, true>("i8 32x32x32 staged + barrier + LDS-DMA", out, src);
        runst16<2, 8, true>("i8 16x16x64 staged + barrier + LDS-DMA", out, src);
        runst<0, 4, true>("i8 32x32x32 staged: reads + MFMA only", out, src);
        runst16<0, 4, true>("i8 16x16x64 staged: reads + MFMA only", out, src);
    }
    return 0;
}
