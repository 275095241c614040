// Counter-based synthetic inputs: any shard reproduces the same rows without a host copy.
#include "hrs_internal.h"

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float u01(uint32_t bits) {  // (0, 1]
    return ((float)(bits >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// One wave per row: lane l produces columns l, l+64, ...; pairs of columns share one Box-Muller draw.
__global__ __launch_bounds__(256) void k_synth_rows(uint64_t seed, int64_t row0, int64_t nrows, int D, int normalize,
                                                    float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const uint64_t grow = (uint64_t)(row0 + r);
    float* dst = out + r * D;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) {
        const uint64_t h = mix64(mix64(seed ^ (grow * 0xD1B54A32D192ED03ull)) + (uint64_t)(c >> 1));
        const float u1 = u01((uint32_t)h), u2 = u01((uint32_t)(h >> 32));
        const float rad = sqrtf(-2.0f * __logf(u1));
        float s, co;
        __sincosf(6.28318530717958647692f * u2, &s, &co);
        const float v = rad * ((c & 1) ? s : co);
        dst[c] = v;
        ss += v * v;
    }
    if (normalize) {
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float inv = 1.0f / sqrtf(ss);
        for (int c = lane; c < D; c += 64) dst[c] *= inv;
    }
}

extern "C" int hrs_synth_rows(hrs_ctx* ctx, uint64_t seed, int64_t row0, int64_t nrows, int D, int normalize,
                              float* out_dev, void* stream) {
    HRS_REQUIRE(ctx && out_dev && nrows >= 0 && D > 0, "hrs_synth_rows: bad arguments");
    if (nrows == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_synth_rows, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, seed, row0,
                       nrows, D, normalize, out_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// One thread per row: draw terms by inverse CDF until L distinct ones are held (sorted insertion).
template <int LMAX>
__global__ __launch_bounds__(64) void k_synth_postings(uint64_t seed, int64_t row0, int64_t nrows, int V, int L,
                                                       const float* __restrict__ cdf, int32_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= nrows) return;
    int32_t t[LMAX];
    int n = 0;
    const uint64_t base = mix64(seed ^ ((uint64_t)(row0 + r) * 0xA24BAED4963EE407ull));
    for (uint32_t draw = 0; n < L; ++draw) {
        int term;
        if (draw < (uint32_t)(8 * L + 64)) {
            const float u = u01((uint32_t)mix64(base + draw));
            int lo = 0, hi = V - 1;  // first index with cdf >= u
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] >= u)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            term = lo;
        } else {
            term = (int)((mix64(base + draw) >> 11) % (uint64_t)V);  // guarantees termination for L <= V
        }
        int pos = 0;
        while (pos < n && t[pos] < term) ++pos;
        if (pos < n && t[pos] == term) continue;
        for (int j = n; j > pos; --j) t[j] = t[j - 1];
        t[pos] = term;
        ++n;
    }
    for (int j = 0; j < L; ++j) out[r * L + j] = t[j];
}

extern "C" int hrs_synth_postings(hrs_ctx* ctx, uint64_t seed, int64_t row0, int64_t nrows, int V, int L,
                                  const float* cdf_dev, int32_t* out_dev, void* stream) {
    HRS_REQUIRE(ctx && cdf_dev && out_dev && nrows >= 0, "hrs_synth_postings: bad arguments");
    HRS_REQUIRE(L >= 1 && L <= 128 && L <= V, "hrs_synth_postings: L=%d outside [1, min(128, V)]", L);
    if (nrows == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL((k_synth_postings<128>), dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       seed, row0, nrows, V, L, cdf_dev, out_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}
