/* The C ABI used from plain C, no Python and no torch: device buffers come from hipMalloc, the stream
 * is the null stream.  Builds a 200k x 384 corpus with hrs_synth_rows, asks for the top-10 of 16 queries
 * (corpus rows themselves) on both scoring paths and checks: same keys on both paths, every query's
 * first hit is its own row at distance 0, keys strictly ascending.  Exit code 0 = pass. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hrs.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HRS(x) do { int r_ = (x); if (r_ != 0) { char m_[512]; hrs_last_error(m_, sizeof m_); printf("%s -> %d: %s\n", #x, r_, m_); return 3; } } while (0)

int main(void) {
    const int64_t N = 200000;
    const int D = 384, B = 16, K = 10;
    hrs_ctx* ctx = NULL;
    CHECK_HRS(hrs_ctx_create(0, &ctx));
    float *C = NULL, *Q = NULL, *dist = NULL;
    int64_t* rows = NULL;
    uint64_t *k1 = NULL, *k2 = NULL;
    CHECK_HIP(hipMalloc((void**)&C, (size_t)N * D * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&Q, (size_t)B * D * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dist, (size_t)B * K * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&rows, (size_t)B * K * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void**)&k1, (size_t)B * K * sizeof(uint64_t)));
    CHECK_HIP(hipMalloc((void**)&k2, (size_t)B * K * sizeof(uint64_t)));
    CHECK_HRS(hrs_synth_rows(ctx, 99, 0, N, D, 1, C, NULL));
    for (int b = 0; b < B; ++b)  /* query b = corpus row 1000 b + 7 */
        CHECK_HIP(hipMemcpyAsync(Q + (size_t)b * D, C + (size_t)(1000 * b + 7) * D, D * sizeof(float), hipMemcpyDeviceToDevice, NULL));
    hrs_corpus* corpus = NULL;
    CHECK_HRS(hrs_corpus_register(ctx, C, N, D, 0, NULL, &corpus));
    CHECK_HRS(hrs_score_topk(ctx, corpus, Q, B, K, HRS_SPACE_L2, NULL, HRS_MODE_MFMA, k1, dist, rows, NULL));
    CHECK_HRS(hrs_score_topk(ctx, corpus, Q, B, K, HRS_SPACE_L2, NULL, HRS_MODE_EXACT, k2, NULL, NULL, NULL));
    /* the same call in two halves, as a pipelining caller issues it */
    uint64_t* k3 = NULL;
    CHECK_HIP(hipMalloc((void**)&k3, (size_t)B * K * sizeof(uint64_t)));
    CHECK_HRS(hrs_score_topk_begin(ctx, corpus, Q, B, K, HRS_SPACE_L2, NULL, HRS_MODE_MFMA, NULL));
    if (hrs_score_topk_begin(ctx, corpus, Q, B, K, HRS_SPACE_L2, NULL, HRS_MODE_MFMA, NULL) != HRS_ERR_ARG) { printf("a second begin was accepted\n"); return 4; }
    CHECK_HRS(hrs_score_topk_finish(ctx, k3, NULL, NULL, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    uint64_t h3[16 * 10];
    CHECK_HIP(hipMemcpy(h3, k3, sizeof h3, hipMemcpyDeviceToHost));
    hipFree(k3);
    uint64_t h1[16 * 10], h2[16 * 10];
    float hd[16 * 10];
    int64_t hr[16 * 10];
    CHECK_HIP(hipMemcpy(h1, k1, sizeof h1, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h2, k2, sizeof h2, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hd, dist, sizeof hd, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hr, rows, sizeof hr, hipMemcpyDeviceToHost));
    int bad = 0;
    if (memcmp(h1, h2, sizeof h1) != 0) { printf("the two scoring paths disagree\n"); bad = 1; }
    if (memcmp(h1, h3, sizeof h1) != 0) { printf("the two-phase call disagrees\n"); bad = 1; }
    for (int b = 0; b < B; ++b) {
        if (hr[b * K] != 1000 * b + 7 || hd[b * K] != 0.0f) { printf("query %d: first hit row %lld dist %g\n", b, (long long)hr[b * K], hd[b * K]); bad = 1; }
        for (int j = 1; j < K; ++j)
            if (h1[b * K + j] <= h1[b * K + j - 1]) { printf("query %d: keys not ascending\n", b); bad = 1; }
    }
    int64_t st[3];
    CHECK_HRS(hrs_score_stats(ctx, NULL, st));
    printf("abi_smoke: %s (fallback queries %lld, max candidates %lld)\n", bad ? "FAIL" : "ok", (long long)st[0], (long long)st[1]);
    CHECK_HRS(hrs_corpus_release(ctx, corpus));
    CHECK_HRS(hrs_ctx_destroy(ctx));
    hipFree(C); hipFree(Q); hipFree(dist); hipFree(rows); hipFree(k1); hipFree(k2);
    return bad;
}
