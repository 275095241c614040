// Context, error reporting, workspaces and the entry points that only dispatch.
#include <stdarg.h>
#include <stdlib.h>

#include "hrs_internal.h"

static thread_local char g_err[512] = "";

void hrs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int hrs_version(void) { return 1; }

extern "C" int hrs_last_error(char* buf, size_t n) {
    size_t len = strlen(g_err);
    if (buf && n) {
        size_t m = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, m);
        buf[m] = 0;
    }
    return (int)len;
}

// Grows a workspace without waiting for the device.  hipFree synchronises the whole device, and a peer context may
// be in the middle of its pipeline on another stream, so the outgrown block is neither waited for nor freed here: it
// is parked on the context (kernels already queued may still use it) and released by hrs_ctx_destroy, behind that
// call's own device synchronisation.  Blocks double, so all the parked blocks of a workspace together are smaller
// than the block that replaced them.
int hrs_ws_reserve(hrs_ctx* ctx, hrs_workspace* ws, size_t bytes) {
    if (ws->bytes >= bytes) return HRS_OK;
    size_t want = bytes + 4096;
    if (want < 2 * ws->bytes) want = 2 * ws->bytes;
    void* fresh = nullptr;
    hipError_t e = hipMalloc(&fresh, want);
    if (e != hipSuccess) {
        hrs_set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return HRS_ERR_NOMEM;  // the old block stays in place and usable
    }
    if (ws->ptr) ctx->retired.push_back(ws->ptr);
    ws->ptr = fresh;
    ws->bytes = want;
    return HRS_OK;
}

void hrs_prof_begin(hrs_ctx* ctx, hipStream_t st) {
    if (!ctx->profile) return;
    if (ctx->ev_used == ctx->ev_start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        ctx->ev_start.push_back(a);
        ctx->ev_stop.push_back(b);
    }
    (void)hipEventRecord(ctx->ev_start[ctx->ev_used], st);
}

void hrs_prof_end(hrs_ctx* ctx, hipStream_t st) {
    if (!ctx->profile || ctx->ev_used >= ctx->ev_stop.size()) return;
    (void)hipEventRecord(ctx->ev_stop[ctx->ev_used], st);
    ctx->ev_used++;
}

extern "C" int hrs_profile_enable(hrs_ctx* ctx, int on) {
    HRS_REQUIRE(ctx, "hrs_profile_enable: null ctx");
    ctx->profile = on != 0;
    ctx->ev_used = 0;
    return HRS_OK;
}

extern "C" int hrs_profile_read(hrs_ctx* ctx, double* total_ms, int64_t* launches) {
    HRS_REQUIRE(ctx && total_ms && launches, "hrs_profile_read: null argument");
    double sum = 0.0;
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        HRS_CHECK_HIP(hipEventSynchronize(ctx->ev_stop[i]));
        float ms = 0.f;
        HRS_CHECK_HIP(hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)ctx->ev_used;
    ctx->ev_used = 0;
    return HRS_OK;
}

extern "C" int hrs_ctx_create(int device, hrs_ctx** out) {
    HRS_REQUIRE(out != nullptr, "hrs_ctx_create: out is null");
    int ndev = 0;
    HRS_CHECK_HIP(hipGetDeviceCount(&ndev));
    HRS_REQUIRE(device >= 0 && device < ndev, "hrs_ctx_create: device %d out of range (have %d)", device, ndev);
    HRS_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    HRS_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        hrs_set_error("hrs_ctx_create: device %d is %s; this library holds gfx950 (MI355X) code only", device,
                      prop.gcnArchName);
        return HRS_ERR_UNSUPPORTED;
    }
    int rc0 = hrs_exact_static_init(device);
    if (rc0) return rc0;
    hrs_ctx* c = new hrs_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : HRS_NUM_CU;
    c->scan_qt1 = getenv("HRS_SCAN_QT1") != nullptr;
    c->scan_shape32 = getenv("HRS_SCAN_SHAPE32") != nullptr;
    hipError_t e = hipMalloc((void**)&c->stats_dev, 16 * sizeof(int64_t));
    if (e != hipSuccess) {
        delete c;
        hrs_set_error("hrs_ctx_create: hipMalloc failed: %s", hipGetErrorString(e));
        return HRS_ERR_NOMEM;
    }
    e = hipMemset(c->stats_dev, 0, 16 * sizeof(int64_t));
    if (e != hipSuccess) {
        (void)hipFree(c->stats_dev);
        delete c;
        hrs_set_error("hrs_ctx_create: hipMemset failed: %s", hipGetErrorString(e));
        return HRS_ERR_HIP;
    }
    *out = c;
    return HRS_OK;
}

extern "C" int hrs_ctx_set_peer(hrs_ctx* ctx, hrs_ctx* peer) {
    HRS_REQUIRE(ctx, "hrs_ctx_set_peer: null ctx");
    HRS_REQUIRE(!peer || peer->device == ctx->device, "hrs_ctx_set_peer: contexts on different devices");
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    if (!ctx->scan_done) HRS_CHECK_HIP(hipEventCreateWithFlags(&ctx->scan_done, hipEventDisableTiming));
    if (!ctx->pre_done) HRS_CHECK_HIP(hipEventCreateWithFlags(&ctx->pre_done, hipEventDisableTiming));
    if (ctx->peer) ctx->peer->waiter = nullptr;
    ctx->peer = peer;
    if (peer) {
        if (peer->waiter && peer->waiter != ctx) peer->waiter->peer = nullptr;  // one waiter per context
        peer->waiter = ctx;
    }
    return HRS_OK;
}

extern "C" int hrs_ctx_destroy(hrs_ctx* ctx) {
    if (!ctx) return HRS_OK;
    if (ctx->waiter) ctx->waiter->peer = nullptr;  // nobody may wait on a destroyed context's event
    if (ctx->peer) ctx->peer->waiter = nullptr;
    // teardown: nothing useful can be done about an error here
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->scan_done) (void)hipEventDestroy(ctx->scan_done);
    if (ctx->pre_done) (void)hipEventDestroy(ctx->pre_done);
    if (ctx->ws_partial.ptr) (void)hipFree(ctx->ws_partial.ptr);
    if (ctx->ws_mfma.ptr) (void)hipFree(ctx->ws_mfma.ptr);
    if (ctx->ws_keys.ptr) (void)hipFree(ctx->ws_keys.ptr);
    if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
    for (void* p : ctx->retired) (void)hipFree(p);
    for (hipEvent_t e : ctx->ev_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_stop) (void)hipEventDestroy(e);
    delete ctx;
    return HRS_OK;
}

extern "C" int hrs_corpus_register(hrs_ctx* ctx, const float* C_dev, int64_t N, int D, int64_t row_offset,
                                   void* stream, hrs_corpus** out) {
    HRS_REQUIRE(ctx && out, "hrs_corpus_register: null ctx/out");
    HRS_REQUIRE(N >= 0 && (N == 0 || C_dev), "hrs_corpus_register: N=%lld with null matrix", (long long)N);
    HRS_REQUIRE(D > 0 && D % 4 == 0 && D <= 2048, "hrs_corpus_register: D=%d must be a multiple of 4 in (0, 2048]", D);
    HRS_REQUIRE(row_offset >= 0 && N + row_offset < (1ll << 32), "hrs_corpus_register: rows beyond 2^32");
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hrs_corpus* c = new hrs_corpus();
    c->C = C_dev;
    c->N = N;
    c->D = D;
    c->row_offset = row_offset;
    int rc = hrs_build_shadow(ctx, c, HRS_SHADOW_AUTO, (hipStream_t)stream);
    if (rc) {
        delete c;
        return rc;
    }
    *out = c;
    return HRS_OK;
}

static int check_topk_args(const char* fn, hrs_ctx* ctx, hrs_corpus* c, const float* Q_dev, int B, int k, int space, int mode);

extern "C" int hrs_corpus_append(hrs_ctx* ctx, hrs_corpus* c, const float* C_dev, int64_t N, void* stream) {
    HRS_REQUIRE(ctx && c, "hrs_corpus_append: null handle");
    HRS_REQUIRE(N >= c->N, "hrs_corpus_append: N=%lld is below the %lld rows registered", (long long)N, (long long)c->N);
    HRS_REQUIRE(N == 0 || C_dev, "hrs_corpus_append: N=%lld with null matrix", (long long)N);
    HRS_REQUIRE(N + c->row_offset < (1ll << 32), "hrs_corpus_append: rows beyond 2^32");
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    return hrs_append_shadow(ctx, c, C_dev, N, (hipStream_t)stream);
}

extern "C" int hrs_score_topk_after(hrs_ctx* ctx, hrs_corpus* c, const float* Q_dev, int B, int k, int space,
                                    const uint8_t* row_mask_dev, const uint64_t* after_keys_dev, uint64_t* keys_dev,
                                    float* dist_dev, int64_t* rows_dev, void* stream) {
    int rc = check_topk_args("hrs_score_topk_after", ctx, c, Q_dev, B, k, space, HRS_MODE_EXACT);
    if (rc) return rc;
    HRS_REQUIRE(after_keys_dev, "hrs_score_topk_after: null after_keys_dev");
    if (B == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    uint64_t* keys = keys_dev;
    if (!keys) {
        int rc0 = hrs_ws_reserve(ctx, &ctx->ws_keys, (size_t)B * k * sizeof(uint64_t));
        if (rc0) return rc0;
        keys = (uint64_t*)ctx->ws_keys.ptr;
    }
    hrs_unpacked un;
    un.dist = dist_dev;
    un.rows = rows_dev;
    return hrs_launch_exact(ctx, c, Q_dev, B, k, space, row_mask_dev, nullptr, nullptr, keys, (hipStream_t)stream, after_keys_dev, 0, -1, 0, un);
}

extern "C" int hrs_score_topk_levels(hrs_ctx* ctx, hrs_corpus* c, const float* Q_dev, int B, int k, int space,
                                     const uint8_t* level_dev, int64_t row_pitch, int levels, uint64_t* keys_dev, void* stream) {
    int rc = check_topk_args("hrs_score_topk_levels", ctx, c, Q_dev, B, k, space, HRS_MODE_EXACT);
    if (rc) return rc;
    HRS_REQUIRE(level_dev && keys_dev, "hrs_score_topk_levels: null argument");
    HRS_REQUIRE(levels >= 1 && levels <= 256 && row_pitch >= c->N, "hrs_score_topk_levels: levels=%d, row_pitch=%lld", levels,
                (long long)row_pitch);
    if (B == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    // all classes in one pass per four queries where the kernel's layout applies (HRS_LEVELS_BY_PASS: the older form)
    static const bool by_pass = getenv("HRS_LEVELS_BY_PASS") != nullptr;
    if (!by_pass) {
        rc = hrs_launch_exact_levels(ctx, c, Q_dev, B, k, space, level_dev, row_pitch, levels, keys_dev, (hipStream_t)stream);
        if (rc != HRS_ERR_UNSUPPORTED) return rc;
    }
    // one masked pass per level: the rows whose byte equals the level, each query under its own mask
    for (int l = 0; l < levels; ++l) {
        rc = hrs_launch_exact(ctx, c, Q_dev, B, k, space, level_dev, nullptr, nullptr, keys_dev + (size_t)l * k, (hipStream_t)stream,
                              nullptr, row_pitch, l, (int64_t)levels * k);
        if (rc) return rc;
    }
    return HRS_OK;
}

extern "C" int hrs_corpus_release(hrs_ctx* ctx, hrs_corpus* c) {
    if (!c) return HRS_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    hrs_free_shadow(c);
    delete c;
    return HRS_OK;
}

extern "C" int hrs_corpus_shadow(hrs_ctx* ctx, hrs_corpus* c, int kind, void* stream) {
    HRS_REQUIRE(ctx && c, "hrs_corpus_shadow: null handle");
    HRS_REQUIRE(kind >= HRS_SHADOW_NONE && kind <= HRS_SHADOW_AUTO, "hrs_corpus_shadow: unknown kind %d", kind);
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    HRS_CHECK_HIP(hipDeviceSynchronize());  // nothing may still read the shadow being replaced
    return hrs_build_shadow(ctx, c, kind, (hipStream_t)stream);
}

extern "C" int hrs_corpus_stats(hrs_ctx* ctx, hrs_corpus* c, void* stream, double* mn, double* mx) {
    HRS_REQUIRE(ctx && c, "hrs_corpus_stats: null handle");
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    double h[2] = {0, 0};
    if (c->N > 0) {
        HRS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        HRS_CHECK_HIP(hipMemcpy(h, c->norm_stats, sizeof(h), hipMemcpyDeviceToHost));
    }
    if (mn) *mn = h[0];
    if (mx) *mx = h[1];
    return HRS_OK;
}

__global__ void k_unpack(const uint64_t* keys, int64_t n, float* dist, int64_t* rows) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys[i];
    bool pad = k == HRS_PAD_KEY;
    if (dist) dist[i] = pad ? __builtin_inff() : orderable_f32((uint32_t)(k >> 32));
    if (rows) rows[i] = pad ? -1 : (int64_t)(k & 0xFFFFFFFFull);
}

extern "C" int hrs_unpack_keys(hrs_ctx* ctx, const uint64_t* keys_dev, int64_t n, float* dist_dev, int64_t* rows_dev,
                               void* stream) {
    HRS_REQUIRE(ctx && keys_dev && n >= 0, "hrs_unpack_keys: bad arguments");
    if (n == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys_dev, n,
                       dist_dev, rows_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// AUTO: large batches take the matrix cores.  Up to four queries: one pass of the fp64 streaming scan over the
// fp32 rows -- unless streaming the shadow (a quarter or half of the bytes) plus the fixed cost of the
// candidate machinery (~0.16 ms) is faster.  Measured cross-overs in elements of the corpus:
// int8 shadow 3e8 (one or two queries) and 1.5e8 (three or four); bf16 shadow 6e8 and 2e8.
static bool pick_mfma(const hrs_corpus* c, int B, int k, int mode) {
    bool use_mfma;
    if (mode == HRS_MODE_EXACT)
        use_mfma = false;
    else if (mode == HRS_MODE_MFMA)
        use_mfma = true;
    else {
        const double elems = (double)c->N * c->D;
        const bool i8 = c->kind == HRS_SHADOW_I8;
        const double cross = B <= 2 ? (i8 ? 3e8 : 6e8) : (i8 ? 1.5e8 : 2e8);
        use_mfma = c->N >= 65536 && (B > 4 || elems >= cross);
    }
    if (use_mfma && (c->N < 4096 || c->N < (int64_t)k * 64)) use_mfma = false;
    return use_mfma;
}

static int check_topk_args(const char* fn, hrs_ctx* ctx, hrs_corpus* c, const float* Q_dev, int B, int k, int space, int mode) {
    HRS_REQUIRE(ctx && c, "%s: null handle", fn);
    HRS_REQUIRE(B >= 0 && (B == 0 || Q_dev), "%s: B=%d with null queries", fn, B);
    HRS_REQUIRE(k >= 1 && k <= HRS_MAX_K, "%s: k=%d outside [1, %d]", fn, k, HRS_MAX_K);
    HRS_REQUIRE(space >= 0 && space <= 2, "%s: unknown space %d", fn, space);
    HRS_REQUIRE(mode >= 0 && mode <= 2, "%s: unknown mode %d", fn, mode);
    return HRS_OK;
}

extern "C" int hrs_score_topk(hrs_ctx* ctx, hrs_corpus* c, const float* Q_dev, int B, int k, int space,
                              const uint8_t* row_mask_dev, int mode, uint64_t* keys_dev, float* dist_dev,
                              int64_t* rows_dev, void* stream) {
    int rc = check_topk_args("hrs_score_topk", ctx, c, Q_dev, B, k, space, mode);
    if (rc) return rc;
    if (B == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    uint64_t* keys = keys_dev;
    if (!keys) {
        int rc0 = hrs_ws_reserve(ctx, &ctx->ws_keys, (size_t)B * k * sizeof(uint64_t));
        if (rc0) return rc0;
        keys = (uint64_t*)ctx->ws_keys.ptr;
    }
    // the kernels that write a query's final keys unpack them too: no launch of its own for that
    hrs_unpacked un;
    un.dist = dist_dev;
    un.rows = rows_dev;
    return pick_mfma(c, B, k, mode) ? hrs_launch_mfma(ctx, c, Q_dev, B, k, space, row_mask_dev, keys, st, HRS_PHASE_ALL, un)
                                    : hrs_launch_exact(ctx, c, Q_dev, B, k, space, row_mask_dev, nullptr, nullptr, keys, st, nullptr, 0, -1, 0, un);
}

extern "C" int hrs_score_topk_begin(hrs_ctx* ctx, hrs_corpus* c, const float* Q_dev, int B, int k, int space,
                                    const uint8_t* row_mask_dev, int mode, void* stream) {
    int rc = check_topk_args("hrs_score_topk_begin", ctx, c, Q_dev, B, k, space, mode);
    if (rc) return rc;
    HRS_REQUIRE(!ctx->pending.active, "hrs_score_topk_begin: the previous call on this context was not finished");
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    ctx->pending.c = c;
    ctx->pending.Q = Q_dev;
    ctx->pending.mask = row_mask_dev;
    ctx->pending.B = B;
    ctx->pending.k = k;
    ctx->pending.space = space;
    ctx->pending.mode = mode;
    // what cannot be split (the exact scan, more than one chunk of queries) runs whole in _finish
    ctx->pending.split = B > 0 && B <= 1024 && c->shadow && ctx->num_cu == HRS_NUM_CU && pick_mfma(c, B, k, mode);
    if (ctx->pending.split) {
        rc = hrs_launch_mfma(ctx, c, Q_dev, B, k, space, row_mask_dev, nullptr, (hipStream_t)stream, HRS_PHASE_SCAN);
        if (rc) return rc;
    }
    ctx->pending.active = true;
    return HRS_OK;
}

extern "C" int hrs_score_topk_finish(hrs_ctx* ctx, uint64_t* keys_dev, float* dist_dev, int64_t* rows_dev, void* stream) {
    HRS_REQUIRE(ctx, "hrs_score_topk_finish: null ctx");
    HRS_REQUIRE(ctx->pending.active, "hrs_score_topk_finish: no call was begun on this context");
    const auto p = ctx->pending;
    ctx->pending.active = false;
    if (!p.split)
        return hrs_score_topk(ctx, p.c, p.Q, p.B, p.k, p.space, p.mask, p.mode, keys_dev, dist_dev, rows_dev, stream);
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    uint64_t* keys = keys_dev;
    if (!keys) {
        int rc0 = hrs_ws_reserve(ctx, &ctx->ws_keys, (size_t)p.B * p.k * sizeof(uint64_t));
        if (rc0) return rc0;
        keys = (uint64_t*)ctx->ws_keys.ptr;
    }
    hrs_unpacked un;
    un.dist = dist_dev;
    un.rows = rows_dev;
    return hrs_launch_mfma(ctx, p.c, p.Q, p.B, p.k, p.space, p.mask, keys, (hipStream_t)stream, HRS_PHASE_TAIL, un);
}

extern "C" int hrs_score_stats(hrs_ctx* ctx, void* stream, int64_t out[3]) {
    HRS_REQUIRE(ctx && out, "hrs_score_stats: null argument");
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    HRS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    int64_t h[8];
    HRS_CHECK_HIP(hipMemcpy(h, ctx->stats_dev, sizeof(h), hipMemcpyDeviceToHost));
    out[0] = h[0];
    out[1] = h[1];
    out[2] = h[2];
    if (getenv("HRS_STATS_VERBOSE")) fprintf(stderr, "hrs_score_stats: fallbacks %lld, longest list %lld, overflows %lld, queries re-scored in more than one round %lld\n", (long long)h[0], (long long)h[1], (long long)h[2], (long long)h[3]);
#ifdef HRS_SCAN_PROF
    {
        int64_t g[16];
        HRS_CHECK_HIP(hipMemcpy(g, ctx->stats_dev, sizeof(g), hipMemcpyDeviceToHost));
        if (g[5] > 0)  // (meaningful when the main pass was the 16 x 16 x 64 one:) it stamps cycles and 100 MHz ticks around its loop
            fprintf(stderr, "main pass (16x16x64): clock in the loop %.3f GHz, loop %.1f us per workgroup\n", (double)g[4] / (double)g[5] * 0.1, g[5] / 256.0 / 100.0);
        fprintf(stderr, "scan per WG (us): prologue %.2f first-tile %.2f loop %.2f flush %.2f | start skew %.2f, first loop end after %.2f, last end after %.2f\n",
                g[4] / 25600.0, g[5] / 25600.0, g[6] / 25600.0, g[7] / 25600.0, (g[9] - g[8]) / 100.0, (g[10] - g[8]) / 100.0, (g[11] - g[8]) / 100.0);
    }
#endif
#ifdef HRS_SEL_PROF
    {
        int64_t g[16];
        HRS_CHECK_HIP(hipMemcpy(g, ctx->stats_dev, sizeof(g), hipMemcpyDeviceToHost));
        fprintf(stderr, "select: init %lld (list load %lld, sort %lld) resc %lld total %lld (x10 ns, summed over queries) rounds %lld maxcand %lld\n", (long long)g[12], (long long)g[6], (long long)g[7], (long long)g[13], (long long)g[14], (long long)g[15], (long long)h[1]);
    }
#endif
    return HRS_OK;
}

extern "C" int hrs_scan_work(hrs_ctx* ctx, void* stream, double* main_pass_share) {
    HRS_REQUIRE(ctx && main_pass_share, "hrs_scan_work: null argument");
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    HRS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    int64_t scored = 0;
    HRS_CHECK_HIP(hipMemcpy(&scored, ctx->stats_dev + 8, sizeof(scored), hipMemcpyDeviceToHost));
    *main_pass_share = (scored > 0 && ctx->last_tiles > 0) ? (double)scored / (double)ctx->last_tiles : 1.0;
    return HRS_OK;
}

extern "C" int hrs_postings_register(hrs_ctx* ctx, const int64_t* doc_offs_dev, const int32_t* term_ids_dev,
                                     int64_t N, int64_t row_offset, hrs_postings** out) {
    HRS_REQUIRE(ctx && out && doc_offs_dev, "hrs_postings_register: null argument");
    HRS_REQUIRE(N >= 0 && row_offset >= 0, "hrs_postings_register: negative size");
    hrs_postings* p = new hrs_postings();
    p->doc_offs = doc_offs_dev;
    p->term_ids = term_ids_dev;
    p->N = N;
    p->row_offset = row_offset;
    *out = p;
    return HRS_OK;
}

extern "C" int hrs_postings_release(hrs_ctx* ctx, hrs_postings* p) {
    (void)ctx;
    delete p;
    return HRS_OK;
}
