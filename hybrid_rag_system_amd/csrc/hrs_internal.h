// Internal declarations shared by the translation units of libhrs.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <vector>

#include "../../include/hrs.h"

#define HRS_NUM_CU 256      // MI355X: 8 XCDs x 32 CUs
#define HRS_EXACT_CB 512    // per-query candidate buffer of the exact scan (keys)
#define HRS_CAND_CAP 4096   // per-query candidate list of the MFMA scan as the fast re-score kernel sorts it
#define HRS_RESCORE_MAX 1024  // largest prefix the fast re-score kernel re-scores, a power of two (with these two its
                              // LDS is exactly 80 KB: two workgroups per CU, as with half of each)
#define HRS_CAND_CAP_BIG 8192  // capacity of the list in memory (longer lists go to the large re-score kernel)
#define HRS_RESCORE 256     // candidates re-scored exactly per query
#define HRS_MAX_STREAMS 1024  // per-query running maxima written by the sampling pass

void hrs_set_error(const char* fmt, ...);

#define HRS_CHECK_HIP(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            hrs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return HRS_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

#define HRS_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            hrs_set_error(__VA_ARGS__); \
            return HRS_ERR_ARG;       \
        }                             \
    } while (0)

// "Done once per device" flags for per-function attributes (hipFuncSetAttribute applies to the current device
// only); safe when contexts of different devices are driven from different threads (doing it twice is harmless).
struct hrs_per_device_once {
    std::atomic<uint64_t> mask{0};
    bool need(int device) const { return device < 0 || device >= 64 || !((mask.load(std::memory_order_acquire) >> device) & 1); }
    void done(int device) {
        if (device >= 0 && device < 64) mask.fetch_or(1ull << device, std::memory_order_release);
    }
};

struct hrs_workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
};

struct hrs_ctx {
    int device = 0;
    int num_cu = HRS_NUM_CU;
    bool scan_qt1 = false;     // HRS_SCAN_QT1 in the environment at creation: one query tile per wave (tuning switch)
    bool scan_shape32 = false; // HRS_SCAN_SHAPE32: the int8 main pass on the 32 x 32 x 32 MFMA shape (tuning switch)
    hrs_workspace ws_partial;  // exact scan: per (query, workgroup) top-k keys, and merge ping-pong
    hrs_workspace ws_mfma;     // MFMA scan: bf16 queries, maxima, tau, counters, candidate lists
    hrs_workspace ws_keys;     // result keys when the caller asks only for distances / rows
    std::vector<void*> retired;  // outgrown workspace blocks, freed when the context is destroyed (see hrs_ws_reserve)
    int64_t* stats_dev = nullptr;  // [16] counters of the last MFMA call
    int64_t last_tiles = 0;        // corpus tiles of the last MFMA call (see hrs_scan_work)
    // two contexts that serve alternate batches on two streams: each scan waits for the peer's last scan,
    // so scans never share the matrix pipes with each other, only with the peer's re-score tail
    hrs_ctx* peer = nullptr;
    hrs_ctx* waiter = nullptr;  // the context whose `peer` this one is (cleared when either is destroyed)
    hipEvent_t scan_done = nullptr;
    bool scan_recorded = false;
    // two-phase calls (hrs_score_topk_begin / _finish): the tail of a call waits until the waiter's next
    // scan has passed its sampled passes, so that it runs under that scan's main pass
    hipEvent_t pre_done = nullptr;
    bool pre_recorded = false;
    struct {
        bool active = false, split = false;
        hrs_corpus* c = nullptr;
        const float* Q = nullptr;
        const uint8_t* mask = nullptr;
        int B = 0, k = 0, space = 0, mode = 0;
    } pending;
    // optional kernel timing (hrs_profile_*): event pairs around the dominant scan kernel
    bool profile = false;
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t ev_used = 0;
};

struct hrs_corpus {
    const float* C = nullptr;  // caller-owned fp32 [N, D]
    int64_t N = 0;
    int D = 0;
    int64_t row_offset = 0;
    int kind = 0;              // hrs_shadow_kind of the scan shadow
    int want_kind = HRS_SHADOW_AUTO;  // what the caller asked for (an empty corpus has no shadow yet)
    int64_t cap = 0;           // rows the shadow allocation holds (>= Npad); grows geometrically on append
    void* shadow = nullptr;    // owned: bf16 or int8 [Npad, Dpad], zero padded
    float4* group_pre = nullptr;   // owned (int8 shadow): per 32-row group (scale, E_g, C_g + E_g, -); then, in the same
    float4* group_main = nullptr;  // allocation, (1 / scale, E_g, C_g + E_g, -)
    int64_t Npad = 0;
    int Dpad = 0;
    double* norm_stats = nullptr;  // owned: {min |c|^2, max |c|^2} as doubles
};

struct hrs_inverted {
    const int64_t* term_offs = nullptr;  // [V + 1]
    const int32_t* doc_ids = nullptr;    // [nnz] local rows of this shard, ascending within a term
    int64_t V = 0, N = 0;
};

struct hrs_postings {
    const int64_t* doc_offs = nullptr;
    const int32_t* term_ids = nullptr;
    int64_t N = 0;
    int64_t row_offset = 0;
};

int hrs_ws_reserve(hrs_ctx* ctx, hrs_workspace* ws, size_t bytes);
// Event pair around one launch of the dominant kernel when profiling is on (no-ops otherwise).
void hrs_prof_begin(hrs_ctx* ctx, hipStream_t st);
void hrs_prof_end(hrs_ctx* ctx, hipStream_t st);

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t pack_key(float d32, uint32_t row) {
    return ((uint64_t)f32_orderable(d32) << 32) | row;
}

// One final key, unpacked into the caller's distance / row arrays (either may be null).
__device__ __forceinline__ void hrs_store_unpacked(uint64_t key, size_t at, float* dist, int64_t* rows) {
    const bool pad = key == HRS_PAD_KEY;
    if (dist) dist[at] = pad ? __builtin_inff() : orderable_f32((uint32_t)(key >> 32));
    if (rows) rows[at] = pad ? -1 : (int64_t)(key & 0xFFFFFFFFull);
}

// Bitonic sort of n (power of two) uint64 keys in LDS, ascending; all threads of the block call it.
// Every thread owns compare-exchange PAIRS (pair p of a pass with stride j joins i = p with a zero bit
// inserted at j, and i + j), four at a time so that their LDS reads are in flight together.
__device__ __forceinline__ void bitonic_sort_u64(uint64_t* s, int n, int tid, int nthreads) {
    const int half = n >> 1;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int p0 = tid; p0 < half; p0 += 4 * nthreads) {
                uint64_t a[4], b[4];
                int ii[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + u * nthreads;
                    ii[u] = p < half ? (((p & ~(j - 1)) << 1) | (p & (j - 1))) : -1;
                    if (ii[u] >= 0) {
                        a[u] = s[ii[u]];
                        b[u] = s[ii[u] + j];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ii[u] >= 0 && (a[u] > b[u]) == ((ii[u] & k) == 0)) {
                        s[ii[u]] = b[u];
                        s[ii[u] + j] = a[u];
                    }
                }
            }
        }
    }
    __syncthreads();
}

// The same order for 256 * E keys (E = 1, 2, 4 or 8) and a workgroup of 256 threads, with the keys in registers:
// thread t holds keys t*E .. t*E+E-1.  Compare-exchanges at a distance below E are local, those within a wave
// (distance below 64 E) go through lane shuffles, and only the few across waves go through LDS -- three stages
// with a barrier instead of one barrier per stage (36 to 66 of them).  s must be complete (a barrier behind the
// last write) on entry; it is complete and sorted on return.
template <int E>
__device__ __forceinline__ void bitonic_sort_u64_regs(uint64_t* s, int tid) {
    constexpr int n = 256 * E;
    uint64_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = s[tid * E + e];
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j >= E; j >>= 1) {  // partners in other threads
            if (j >= 64 * E) {
                __syncthreads();  // every thread has read what it wanted from the previous contents
#pragma unroll
                for (int e = 0; e < E; ++e) s[tid * E + e] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    const uint64_t o = s[i ^ j];
                    const bool take_min = ((i & k) == 0) == ((i & j) == 0);
                    v[e] = take_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
                }
            } else {
                const int d = j / E;  // lane distance
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v[e], d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v[e] >> 32), d);
                    const uint64_t o = ((uint64_t)hi << 32) | lo;
                    const bool take_min = ((i & k) == 0) == ((i & j) == 0);
                    v[e] = take_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
                }
            }
        }
#pragma unroll
        for (int jj = E / 2; jj > 0; jj >>= 1) {  // partners in this thread
            if (jj < k) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & jj) == 0) {
                        const bool up = (((tid * E + e) & k) == 0);
                        const uint64_t a = v[e], b = v[e | jj];
                        if ((a > b) == up) {
                            v[e] = b;
                            v[e | jj] = a;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) s[tid * E + e] = v[e];
    __syncthreads();
}
// n: 256, 512, 1024 or 2048 keys; 256 threads
__device__ __forceinline__ void bitonic_sort_u64_256t(uint64_t* s, int n, int tid) {
    __syncthreads();
    if (n == 256)
        bitonic_sort_u64_regs<1>(s, tid);
    else if (n == 512)
        bitonic_sort_u64_regs<2>(s, tid);
    else if (n == 1024)
        bitonic_sort_u64_regs<4>(s, tid);
    else
        bitonic_sort_u64_regs<8>(s, tid);
}

// LDS-DMA: 16 bytes per lane from (scalar base + per-lane byte offset) to LDS at lds_addr + lane*16.
// Written as inline asm so that the compiler neither counts it nor drains it with a vmcnt(0) before
// unrelated LDS reads; the caller owns the s_waitcnt vmcnt(N) that retires it.  M0 (the DMA's LDS
// base) is compiler-reserved, hence saved and restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* sbase, int voff, uint32_t lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr)
        : "memory");
}
// N consecutive 1 KB pieces in one statement: piece j from (sbase + voff[j]) to lds_addr + j * 1024.  M0 is saved and
// restored once and stepped between the pieces (three scalar instructions per piece instead of six: with one wave per
// SIMD the DMA's issue is serial with the wave's MFMAs).
template <int N>
__device__ __forceinline__ void glds16_run_asm(const void* sbase, const int (&voff)[N], uint32_t lds_addr) {
    static_assert(N == 4 || N == 8, "written out for 4 and 8 pieces");
    unsigned keep;
    if constexpr (N == 8) {
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %9\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %9\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %9\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %9\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %9\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %9\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %9\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %9\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "v"(voff[7]), "s"(sbase), "s"(lds_addr)
            : "memory", "scc");
    } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase), "s"(lds_addr)
            : "memory", "scc");
    }
}
// The same for 4 bytes per lane: LDS at lds_addr + lane*4.
__device__ __forceinline__ void glds4_asm(const void* sbase, int voff, uint32_t lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr)
        : "memory");
}
// A pointer the compiler can prove wave-uniform (it is, by construction, at every call site).
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// Launchers implemented in the kernel files.
// Where the kernel that writes a query's final keys also writes them unpacked (either may be null): [B][k]
// distances and rows, +inf and -1 where fewer than k rows exist.
struct hrs_unpacked {
    float* dist = nullptr;
    int64_t* rows = nullptr;
};
int hrs_launch_exact_levels(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* level,
                            int64_t pitch, int levels, uint64_t* keys_out, hipStream_t st);
int hrs_launch_exact(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* mask,
                     const int32_t* qlist, const int32_t* nq_dev, uint64_t* keys_out, hipStream_t st,
                     const uint64_t* after = nullptr, int64_t mask_qstride = 0, int mask_eq = -1, int64_t out_stride = 0,
                     hrs_unpacked un = hrs_unpacked());
// phase: HRS_PHASE_ALL the whole call; HRS_PHASE_SCAN up to and including the main pass (B <= 1024);
// HRS_PHASE_TAIL the rest of a call whose scan phase ran with the same arguments
enum { HRS_PHASE_ALL = 0, HRS_PHASE_SCAN = 1, HRS_PHASE_TAIL = 2 };
int hrs_launch_mfma(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* mask,
                    uint64_t* keys_out, hipStream_t st, int phase = HRS_PHASE_ALL, hrs_unpacked un = hrs_unpacked());
int hrs_build_shadow(hrs_ctx* ctx, hrs_corpus* c, int kind, hipStream_t st);
int hrs_append_shadow(hrs_ctx* ctx, hrs_corpus* c, const float* C, int64_t N_new, hipStream_t st);
void hrs_free_shadow(hrs_corpus* c);
// Resolves, once per device and with that device current, where the static scratch of hrs_score_topk_static lives.
int hrs_exact_static_init(int device);
