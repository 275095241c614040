// Large-batch scoring on the matrix cores: an MFMA scan over a reduced-precision shadow of the corpus (int8 or
// bf16) with a threshold filter, then an exact fp64 re-score of the surviving candidates and a proof that the
// top-k is the exact one.
//
//   1. k_convert_q[_i8]  queries fp32 -> int8 / bf16 in MFMA operand order (+ squared norms, scales, error norms)
//   2. k_mfma_scan<1>    a sampled subset of corpus tiles: per (query, stream) running maximum
//   3. k_pick_tau        per query: tau = j-th largest stream maximum (a LOWER bound of the j-th largest sampled
//                        score), chosen so that a few hundred rows exceed it
//   4. k_mfma_scan<2>    a second sample: how many sampled scores exceed tau, in sixteen steps (for corpora whose
//                        rows come in runs of neighbours: the main pass raises tau where a list would overflow)
//   5. k_mfma_scan<0>    the whole corpus: every (query, row) with (an upper bound of the) score > tau goes to the
//                        query's candidate list (staged per wave in LDS, appended 64 at a time)
//   6. k_select_rescore  per query: sort the candidates by recorded score, re-score the best exactly (fp64, the
//                        oracle's operations), sort by (fp32 distance, row) and accept iff the k-th exact distance
//                        is provably below anything outside the re-scored set; otherwise the query is queued for
//   7. the exact scan    (hrs_exact.hip), driven by the device-side queue; no host round trip.
//
// Scan kernel geometry (per CU: one workgroup of 8 waves, two per SIMD): each wave keeps the B-operand fragments
// of 32 or 64 queries for the WHOLE depth in registers; 64-row corpus tiles stream through an LDS ring (3 to 5
// tiles deep) by LDS-DMA and are the A operand, shared by the 8 waves.  D(32x32) = A(corpus rows) x B(queries)
// puts the query on the lane, so the filter threshold is one VGPR.
#include <type_traits>

#include "hrs_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// The scan runs on one of two shadows of the corpus.  bf16: v_mfma_f32_32x32x16_bf16, 16 depth
// elements per k-step.  int8: v_mfma_i32_32x32x32_i8, 32 depth elements per k-step, twice the rate and
// half the bytes; each 32-row group g carries one fp32 scale s_g (row value = s_g * (int8 + error)), queries one
// scale each.  With q = sq (qi + eq), c = s_g (ci + ec) -- eq, ec the quantisation residuals, known when the
// vectors are quantised -- and the Cauchy-Schwarz inequality on the three cross terms:
//     q.c / sq  <=  s_g * (qi.ci + |qi| E_g + |eq| (C_g + E_g)),   E_g = max_rows |ec|,  C_g = max_rows |ci|
// (Euclidean norms, rounded up) and the int8 kernel filters, records and ranks by that UPPER BOUND of the scaled
// score, so a row it does not record is provably below the threshold and no further error term enters the
// acceptance test.  (The bound from the residuals' maximum, e |qi|_1 + e |ci|_1 + e^2 D with e = 0.5, is 1.4 x
// looser: 0.48 instead of 0.35 standard deviations of the scores of random unit rows at depth 384, 0.74 instead of
// 0.54 at depth 1024, where it made the int8 shadow useless -- thirty recorded rows for one above the threshold.)
// Either way a fragment is 16 bytes per lane and a row is KS * 32 bytes, so the addressing is shared.
template <bool I8> struct scan_types { typedef bf16x8 frag; typedef f32x16 acc; };
template <> struct scan_types<true> { typedef i32x4 frag; typedef i32x16 acc; };

#ifndef HRS_SCAN_PF
#define HRS_SCAN_PF 4
#endif
// Depth of the LDS ring of corpus tiles (one less is in flight): as deep as 128 KB allows, at most 5.  A
// small batch is bound by the bytes in flight per CU (at depth 384, int8: two 24 KB tiles sustain 5.3 TB/s, four 5.5 TB/s).
#ifndef HRS_SS_MAX
#define HRS_SS_MAX 32   // sampled passes: at most one tile in this many
#define HRS_J_MIN 16    // tau is at least the j-th largest stream maximum
#endif
#ifndef HRS_SCAN_RING_BYTES
#define HRS_SCAN_RING_BYTES (128 * 1024)
#endif
// The kernels with two query tiles per wave serve large batches, which the matrix cores bound.  There the
// workgroup barrier at every tile boundary costs a tenth of the time (eight waves wait for the slowest), so
// these kernels synchronise at every second boundary instead: four buffers, two tiles requested at a time.
// 96 KB at depth 384.  (A re-score workgroup of the previous batch does not share the CU either way: two scan
// waves per SIMD leave it too few registers.)
constexpr bool scan_pairs(int qt) { return qt == 2; }
// Rows above the threshold are staged in LDS, every wave in a region of its own with the fill level in a scalar
// register (slots come from a ballot and a lane count: no atomic, no wait), and go to the per-query lists in
// memory 64 at a time when the region is full.  Appending straight to memory makes every recording wave wait for
// its atomic's return, i.e. for vmcnt(0): that drains the wave's share of the LDS-DMA ring and, through the tile
// barrier, stalls all eight waves; over a small corpus (1M rows: a row above the threshold in every other 32 x 32
// tile) that was half of the pass.
constexpr int scan_stage(bool pair) { return pair ? 256 : 1024; }  // staged entries (12 bytes each)
constexpr int scan_nbuf(int tile_bytes, int qt) {  // qt == 2 here means "the pair-wise schedule"
    return scan_pairs(qt) ? 4 : (tile_bytes * 5 <= HRS_SCAN_RING_BYTES ? 5 : (tile_bytes * 4 <= HRS_SCAN_RING_BYTES ? 4 : 3));
}
constexpr size_t scan_lds_bytes(int tile_bytes, bool pair) {
    return (size_t)scan_nbuf(tile_bytes, pair ? 2 : 1) * tile_bytes + (size_t)scan_stage(pair) * 12;
}
#define MF_QPB 256   // queries per workgroup and 32-query tile of a wave (QT tiles per wave)
// The per-query list lengths sit one to a 128-byte line: atomics on one line are served one after the other, and
// with sixteen or thirty-two counters to a line the final flush of a small-corpus pass (every wave appends at the
// same moment) queued for longer than it took to stream a sixth of the corpus.
#define MF_CNT_STRIDE 32
#define MF_EPS_BASE 0.0039101
#define MF_EPS_PER_D 2.4e-7

// The converted queries are stored fragment-major: the 16-byte pieces that the 64 lanes of a wave load as one
// operand fragment (32 queries x the two halves of a 32-byte k-step) are contiguous, 1 KB per (32 queries, k-step).
// Row-major, a fragment load touched 32 cache lines for 32 bytes each, a wave's 24 to 64 KB of query rows thrashed
// the 16 KB vector cache between k-steps, and all 256 workgroups asked the same L2 channels for the same lines at
// the same moment: the start of every pass took 5 us for that alone.
__device__ __forceinline__ size_t qfrag_offset(int q, int kstep, int half, int ksteps) {
    return ((size_t)(q >> 5) * ksteps + kstep) * 1024 + half * 512 + (q & 31) * 16;
}

struct MfmaArgs {
    const char* shadow;    // [Npad][KS * 32 bytes]
    const char* qb;        // [Bpad / 32][KS][2][32][16 bytes], see qfrag_offset
    const float4* group;   // int8 only, per 32-row group: (s_g, E_g, C_g + E_g, -) for PRE, (1 / s_g, ...) for MAIN
    const float2* kq;      // int8 only, per query: (|qi|, |eq|)
    int64_t N;
    int64_t tiles;         // ceil(N / TR)
    int64_t row_offset;
    int nqg;               // query groups of 256 (1, 2 or 4); nslices = 256 / nqg
    int nq;                // queries of this chunk; waves whose queries are all padding only feed the LDS ring
    int ss;                // sampling stride of the sampled passes (tiles)
    const float* lad_inv;  // ladder pass: per query 16 / (largest sampled score - tau)
    int* hist;             // ladder pass: [Bpad][16] sampled scores above tau, by sixteenth of that range
    const uint8_t* mask;   // nullable, [N]: rows with 0 are never recorded (the `where` filter)
    float* maxima;         // [Bpad][HRS_MAX_STREAMS]
    const float* tau;      // [Bpad] the first threshold (from the sampled maxima)
    float* tau_out;        // [Bpad] MODE 0: the threshold the pass filtered by (raised where the ladder says so)
    int raise_ss;          // MODE 0: sampling stride of the ladder pass
    int raise_limit;       // MODE 0: estimated rows above the threshold that a list may hold
    long long* work;       // nullable: tiles scored by the main pass (k_mfma_scan16 adds; others leave it at 0 = all)
    int ladder_records;    // the ladder pass records the rows above tau of its tiles, and the main pass (k_mfma_scan16,
                           // which knows how to) leaves those tiles out: see ladder_tile
    int* cnt;              // [Bpad * MF_CNT_STRIDE], one counter per 128-byte line
    uint64_t* cand;        // [Bpad][HRS_CAND_CAP_BIG]
#ifdef HRS_SCAN_PROF
    long long* prof;
    float prof_tau_shift;  // HRS_PROF_TAU_SHIFT: added to the main pass's thresholds (a large value: a pass without hits)
#endif
};

// Rows of a topically ordered corpus do not score independently: whole runs of neighbours exceed the first
// threshold together and the candidate lists would overflow.  The ladder pass has counted the sampled scores above
// it in sixteen steps up to the largest sampled score; the threshold of the main pass is the first one moved up to
// the lowest step above which the estimated number of rows (sampled count x sampling stride) fits a list.  It
// never moves down.  Every workgroup of the main pass evaluates this for its own queries (the inputs are final
// before the pass starts), which saves a launch between the ladder pass and the main pass.
__device__ __forceinline__ float raised_tau(const MfmaArgs& a, int q) {
    float t = a.tau[q];
    const float li = a.lad_inv[q];
    // all sixteen counts in one go (loads that depend on a loop's exit test would come one memory latency apiece)
    const int4* hp = (const int4*)(a.hist + (size_t)q * 16);
    const int4 h3 = hp[3], h2 = hp[2], h1 = hp[1], h0 = hp[0];
    const int hv[16] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
    long long above = 0;
    int step = 0;  // rows from step b upward are too many: start one step higher
#pragma unroll
    for (int b = 15; b >= 0; --b) {
        above += hv[b];
        if (step == 0 && above * a.raise_ss > a.raise_limit) step = b + 1;
    }
    if (step > 15) step = 15;
    if (li > 0.f && step > 0) t += (float)step / li;
#ifdef HRS_SCAN_PROF
    t += a.prof_tau_shift;
#endif
    return t;
}

__device__ __forceinline__ uint32_t hash32(uint32_t x);
// The tile that a sampled pass of stride ss takes from block n of a slice (blocks of ss consecutive tiles from t_begin).
// The ladder pass and the main pass that leaves the ladder's tiles out must agree on it.
__device__ __forceinline__ int64_t sampled_tile(int64_t t_begin, int slice, int64_t n, int ss) {
    return t_begin + n * ss + (int64_t)(hash32((uint32_t)(slice * 0x9E3779B1u + n)) % (uint32_t)ss);
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// MODE 0: the whole corpus, rows above tau are recorded.  MODE 1: sampled tiles, per (query, stream)
// running maximum.  MODE 2: sampled tiles, per query a 16-bin count of the scores above tau (the "ladder").
// RS ("row split", batches of up to 256 queries on the kernels with two query tiles per wave): the eight waves are
// four query groups of 64 x the two 32-row halves of a tile, instead of eight query groups of 32 that each read the
// whole tile -- every corpus fragment read from LDS then feeds two MFMAs and a tile is read from LDS four times
// instead of eight.  With one tile per wave the kernel is bound by LDS reads (8 waves x 24 KB per tile is all the
// 128 B/clk an LDS delivers in the time the matrix cores need for the tile).
template <int KS, int TR, int MODE, int QT, bool I8, bool RS = false>
__global__ __launch_bounds__(512) void k_mfma_scan(MfmaArgs a) {
    constexpr bool PRE = MODE != 0;      // sampled pass
    constexpr bool MAXIMA = MODE == 1, LADDER = MODE == 2;
    typedef typename scan_types<I8>::frag frag_t;
    typedef typename scan_types<I8>::acc acc_t;
    constexpr int MF_QPW = 32 * QT;           // queries per wave
    constexpr int NW = 8;                     // waves per workgroup (256 * QT queries)
    constexpr int RB = KS * 32;               // row bytes
    constexpr int TILE_BYTES = TR * RB;
    constexpr bool PAIR = scan_pairs(QT) && !RS;  // barrier at every second tile boundary
    constexpr int NBUF = scan_nbuf(TILE_BYTES, PAIR ? 2 : 1);
    constexpr int IPW = TR * KS / (32 * NW);  // LDS-DMA instructions per wave per tile
    constexpr int NSUB = TR / 32;             // 32-row groups per tile (one int8 scale each)
    constexpr int WSUB = RS ? 1 : NSUB;       // ... of which a wave scores this many
    static_assert(!RS || (QT == 2 && NSUB == 2), "row split: two query tiles per wave, two row halves per tile");
    constexpr int STG = scan_stage(PAIR);
    constexpr int PPR = 2 * KS;               // 16-byte pieces per row
    // Bank-conflict-free row reads need an XOR permutation of the pieces of a row that depends on the
    // row: 4 bits (r & 15) when the row is a multiple of 256 bytes, else 3 bits ((r >> 1) & 7) for rows
    // that are an odd multiple of 128 bytes (adjacent rows already differ by half the bank window).
    constexpr int SWB = (PPR % 16 == 0) ? 4 : 3;
    constexpr int NB = 1 << (SWB - 1);        // per-lane fragment base addresses
    constexpr int FSTEP = 16 << SWB;          // bytes between fragments NB k-steps apart
    static_assert(PPR % 8 == 0, "row bytes must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int qg = loc % a.nqg;
    const int nslices = HRS_NUM_CU / a.nqg;
    const int slice = xcd * (32 / a.nqg) + loc / a.nqg;
    const int sb0 = RS ? (wv >> 2) : 0;       // first 32-row group of a tile that this wave scores
    const int q0 = RS ? qg * MF_QPB + (wv & 3) * MF_QPW : qg * (MF_QPB * QT) + wv * MF_QPW;

#ifdef HRS_SCAN_PROF
    const long long pt0 = wall_clock64();
    long long pt1 = pt0, pt2 = pt0, pt3 = pt0;
#endif
    // The queries of this wave: fragments, thresholds and error terms.  Requested right behind the first corpus tile
    // (the requests are independent, and a pass over a small corpus is short enough for a serial start to show).
    frag_t qf[QT][KS];
    float tau[QT], taum[QT], linv[QT];
    float2 kq[QT];  // int8: (|qi|, |eq|) of the lane's query
    float mx[QT][WSUB];
    auto load_queries = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                qf[qt][s] = *(const frag_t*)(a.qb + qfrag_offset(q0 + qt * 32 + i, s, h, KS));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            tau[qt] = MAXIMA ? 0.f : (LADDER ? a.tau[q0 + qt * 32 + i] : raised_tau(a, q0 + qt * 32 + i));
            if (MODE == 0 && slice == 0 && sb0 == 0 && h == 0) a.tau_out[q0 + qt * 32 + i] = tau[qt];  // one writer per query
            kq[qt] = I8 ? a.kq[q0 + qt * 32 + i] : make_float2(0.f, 0.f);
            linv[qt] = LADDER ? a.lad_inv[q0 + qt * 32 + i] : 0.f;
#pragma unroll
            for (int sb = 0; sb < WSUB; ++sb) mx[qt][sb] = -__builtin_inff();
        }
        // every query fragment must have landed before the loop: the compiler would otherwise place its
        // partial vmcnt waits at their first use inside the loop, where they would drain the LDS-DMA ring
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qf[qt][s]));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) asm volatile("" : "+v"(tau[qt]), "+v"(kq[qt].x), "+v"(kq[qt].y), "+v"(linv[qt]));
        // int8 main pass: the threshold lowered by more than the roundings of the per-block compare can add up to
        // (the reciprocal of the group scale, one product, three fused operations: under 2.5e-7 in all)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) taum[qt] = tau[qt] - fabsf(tau[qt]) * 4e-7f;
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store above too: the loop's waits count requests
    };

    const int64_t t_begin = a.tiles * slice / nslices;
    const int64_t t_end = a.tiles * (slice + 1) / nslices;
    const int64_t count = PRE ? (t_end - t_begin) / a.ss : (t_end - t_begin);
    const int64_t full_tiles = a.N / TR;  // tiles without padding rows

    auto tile_at = [&](int64_t n) -> int64_t {
        if (n >= count) n = count - 1;
        if (!PRE) return t_begin + n;
        int64_t t = sampled_tile(t_begin, slice, n, a.ss);
        if (t >= full_tiles) t = full_tiles - 1;
        return t < 0 ? 0 : t;
    };
    // ladder pass that records: a block whose tile had to be moved (the corpus' last, partial tile) records nothing --
    // the main pass does not leave that block's tile out either
    auto records_tile = [&](int64_t n) -> bool { return LADDER && a.ladder_records && n < count && sampled_tile(t_begin, slice, n, a.ss) < full_tiles; };
    // per-lane source offsets of this wave's DMA pieces (bytes from the tile base): piece p of row r
    // lands at LDS position p ^ (r & 15) of that row (the read side applies the same XOR)
    int srcoff[IPW];
#pragma unroll
    for (int ii = 0; ii < IPW; ++ii) {
        const int g = (wv * IPW + ii) * 64 + lane;
        const int r = g / PPR, p = g % PPR;
        srcoff[ii] = r * RB + ((p ^ (SWB == 4 ? (r & 15) : ((r >> 1) & 7))) << 4);
    }
    const uint32_t lds0 = lds_addr_of(smem);
    auto issue = [&](int64_t n, int buf) {
        const char* base = uniform_ptr(a.shadow + tile_at(n) * (int64_t)TILE_BYTES);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + buf * TILE_BYTES + wv * (IPW * 1024));
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii) glds16_asm(base, srcoff[ii], dst + ii * 1024);
    };
    // read side: k-step s of row i wants piece 2s+h, stored at (2s+h) ^ w(i).  With v = h ^ w the low
    // SWB bits are ((2s) & mask) ^ v: NB per-lane bases, everything else is an immediate.
    int bc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) bc[c] = (int)lds0 + (sb0 * 32 + i) * RB + (((2 * c) ^ (h ^ (SWB == 4 ? (i & 15) : ((i >> 1) & 7)))) << 4);

    // staged candidates (MODE 0): this wave's region of keys and of their queries
    constexpr int WCAP = STG / NW;
    uint64_t* const wk = (uint64_t*)(smem + NBUF * TILE_BYTES) + wv * WCAP;
    uint32_t* const wq = (uint32_t*)((uint64_t*)(smem + NBUF * TILE_BYTES) + STG) + wv * WCAP;
    int wcnt = 0;  // entries staged (wave-uniform)
    auto to_memory = [&](uint64_t key, int q) __attribute__((always_inline)) {
        if (!a.mask || a.mask[(int64_t)(uint32_t)key - a.row_offset]) {  // the `where` filter
            const int pos = atomicAdd(&a.cnt[q * MF_CNT_STRIDE], 1);
            if (pos < HRS_CAND_CAP_BIG) a.cand[(size_t)q * HRS_CAND_CAP_BIG + pos] = key;
        }
    };
    auto flush_wave = [&]() __attribute__((always_inline)) {
        for (int e = lane; e < wcnt; e += 64) to_memory(wk[e], (int)wq[e]);
        wcnt = 0;
    };
    // Lanes with `hit` record (key, q); every lane of the wave calls this (the ballot is the slot allocator).
    auto record = [&](bool hit, uint64_t key, int q) __attribute__((always_inline)) {
        const uint64_t bal = __builtin_amdgcn_ballot_w64(hit);
        if (bal == 0) return;
        const int np = __builtin_popcountll(bal);
        if (wcnt + np > WCAP) flush_wave();
        const int slot = wcnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (hit) {
            if (slot < WCAP) {
                wk[slot] = key;
                wq[slot] = (uint32_t)q;
            } else {
                to_memory(key, q);  // more hits in one instruction than the region holds
            }
        }
        wcnt = __builtin_amdgcn_readfirstlane(min(wcnt + np, WCAP));
    };

    if (count > 0) {
        typedef const __attribute__((address_space(3))) char* lds_cptr;
        typedef const __attribute__((address_space(3))) frag_t* lds_frag;
        constexpr int PF = HRS_SCAN_PF < KS ? HRS_SCAN_PF : KS;  // A fragments in flight
        // int8: the group terms of a tile come by scalar load (its counter is lgkmcnt, so the LDS-DMA ring is left
        // alone).  The load is a hand-written statement at the top of the tile and its wait a second one behind the
        // tile's first block of MFMAs, where the wave has no LDS read outstanding and the terms have long arrived: a load
        // that the compiler tracks makes it wait for lgkmcnt(0) at the next LDS read (scalar loads return out of order),
        // i.e. for the latency of the scalar load once per tile, in the middle of the MFMA chain.  The compiler does
        // not know that the destination registers are pending between the two statements: the registers are one
        // variable that only the wait statement reads (tied), with no value carried around the loop, and a test
        // reads the generated code for any other access in between.
        typedef float f32xg __attribute__((ext_vector_type(4 * NSUB)));
        auto terms_request = [&](int64_t tile, f32xg& dst) __attribute__((always_inline)) {
            const char* p = uniform_ptr(a.group + tile * NSUB);
            if constexpr (NSUB == 2)
                asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(dst) : "s"(p));
            else
                asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(dst) : "s"(p));
        };
#ifdef HRS_SCAN_PROF
        pt1 = wall_clock64();
#endif
        // first tile, queries, rest of the ring: requests come back in order, and behind the whole ring (120 KB per
        // CU, 30 MB over the chip) the queries would arrive 5 us later than behind one tile
        issue(0, 0);
        load_queries();
#pragma unroll
        for (int g = 1; g < NBUF; ++g) issue(g, g);
        // the first tile (PAIR: the first two) has landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PAIR ? NBUF - 2 : NBUF - 1) * IPW) : "memory");
        __builtin_amdgcn_s_barrier();
#ifdef HRS_SCAN_PROF
        pt2 = wall_clock64();
#endif
        int buf = 0;
        if (q0 >= a.nq) {
            // padding queries only (a small batch): this wave keeps its share of the LDS-DMA ring going and
            // meets the same barriers, and leaves the matrix cores to the waves that hold queries
            for (int64_t n = 0; n + 1 < count; ++n) {
                if (!PAIR) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
                    __builtin_amdgcn_s_barrier();
                    issue(n + NBUF, buf);
                } else if (n & 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    issue(n + 3, (buf + 3) & 3);
                    issue(n + 4, buf);
                }
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
        } else {
        lds_cptr tb[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) tb[c] = (lds_cptr)(size_t)(uint32_t)bc[c];
        // the first PF fragments of the next 32-row subtile are requested before the current subtile's
        // epilogue (and, at a tile boundary, right after the barrier), so their LDS latency is covered
        frag_t pre[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) pre[s] = *(lds_frag)(tb[s & (NB - 1)] + (s / NB) * FSTEP);
        for (int64_t n = 0; n < count; ++n) {
            const int64_t tile = tile_at(n);
            const bool rec = records_tile(n);
            f32xg terms;
            if constexpr (I8) terms_request(tile, terms);
            f32x4 scw[WSUB];  // the group terms of the 32-row groups this wave scores
#pragma unroll
            for (int sb = 0; sb < WSUB; ++sb) {
                acc_t acc[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[qt][r] = 0;
                frag_t av[KS];
#pragma unroll
                for (int s = 0; s < PF; ++s) av[s] = pre[s];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s + PF < KS) av[s + PF] = *(lds_frag)(tb[(s + PF) & (NB - 1)] + sb * 32 * RB + ((s + PF) / NB) * FSTEP);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        if constexpr (I8)
                            acc[qt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s], qf[qt][s], acc[qt], 0, 0, 0);
                        else
                            acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s], qf[qt][s], acc[qt], 0, 0, 0);
                    }
                }
                // schedule: one MFMA per further LDS read, then the last PF MFMAs
#pragma unroll
                for (int s = 0; s < KS - PF; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, QT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, QT * PF, 0);
                if constexpr (I8) {
                    if (sb == 0) {
                        __builtin_amdgcn_sched_barrier(0);  // (the wait stays behind the block's last MFMA)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(terms)::"memory");
                        if constexpr (NSUB == 2) {
                            const f32x4 lo = __builtin_shufflevector(terms, terms, 0, 1, 2, 3);
                            const f32x4 hi = __builtin_shufflevector(terms, terms, 4, 5, 6, 7);
                            if constexpr (RS) {
                                scw[0] = sb0 ? hi : lo;
                            } else {
                                scw[0] = lo;
                                scw[WSUB - 1] = hi;
                            }
                        } else {
                            scw[0] = terms;
                        }
                    }
                }
                auto next_tile = [&]() __attribute__((always_inline)) {
                    if (!PAIR) {
                        // tile boundary: tile n+1 has landed once the younger tiles alone are outstanding
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
                        __builtin_amdgcn_s_barrier();
                        issue(n + NBUF, buf);  // tile n's buffer is free: every wave has consumed its fragments
                    } else if (n & 1) {
                        // end of a pair: tiles n+1 and n+2, requested two tiles ago, have landed; behind the barrier
                        // the buffers of tiles n-1 and n are free for tiles n+3 and n+4.  Between the tiles of a
                        // pair nothing is awaited: both were complete at the last barrier.
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        issue(n + 3, (buf + 3) & 3);
                        issue(n + 4, buf);
                    }
                    buf = buf + 1 == NBUF ? 0 : buf + 1;
#pragma unroll
                    for (int c = 0; c < NB; ++c) tb[c] = (lds_cptr)(size_t)(uint32_t)(bc[c] + buf * TILE_BYTES);
#pragma unroll
                    for (int s = 0; s < PF; ++s) pre[s] = *(lds_frag)(tb[s & (NB - 1)] + (s / NB) * FSTEP);
                };
                // Main pass: the two waves of a SIMD would leave every barrier in step -- both in their MFMAs, then both
                // in the vector work on their results, the matrix pipe idle.  Waves 0..3 look at the last group's results
                // in front of the barrier, waves 4..7 behind it, which sets the partners half a phase apart.
                const bool early = MODE == 0 && sb + 1 == WSUB && n + 1 < count && (!PAIR || (n & 1)) && wv < 4;
                if (sb + 1 < WSUB) {
#pragma unroll
                    for (int s = 0; s < PF; ++s) pre[s] = *(lds_frag)(tb[s & (NB - 1)] + (sb + 1) * 32 * RB + (s / NB) * FSTEP);
                } else if (n + 1 < count && !early) {
                    next_tile();
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    // every lane calls this for every r of a 32 x 32 tile that holds a row above the threshold; all but
                    // the compare and the ballot happen only for an r with such a row
                    auto append = [&](bool hit, int r, auto&& score) __attribute__((always_inline)) {
                        if (__builtin_amdgcn_ballot_w64(hit) == 0) return;
                        const uint32_t row = (uint32_t)tile * TR + (sb0 + sb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        record(hit && row < (uint32_t)a.N, ((uint64_t)(~f32_orderable(score())) << 32) | (uint32_t)(row + (uint32_t)a.row_offset), q0 + qt * 32 + i);
                    };
                    if constexpr (I8) {
#ifdef HRS_SCAN_NOEPI  // timing only: a main pass that never looks at its accumulators (no results)
                        if (MODE == 0) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[qt][r]));
                            continue;
                        }
#endif
                        // the maximum of the sixteen rows as a chain of three-input maxima (eight instructions; a tree over
                        // groups of four is ten); a tile with a row above the threshold is then searched group by group
                        int m = max(max(acc[qt][0], acc[qt][1]), acc[qt][2]);
#pragma unroll
                        for (int r = 3; r < 15; r += 2) m = max(max(m, acc[qt][r]), acc[qt][r + 1]);
                        m = max(m, acc[qt][15]);
                        // what the three cross terms of the quantisation residuals can add: |qi| E_g + |eq| (C_g + E_g)
                        // (the norms are rounded up by more than these two roundings can take away)
                        const float xb = fmaf(kq[qt].x, scw[sb].y, kq[qt].y * scw[sb].z);
                        if (MAXIMA) {
                            mx[qt][sb] = fmaxf(mx[qt][sb], ((float)m + xb) * scw[sb].x);  // .x = s_g
                        } else if (LADDER) {
                            // (wave-uniform test when the rows are also recorded: the slot allocator is a ballot)
                            const bool some = ((float)m + xb) * scw[sb].x > tau[qt];
                            if (rec ? __builtin_amdgcn_ballot_w64(some) != 0 : some) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float v = ((float)acc[qt][r] + xb) * scw[sb].x;
                                    if (v > tau[qt]) atomicAdd(&a.hist[(q0 + qt * 32 + i) * 16 + max(0, min(15, (int)((v - tau[qt]) * linv[qt])))], 1);
                                    if (rec) append(v > tau[qt], r, [&]() { return v; });
                                }
                            }
                        } else {
                            // s_g (acc + xb) > tau  <=>  acc > tau / s_g - xb.  Three fused operations on the lowered
                            // threshold and a converting floor (saturating: a threshold out of reach stays one); the
                            // half is slack for their absolute rounding, the lowering for the relative one, so that
                            // no qualifying row is dropped.  .x = 1 / s_g
                            float tf = fmaf(taum[qt], scw[sb].x, -0.5f);
                            tf = fmaf(-kq[qt].x, scw[sb].y, tf);
                            tf = fmaf(-kq[qt].y, scw[sb].z, tf);
                            int ti;
                            asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ti) : "v"(tf));
                            if (__builtin_amdgcn_ballot_w64(m > ti) != 0) {  // wave-uniform: some lane holds such a row
                                const float gs = scw[sb].w;  // s_g itself (.x is its reciprocal)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const int m4 = max(max(acc[qt][4 * g], acc[qt][4 * g + 1]), max(acc[qt][4 * g + 2], acc[qt][4 * g + 3]));
                                    if (__builtin_amdgcn_ballot_w64(m4 > ti) == 0) continue;
#pragma unroll
                                    for (int r = 4 * g; r < 4 * g + 4; ++r) append(acc[qt][r] > ti, r, [&]() { return ((float)acc[qt][r] + xb) * gs; });
                                }
                            }
                        }
                    } else {
                        float m = acc[qt][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[qt][r]);
                        if (MAXIMA) {
                            mx[qt][sb] = fmaxf(mx[qt][sb], m);
                        } else if (LADDER) {
                            if (m > tau[qt]) {
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    if (acc[qt][r] > tau[qt])
                                        atomicAdd(&a.hist[(q0 + qt * 32 + i) * 16 + max(0, min(15, (int)((acc[qt][r] - tau[qt]) * linv[qt])))], 1);
                            }
                        } else if (__builtin_amdgcn_ballot_w64(m > tau[qt]) != 0) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) append(acc[qt][r] > tau[qt], r, [&]() { return acc[qt][r]; });
                        }
                    }
                }
                if (early) next_tile();
            }
        }
        }  // waves with queries
#ifdef HRS_SCAN_PROF
        pt3 = wall_clock64();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 0 || (LADDER && a.ladder_records)) flush_wave();  // what is still staged
    } else {
        load_queries();  // (for the threshold that the first slice publishes)
    }
#ifdef HRS_SCAN_PROF
    if (MODE == 0 && lane == 0 && wv == 0) {
        unsigned long long* st = (unsigned long long*)a.prof;
        atomicAdd(&st[4], (unsigned long long)(pt1 - pt0));
        atomicAdd(&st[5], (unsigned long long)(pt2 - pt1));
        atomicAdd(&st[6], (unsigned long long)(pt3 - pt2));
        atomicAdd(&st[7], (unsigned long long)(wall_clock64() - pt3));
        atomicMin(&st[8], (unsigned long long)pt0);
        atomicMax(&st[9], (unsigned long long)pt0);
        atomicMin(&st[10], (unsigned long long)pt3);
        atomicMax(&st[11], (unsigned long long)wall_clock64());
    }
#endif
    if (MAXIMA) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int sb = 0; sb < WSUB; ++sb)
                a.maxima[(size_t)(q0 + qt * 32 + i) * HRS_MAX_STREAMS + (slice * 2 + h) * NSUB + sb0 + sb] = mx[qt][sb];
    }
}



// ---- the int8 main pass on v_mfma_i32_16x16x64_i8 ---------------------------------------------------------------
// Same work, geometry and results as k_mfma_scan<KS, 64, 0, 2, true> (eight waves, 64 queries per wave, tiles of 64
// rows through the same LDS image, barriers at every second tile boundary), on the 16 x 16 x 64 shape: per 32-row
// group a wave issues 2 row halves x 4 query groups x KS/2 k-steps MFMAs of half the cycles each, and every corpus
// fragment read from LDS feeds four of them.  The matrix pipe takes the same cycles either way, but under load the chip
// holds a higher clock on this shape: an LDS-fed loop on random int8 operands ran 9 to 12 % faster with it
// (tools/micro/mfma_micro.hip).  A lane of the 16 x 16 result holds four rows of ONE of sixteen queries, so a lane
// carries the thresholds of four queries (one per query group) instead of two.
// Query fragments: [Bpad / 16][KS / 2][4 depth quarters][16 queries][16 bytes] (k_convert_q_i8 writes them beside the
// 32 x 32 form, which the sampled passes keep using).
__device__ __forceinline__ size_t qfrag16_offset(int q, int c, int k64) {  // byte c of query q's padded int8 row
    return ((size_t)(q >> 4) * k64 + (c >> 6)) * 1024 + ((c >> 4) & 3) * 256 + (q & 15) * 16 + (c & 15);
}

// QG: 16-query operands per wave (4 up to depth 384, 2 beyond: the fragments of 32 queries x 1024 depth are 128
// registers); TR: rows per tile (64, or 32 where four tiles of 64 rows would not fit the LDS).
template <int KS, int QG, int TR>
__global__ __launch_bounds__(512) void k_mfma_scan16(MfmaArgs a, const char* __restrict__ qb16) {
    constexpr int NW = 8, K64 = KS / 2;
    constexpr int RB = KS * 32, TILE_BYTES = TR * RB, NBUF = 4, IPW = TR * KS / (32 * NW), NSUB = TR / 32;
    constexpr int STG = scan_stage(true), PPR = 2 * KS;
    constexpr int SWB = (PPR % 16 == 0) ? 4 : 3;
    constexpr int NB = 1 << (SWB - 2);   // per-lane fragment bases: k-steps NB apart differ by an immediate
    constexpr int FSTEP = 16 << SWB;
    constexpr int NF = 2 * K64;          // fragments per 32-row group: (k-step, row half)
    static_assert(KS % 2 == 0 && PPR % 8 == 0 && (TR == 32 || TR == 64) && QG * K64 * 4 <= 128 && NBUF * TILE_BYTES <= 128 * 1024,
                  "depth in steps of 64; query fragments within 128 registers; four tiles within 128 KB");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kg = lane >> 4;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int qg = loc % a.nqg;
    const int nslices = HRS_NUM_CU / a.nqg;
    const int slice = xcd * (32 / a.nqg) + loc / a.nqg;
    const int q0 = qg * (NW * QG * 16) + wv * (QG * 16);

    i32x4 qf[QG][K64];
    float taum[QG];
    float2 kq[QG];
    auto load_queries = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int s = 0; s < K64; ++s) qf[g][s] = *(const i32x4*)(qb16 + ((size_t)((q0 >> 4) + g) * K64 + s) * 1024 + lane * 16);
        float tau[QG];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            tau[g] = raised_tau(a, q0 + g * 16 + j);
            if (slice == 0 && kg == 0) a.tau_out[q0 + g * 16 + j] = tau[g];  // one writer per query
            kq[g] = a.kq[q0 + g * 16 + j];
        }
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int s = 0; s < K64; ++s) asm volatile("" : "+v"(qf[g][s]));
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            asm volatile("" : "+v"(tau[g]), "+v"(kq[g].x), "+v"(kq[g].y));
            taum[g] = tau[g] - fabsf(tau[g]) * 4e-7f;  // (as in k_mfma_scan: lowered by more than the compare's roundings)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store above too: the loop's waits count requests
    };

    const int64_t t_begin = a.tiles * slice / nslices;
    const int64_t t_end = a.tiles * (slice + 1) / nslices;
    // Where the ladder pass has recorded the rows of its own tiles (a.ladder_records), those tiles are left out here:
    // one per block of raise_ss tiles, at the position sampled_tile gives -- unless that position is the corpus' last,
    // partial tile, which the ladder pass moved away from and did not record.  The tiles of this slice are walked by
    // two cursors of the same kind, one for the DMA requests and one, a few tiles behind, for the tile being scored.
    const int ss2 = a.raise_ss;
    const int64_t full_tiles = a.N / TR;
    const int64_t nblk = a.ladder_records ? (t_end - t_begin) / ss2 : 0;  // blocks the ladder pass took a tile from
    struct Cursor {
        int64_t next;     // the tile the next call returns (t_end: none left)
        int64_t blk;      // its block
        int64_t blk_end;  // first tile behind that block
        int64_t skip;     // the tile left out of that block (-1: none)
    };
    auto skip_of = [&](int64_t blk) -> int64_t {
        if (blk >= nblk) return -1;
        const int64_t t = sampled_tile(t_begin, slice, blk, ss2);
        return t < full_tiles ? t : -1;
    };
    auto cursor_settle = [&](Cursor& c) {  // moves `next` off a tile that is left out
        while (c.next < t_end) {
            if (c.next >= c.blk_end) {  // (no division here: this runs once per tile, on the scalar unit)
                ++c.blk;
                c.blk_end += ss2;
                c.skip = skip_of(c.blk);
            }
            if (c.next != c.skip) break;
            ++c.next;
        }
    };
    auto cursor_init = [&](Cursor& c) {
        c.next = t_begin;
        c.blk = 0;
        c.blk_end = t_begin + ss2;
        c.skip = skip_of(0);
        cursor_settle(c);
    };
    int64_t last_tile = t_begin;  // (what the request cursor returns once the slice is used up: requests stay uniform)
    auto cursor_take = [&](Cursor& c) -> int64_t {
        if (c.next >= t_end) return last_tile;
        const int64_t t = c.next++;
        cursor_settle(c);
        return t;
    };
    // tiles of this slice that this pass scores: all but one per block -- only the block that holds the corpus' partial
    // last tile may have kept all of its own
    int64_t count = (t_end - t_begin) - nblk;
    if (nblk && full_tiles >= t_begin && full_tiles < t_begin + nblk * ss2 && skip_of((full_tiles - t_begin) / ss2) < 0) ++count;
    Cursor cur_dma, cur_run;
    cursor_init(cur_dma);
    cursor_init(cur_run);
#ifndef HRS_SCAN_PROF
    // (for hrs_scan_work: how many of the corpus' tiles this pass scores -- one workgroup per slice reports)
    if (qg == 0 && threadIdx.x == 0 && a.work) atomicAdd((unsigned long long*)a.work, (unsigned long long)count);
#endif
    int srcoff[IPW];
#pragma unroll
    for (int ii = 0; ii < IPW; ++ii) {
        const int g = (wv * IPW + ii) * 64 + lane;
        const int r = g / PPR, p = g % PPR;
        srcoff[ii] = r * RB + ((p ^ (SWB == 4 ? (r & 15) : ((r >> 1) & 7))) << 4);
    }
    const uint32_t lds0 = lds_addr_of(smem);
    auto issue = [&](int64_t, int buf) {  // requests come in the order of the tiles: the cursor knows which is next
        const int64_t t = cursor_take(cur_dma);
        last_tile = t;
        const char* base = uniform_ptr(a.shadow + t * (int64_t)TILE_BYTES);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + buf * TILE_BYTES + wv * (IPW * 1024));
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii) glds16_asm(base, srcoff[ii], dst + ii * 1024);
    };
    // read side: fragment (k-step s, row half rh) of a 32-row group: row 16 rh + j, 16-byte piece 4 s + kg of the row,
    // stored at (4 s + kg) ^ w(row); w does not depend on the row half (w = row & 15, or (row >> 1) & 7)
    int bc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) bc[c] = (int)lds0 + j * RB + (((4 * c + kg) ^ (SWB == 4 ? j : ((j >> 1) & 7))) << 4);

    constexpr int WCAP = STG / NW;
    uint64_t* const wk = (uint64_t*)(smem + NBUF * TILE_BYTES) + wv * WCAP;
    uint32_t* const wq = (uint32_t*)((uint64_t*)(smem + NBUF * TILE_BYTES) + STG) + wv * WCAP;
    int wcnt = 0;
    auto to_memory = [&](uint64_t key, int q) __attribute__((always_inline)) {
        if (!a.mask || a.mask[(int64_t)(uint32_t)key - a.row_offset]) {
            const int pos = atomicAdd(&a.cnt[q * MF_CNT_STRIDE], 1);
            if (pos < HRS_CAND_CAP_BIG) a.cand[(size_t)q * HRS_CAND_CAP_BIG + pos] = key;
        }
    };
    auto flush_wave = [&]() __attribute__((always_inline)) {
        for (int e = lane; e < wcnt; e += 64) to_memory(wk[e], (int)wq[e]);
        wcnt = 0;
    };
    auto record = [&](bool hit, uint64_t key, int q) __attribute__((always_inline)) {
        const uint64_t bal = __builtin_amdgcn_ballot_w64(hit);
        if (bal == 0) return;
        const int np = __builtin_popcountll(bal);
        if (wcnt + np > WCAP) flush_wave();
        const int slot = wcnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (hit) {
            if (slot < WCAP) {
                wk[slot] = key;
                wq[slot] = (uint32_t)q;
            } else {
                to_memory(key, q);
            }
        }
        wcnt = __builtin_amdgcn_readfirstlane(min(wcnt + np, WCAP));
    };

    if (count > 0) {
        typedef const __attribute__((address_space(3))) char* lds_cptr;
        typedef const __attribute__((address_space(3))) i32x4* lds_frag;
        constexpr int PF = 4 < NF ? 4 : NF;
        typedef float f32xg __attribute__((ext_vector_type(4 * NSUB)));
        auto terms_request = [&](int64_t tile, f32xg& dst) __attribute__((always_inline)) {
            const char* p = uniform_ptr(a.group + tile * NSUB);
            if constexpr (NSUB == 2)
                asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(dst) : "s"(p));
            else
                asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(dst) : "s"(p));
        };
        issue(0, 0);
        load_queries();
#pragma unroll
        for (int g = 1; g < NBUF; ++g) issue(g, g);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
        __builtin_amdgcn_s_barrier();
        int buf = 0;
        if (q0 >= a.nq) {
            for (int64_t n = 0; n + 1 < count; ++n) {
                if (n & 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    issue(n + 3, (buf + 3) & 3);
                    issue(n + 4, buf);
                }
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
        } else {
            lds_cptr tb[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) tb[c] = (lds_cptr)(size_t)(uint32_t)bc[c];
            // fragment f = 2 s + rh of the 32-row group sb of the current tile
            auto frag_ptr = [&](int f, int sb) __attribute__((always_inline)) {
                const int s = f >> 1, rh = f & 1;
                return (lds_frag)(tb[s & (NB - 1)] + (sb * 32 + rh * 16) * RB + (s / NB) * FSTEP);
            };
            i32x4 pre[PF];
#pragma unroll
            for (int f = 0; f < PF; ++f) pre[f] = *frag_ptr(f, 0);
#ifdef HRS_SCAN_PROF  // the clock the chip holds in this loop: shader cycles over 100 MHz ticks
            const uint64_t ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
            for (int64_t n = 0; n < count; ++n) {
                const int64_t tile = cursor_take(cur_run);
                f32xg terms;
                terms_request(tile, terms);
                f32x4 scw[NSUB];
#pragma unroll
                for (int sb = 0; sb < NSUB; ++sb) {
                    i32x4 acc[2][QG];
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                        for (int g = 0; g < QG; ++g) acc[rh][g] = (i32x4){0, 0, 0, 0};
                    i32x4 av[NF];
#pragma unroll
                    for (int f = 0; f < PF; ++f) av[f] = pre[f];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if (f + PF < NF) av[f + PF] = *frag_ptr(f + PF, sb);
#pragma unroll
                        for (int g = 0; g < QG; ++g) acc[f & 1][g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[f], qf[g][f >> 1], acc[f & 1][g], 0, 0, 0);
                    }
#pragma unroll
                    for (int f = 0; f < NF - PF; ++f) {
                        __builtin_amdgcn_sched_group_barrier(0x008, QG, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, QG * PF, 0);
                    if (sb == 0) {
                        __builtin_amdgcn_sched_barrier(0);  // (the wait stays behind the block's last MFMA)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(terms)::"memory");
                        scw[0] = __builtin_shufflevector(terms, terms, 0, 1, 2, 3);
                        if constexpr (NSUB == 2) scw[1] = __builtin_shufflevector(terms, terms, 4, 5, 6, 7);
                    }
                    // lane (j, kg) holds rows 16 rh + 4 kg + r (r = 0..3) of query q0 + 16 g + j, for g = 0..3
                    auto epilogue = [&]() __attribute__((always_inline)) {
                        const float gs = scw[sb].w;  // s_g itself (.x is its reciprocal)
#ifdef HRS_SCAN_NOEPI  // timing only: a main pass that never looks at its accumulators (no results)
#pragma unroll
                        for (int g = 0; g < QG; ++g) asm volatile("" ::"v"(acc[0][g]), "v"(acc[1][g]));
                        return;
#endif
#pragma unroll
                        for (int g = 0; g < QG; ++g) {
                            int m = max(max(acc[0][g][0], acc[0][g][1]), acc[0][g][2]);
                            m = max(max(m, acc[0][g][3]), acc[1][g][0]);
                            m = max(max(m, acc[1][g][1]), acc[1][g][2]);
                            m = max(m, acc[1][g][3]);
                            // s_g (acc + xb) > tau  <=>  acc > tau / s_g - xb, xb = |qi| E_g + |eq| (C_g + E_g): see k_mfma_scan
                            float tf = fmaf(taum[g], scw[sb].x, -0.5f);
                            tf = fmaf(-kq[g].x, scw[sb].y, tf);
                            tf = fmaf(-kq[g].y, scw[sb].z, tf);
                            int ti;
                            asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ti) : "v"(tf));
                            if (__builtin_amdgcn_ballot_w64(m > ti) != 0) {  // wave-uniform: some lane holds such a row
                                const float xb = fmaf(kq[g].x, scw[sb].y, kq[g].y * scw[sb].z);
#pragma unroll
                                for (int rh = 0; rh < 2; ++rh) {
                                    const int m4 = max(max(acc[rh][g][0], acc[rh][g][1]), max(acc[rh][g][2], acc[rh][g][3]));
                                    if (__builtin_amdgcn_ballot_w64(m4 > ti) == 0) continue;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const bool hit = acc[rh][g][r] > ti;
                                        if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
                                        const uint32_t row = (uint32_t)tile * TR + sb * 32 + rh * 16 + 4 * kg + r;
                                        const float score = ((float)acc[rh][g][r] + xb) * gs;
                                        record(hit && row < (uint32_t)a.N, ((uint64_t)(~f32_orderable(score)) << 32) | (uint32_t)(row + (uint32_t)a.row_offset), q0 + g * 16 + j);
                                    }
                                }
                            }
                        }
                    };
                    // The two waves of a SIMD would otherwise leave every barrier in step: both in their MFMAs, then both
                    // in the vector work on their results, with the matrix pipe idle.  Waves 0..3 look at the last group's
                    // results in front of the barrier, waves 4..7 behind it, which sets the partners half a phase apart.
                    auto next_tile = [&]() __attribute__((always_inline)) {
                        if (n & 1) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            issue(n + 3, (buf + 3) & 3);
                            issue(n + 4, buf);
                        }
                        buf = buf + 1 == NBUF ? 0 : buf + 1;
#pragma unroll
                        for (int c = 0; c < NB; ++c) tb[c] = (lds_cptr)(size_t)(uint32_t)(bc[c] + buf * TILE_BYTES);
#pragma unroll
                        for (int f = 0; f < PF; ++f) pre[f] = *frag_ptr(f, 0);
                    };
                    const bool early = sb + 1 == NSUB && n + 1 < count && (n & 1) && wv < 4;
                    if (sb + 1 < NSUB) {
#pragma unroll
                        for (int f = 0; f < PF; ++f) pre[f] = *frag_ptr(f, sb + 1);
                    } else if (n + 1 < count && !early) {
                        next_tile();
                    }
                    epilogue();
                    if (early) next_tile();
                }
            }
#ifdef HRS_SCAN_PROF
            if (lane == 0 && wv == 0) {
                unsigned long long* st = (unsigned long long*)a.prof;
                atomicAdd(&st[4], (unsigned long long)(__builtin_amdgcn_s_memtime() - ck0));
                atomicAdd(&st[5], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - rt0));
            }
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flush_wave();
    } else {
        load_queries();  // (for the threshold that the first slice publishes)
    }
}

// ---- depth 1024 (bge-large class) ----------------------------------------------------------------
// 32 queries x 1024 depth are 256 registers per lane: more than the arch-VGPR half of the register
// file, so a workgroup is four waves (one per SIMD, 512 registers each) and most query fragments live
// in AGPRs, which only an asm MFMA will read as a source operand.  128 queries per workgroup; corpus
// tiles of 32 rows stream as two 512-deep stages (32 KB) through a 4-deep LDS ring.
__device__ __forceinline__ void mfma_qa(f32x16& acc, const bf16x8& rows, const bf16x8& q) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(rows), "a"(q));
}
__device__ __forceinline__ void mfma_qv(f32x16& acc, const bf16x8& rows, const bf16x8& q) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(rows), "v"(q));
}
__device__ __forceinline__ void mfma_result_fence(f32x16& acc) {  // 8-pass MFMA result -> VALU reader
    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc));
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_mfma_scan_wide(MfmaArgs a) {
    constexpr bool PRE = MODE != 0, MAXIMA = MODE == 1, LADDER = MODE == 2;
    constexpr int DP = 1024, KC = 512, SPT = 2, KSS = 32, ROWB = KC * 2, STAGE = 32 * ROWB, IPW = 8, NBUF = 4, NA = 56;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int qg = loc % a.nqg;                       // a.nqg: query groups of 128 here (1, 2, 4 or 8)
    const int nslices = HRS_NUM_CU / a.nqg;
    const int slice = xcd * (32 / a.nqg) + loc / a.nqg;
    const int q0 = qg * 128 + wv * 32;

    // this wave's queries: requested right behind the first corpus stage (independent requests; a short pass
    // shows a serial start)
    bf16x8 qf[SPT * KSS];
    float tau = 0.f, linv = 0.f;
    float mx = -__builtin_inff();
    auto load_queries = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < SPT * KSS; ++s) qf[s] = *(const bf16x8*)(a.qb + qfrag_offset(q0 + i, s, h, SPT * KSS));
        tau = MAXIMA ? 0.f : (LADDER ? a.tau[q0 + i] : raised_tau(a, q0 + i));
        if (MODE == 0 && slice == 0 && h == 0) a.tau_out[q0 + i] = tau;  // one writer per query
        linv = LADDER ? a.lad_inv[q0 + i] : 0.f;
#pragma unroll
        for (int s = 0; s < SPT * KSS; ++s) {
            if (s < NA)
                asm volatile("" : "+a"(qf[s]));
            else
                asm volatile("" : "+v"(qf[s]));
        }
        asm volatile("" : "+v"(tau), "+v"(linv));
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store above too: the loop's waits count requests
    };

    const int64_t t_begin = a.tiles * slice / nslices;
    const int64_t t_end = a.tiles * (slice + 1) / nslices;
    const int64_t count = PRE ? (t_end - t_begin) / a.ss : (t_end - t_begin);
    const int64_t full_tiles = a.N / 32;
    auto tile_at = [&](int64_t n) -> int64_t {
        if (n >= count) n = count - 1;
        if (!PRE) return t_begin + n;
        int64_t t = t_begin + n * a.ss + (int64_t)(hash32((uint32_t)(slice * 0x9E3779B1u + n)) % (uint32_t)a.ss);
        if (t >= full_tiles) t = full_tiles - 1;
        return t < 0 ? 0 : t;
    };
    const int64_t stages = count * SPT;
    if (stages > 0) {
        int srcoff[IPW];
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii) {
            const int g = (wv * IPW + ii) * 64 + lane;
            const int r = g >> 6, p = g & 63;
            srcoff[ii] = r * (DP * 2) + ((p ^ (r & 15)) << 4);
        }
        const uint32_t lds0 = lds_addr_of(smem);
        auto issue = [&](int64_t g) {
            if (g >= stages) g = stages - 1;
            const int kc = (int)(g % SPT);
            const char* base = uniform_ptr((const char*)a.shadow + (tile_at(g / SPT) * 32 * DP + kc * KC) * 2);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(g % NBUF) * STAGE + wv * (IPW * 1024));
#pragma unroll
            for (int ii = 0; ii < IPW; ++ii) glds16_asm(base, srcoff[ii], dst + ii * 1024);
        };
        typedef const __attribute__((address_space(3))) char* lds_cptr;
        typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
        const int rowoff = (int)lds0 + i * ROWB;
        const int vx = h ^ (i & 15);
        // Large batches are bound by the matrix cores and every barrier is a bubble in all four pipes: the waves
        // then meet once per tile (two stages), where both stages of the tile have landed and the two buffers of
        // the previous tile are requested anew.  Small batches are bound by the bytes in flight and keep the
        // per-stage barrier with three stages outstanding.
        const bool pairs = a.nq >= 512;
        issue(0);  // first stage, queries, rest of the ring: requests come back in order
        load_queries();
        issue(1);
        if (!pairs) issue(2);
        f32x16 acc;
        for (int64_t g = 0; g < stages; ++g) {
            const int kc = (int)(g % SPT);
            if (!pairs) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
                __builtin_amdgcn_s_barrier();
                issue(g + 3);
            } else if (kc == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue(g + 2);
                issue(g + 3);
            }
            lds_cptr tb[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) tb[c] = (lds_cptr)(size_t)(uint32_t)(rowoff + (int)(g % NBUF) * STAGE + (((2 * c) ^ vx) << 4));
            if (kc == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            constexpr int PF = 6;
            auto run = [&](auto kc_tag) __attribute__((always_inline)) {
                constexpr int KCI = decltype(kc_tag)::value;
                bf16x8 av[KSS];
#pragma unroll
                for (int s = 0; s < PF; ++s) av[s] = *(lds_frag)(tb[s & 7] + (s >> 3) * 256);
#pragma unroll
                for (int s = 0; s < KSS; ++s) {
                    if (s + PF < KSS) av[s + PF] = *(lds_frag)(tb[(s + PF) & 7] + ((s + PF) >> 3) * 256);
                    if (KCI * KSS + s < NA)
                        mfma_qa(acc, av[s], qf[KCI * KSS + s]);  // D[corpus row][query]
                    else
                        mfma_qv(acc, av[s], qf[KCI * KSS + s]);
                }
            };
            if (kc == 0) {
                run(std::integral_constant<int, 0>{});
            } else {
                run(std::integral_constant<int, 1>{});
                mfma_result_fence(acc);
                float m = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
                if (MAXIMA) {
                    mx = fmaxf(mx, m);
                } else if (LADDER) {
                    if (m > tau) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (acc[r] > tau) atomicAdd(&a.hist[(q0 + i) * 16 + max(0, min(15, (int)((acc[r] - tau) * linv)))], 1);
                    }
                } else if (m > tau) {
                    const int64_t tile = tile_at(g / SPT);
                    const int q = q0 + i;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (acc[r] > tau) {
                            const int64_t row = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            if (row < a.N && (!a.mask || a.mask[row])) {
                                const int pos = atomicAdd(&a.cnt[q * MF_CNT_STRIDE], 1);
                                if (pos < HRS_CAND_CAP_BIG)
                                    a.cand[(size_t)q * HRS_CAND_CAP_BIG + pos] =
                                        ((uint64_t)(~f32_orderable(acc[r])) << 32) | (uint32_t)(row + a.row_offset);
                            }
                        }
                    }
                }
            }
        }
    } else {
        load_queries();  // (for the threshold that the first slice publishes)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MAXIMA) a.maxima[(size_t)(q0 + i) * HRS_MAX_STREAMS + slice * 2 + h] = mx;
}

// fp32 [N, D] -> bf16 [Npad, Dpad] (zero padded) and min/max squared row norm. One wave per row.
__global__ __launch_bounds__(256) void k_build_shadow(const float* __restrict__ C, int64_t N, int D, int64_t Npad,
                                                      int Dpad, __bf16* __restrict__ out,
                                                      unsigned long long* __restrict__ stats, int64_t r0) {
    // 16 rows per workgroup (4 per wave): one pair of global atomics per workgroup
    __shared__ unsigned long long nmin, nmax;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        nmin = ~0ull;
        nmax = 0ull;
    }
    __syncthreads();
    bool any = false;
    for (int j = 0; j < 4; ++j) {
        const int64_t r = r0 + (int64_t)blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + j;  // r0: first row (re)built
        if (r >= Npad) break;
        double ss = 0.0;
        for (int c = lane; c < Dpad; c += 64) {
            float v = (r < N && c < D) ? C[r * D + c] : 0.f;
            out[r * Dpad + c] = (__bf16)v;
            ss += (double)v * (double)v;
        }
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0 && r < N) {
            // non-negative doubles order like their bit patterns
            atomicMin(&nmin, (unsigned long long)__double_as_longlong(ss));
            atomicMax(&nmax, (unsigned long long)__double_as_longlong(ss));
            any = true;
        }
    }
    (void)any;
    __syncthreads();
    if (threadIdx.x == 0 && r0 + (int64_t)blockIdx.x * 16 < N) {
        atomicMin(&stats[0], nmin);
        atomicMax(&stats[1], nmax);
    }
}


// fp32 [N, D] -> int8 [Npad, Dpad] with one scale per 32-row group (s_g = max |c| of the group / 127,
// value = rint(c / s_g)) and the group's pairs (s_g, M_g), (1 / s_g, M_g).  One workgroup per group.
// Also min/max squared row norm (stats[0..1], doubles).
__global__ __launch_bounds__(256) void k_build_shadow_i8(const float* __restrict__ C, int64_t N, int D, int Dpad,
                                                         int8_t* __restrict__ out, float4* __restrict__ pre_terms,
                                                         float4* __restrict__ main_terms,
                                                         unsigned long long* __restrict__ stats, int64_t g0) {
    __shared__ float red[4];
    __shared__ unsigned long long e2max, c2max;  // largest squared norm of a row's residuals / of its int8 values
    __shared__ unsigned long long nmin, nmax;  // squared norms as bit patterns (non-negative doubles order like them)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t g = g0 + blockIdx.x, r0 = g * 32;  // g0: first group (re)built
    if (threadIdx.x == 0) {
        e2max = c2max = 0ull;
        nmin = ~0ull;
        nmax = 0ull;
    }
    float mxv = 0.f;
    for (int rr = wv; rr < 32; rr += 4) {
        const int64_t r = r0 + rr;
        if (r >= N) break;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = *(const float4*)(C + r * D + c);
            mxv = fmaxf(mxv, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    }
    for (int o = 32; o > 0; o >>= 1) mxv = fmaxf(mxv, __shfl_xor(mxv, o));
    if (lane == 0) red[wv] = mxv;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float gs = gmax / 127.0f;
    for (int rr = wv; rr < 32; rr += 4) {
        const int64_t r = r0 + rr;
        double e2 = 0.0, c2 = 0.0;  // of this row: squared residuals (value / s_g - int8, in real arithmetic), squared int8
        double ss = 0.0;
        for (int c = lane * 4; c < Dpad; c += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < N && c < D) v = *(const float4*)(C + r * D + c);
            ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
            if (gs > 0.f) {
                q0 = (int)rintf(v.x / gs);
                q1 = (int)rintf(v.y / gs);
                q2 = (int)rintf(v.z / gs);
                q3 = (int)rintf(v.w / gs);
                const double r0d = (double)v.x / (double)gs - q0, r1d = (double)v.y / (double)gs - q1;
                const double r2d = (double)v.z / (double)gs - q2, r3d = (double)v.w / (double)gs - q3;
                e2 += r0d * r0d + r1d * r1d + r2d * r2d + r3d * r3d;
            }
            c2 += (double)(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            const uint32_t packed = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) |
                                    ((uint32_t)(q3 & 0xff) << 24);
            *(uint32_t*)(out + r * Dpad + c) = packed;
        }
        for (int o = 32; o > 0; o >>= 1) {
            e2 += __shfl_xor(e2, o);
            c2 += __shfl_xor(c2, o);
            ss += __shfl_xor(ss, o);
        }
        if (lane == 0 && r < N) {
            atomicMin(&nmin, (unsigned long long)__double_as_longlong(ss));
            atomicMax(&nmax, (unsigned long long)__double_as_longlong(ss));
            atomicMax(&e2max, (unsigned long long)__double_as_longlong(e2));
            atomicMax(&c2max, (unsigned long long)__double_as_longlong(c2));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (r0 < N) {  // one pair of global atomics per group, not per row
            atomicMin(&stats[0], nmin);
            atomicMax(&stats[1], nmax);
        }
        // the two norms, rounded up (fp64 square root, then a relative and an absolute margin far above the fp32 rounding)
        const float eg = (float)(sqrt(__longlong_as_double((long long)e2max)) * 1.000002 + 1e-4);
        const float cg = (float)(sqrt(__longlong_as_double((long long)c2max)) * 1.000002 + 1e-4);
        // an all-zero group scores exactly 0: a huge reciprocal keeps it out for any positive threshold
        pre_terms[g] = make_float4(gs, eg, cg + eg, 0.f);
        main_terms[g] = make_float4(gs > 0.f ? 1.0f / gs : 1e30f, eg, cg + eg, gs);

    }
}

// min/max squared row norm of a corpus that has no shadow (the shadow builders compute them on the way).
// One wave per row, 16 rows per workgroup, one pair of global atomics per workgroup.
__global__ __launch_bounds__(256) void k_row_norm_stats(const float* __restrict__ C, int64_t N, int D,
                                                        unsigned long long* __restrict__ stats, int64_t r0) {
    __shared__ unsigned long long nmin, nmax;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        nmin = ~0ull;
        nmax = 0ull;
    }
    __syncthreads();
    for (int j = 0; j < 4; ++j) {
        const int64_t r = r0 + (int64_t)blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + j;
        if (r >= N) break;
        double ss = 0.0;
        for (int c = lane; c < D; c += 64) {
            const float v = C[r * D + c];
            ss += (double)v * (double)v;
        }
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) {
            atomicMin(&nmin, (unsigned long long)__double_as_longlong(ss));
            atomicMax(&nmax, (unsigned long long)__double_as_longlong(ss));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && r0 + (int64_t)blockIdx.x * 16 < N) {
        atomicMin(&stats[0], nmin);
        atomicMax(&stats[1], nmax);
    }
}

__global__ void k_init_stats(unsigned long long* stats) {
    stats[0] = (unsigned long long)__double_as_longlong(1e300);
    stats[1] = 0ull;
}

void hrs_free_shadow(hrs_corpus* c) {
    if (c->shadow) (void)hipFree(c->shadow);
    if (c->group_pre) (void)hipFree(c->group_pre);
    if (c->norm_stats) (void)hipFree(c->norm_stats);
    c->shadow = nullptr;
    c->group_pre = c->group_main = nullptr;
    c->norm_stats = nullptr;
    c->kind = HRS_SHADOW_NONE;
    c->cap = 0;
}

static int64_t shadow_npad(int64_t N) { return (N + 63) / 64 * 64 + 64; }  // one spare tile: clamped prefetches stay in bounds

// Allocates shadow storage for `cap` rows (a multiple of 64) of the corpus' kind / Dpad.
static int shadow_alloc(const hrs_corpus* c, int64_t cap, void** shadow, float4** group_pre) {
    const size_t esz = c->kind == HRS_SHADOW_I8 ? 1 : sizeof(__bf16);
    *shadow = nullptr;
    *group_pre = nullptr;
    hipError_t e = hipMalloc(shadow, (size_t)cap * c->Dpad * esz);
    if (e == hipSuccess && c->kind == HRS_SHADOW_I8) e = hipMalloc((void**)group_pre, (size_t)(2 * (cap / 32) + 64) * sizeof(float4));
    if (e != hipSuccess) {
        if (*shadow) (void)hipFree(*shadow);
        *shadow = nullptr;
        hrs_set_error("corpus shadow allocation of %lld rows failed: %s", (long long)cap, hipGetErrorString(e));
        return HRS_ERR_NOMEM;
    }
    return HRS_OK;
}

// Quantises rows [r_from, Npad) of the corpus into the shadow (r_from a multiple of 32) and folds their norms
// into the statistics.
static int shadow_fill(hrs_corpus* c, int64_t r_from, hipStream_t st) {
    if (c->kind == HRS_SHADOW_I8) {
        const int64_t g0 = r_from / 32, groups = c->Npad / 32;
        hipLaunchKernelGGL(k_build_shadow_i8, dim3((unsigned)(groups - g0)), dim3(256), 0, st, c->C, c->N, c->D, c->Dpad,
                           (int8_t*)c->shadow, c->group_pre, c->group_main, (unsigned long long*)c->norm_stats, g0);
    } else {
        hipLaunchKernelGGL(k_build_shadow, dim3((unsigned)((c->Npad - r_from + 15) / 16)), dim3(256), 0, st, c->C, c->N, c->D,
                           c->Npad, c->Dpad, (__bf16*)c->shadow, (unsigned long long*)c->norm_stats, r_from);
    }
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// (Re)build the scan shadow of a corpus: int8 with group scales (default) or bf16.
int hrs_build_shadow(hrs_ctx* ctx, hrs_corpus* c, int kind, hipStream_t st) {
    (void)ctx;
    hrs_free_shadow(c);
    c->want_kind = kind;
    if (kind == HRS_SHADOW_AUTO) kind = c->D <= 1024 ? HRS_SHADOW_I8 : HRS_SHADOW_BF16;
    c->Npad = shadow_npad(c->N);
    if (c->N == 0) return HRS_OK;
    hipError_t e = hipMalloc((void**)&c->norm_stats, 2 * sizeof(double));
    if (e != hipSuccess) {
        hrs_set_error("corpus statistics allocation failed: %s", hipGetErrorString(e));
        return HRS_ERR_NOMEM;
    }
    hipLaunchKernelGGL(k_init_stats, dim3(1), dim3(1), 0, st, (unsigned long long*)c->norm_stats);
    if (kind == HRS_SHADOW_NONE || c->D > 1024) {  // no shadow (deeper rows: exact scan only); the row norms all the same
        hipLaunchKernelGGL(k_row_norm_stats, dim3((unsigned)((c->N + 15) / 16)), dim3(256), 0, st, c->C, c->N, c->D,
                           (unsigned long long*)c->norm_stats, (int64_t)0);
        HRS_CHECK_HIP(hipGetLastError());
        return HRS_OK;
    }
    c->Dpad = (c->D + 127) / 128 * 128;
    if (kind == HRS_SHADOW_I8) {
        if (c->Dpad > 512) c->Dpad = c->Dpad <= 768 ? 768 : 1024;
    } else {
        if (c->Dpad > 512 && c->Dpad < 1024) c->Dpad = 1024;  // one wide kernel covers (512, 1024]
    }
    c->kind = kind;
    int rc = shadow_alloc(c, c->Npad, &c->shadow, &c->group_pre);
    if (rc) {
        hrs_free_shadow(c);
        return rc;
    }
    c->cap = c->Npad;
    if (kind == HRS_SHADOW_I8) c->group_main = c->group_pre + c->cap / 32 + 32;
    return shadow_fill(c, 0, st);
}

// The corpus grew to N_new rows at C (the first c->N rows unchanged, possibly at a new address): only the new
// rows -- and the last, partly filled 32-row group, whose int8 scale may change -- are quantised.  The shadow's
// allocation doubles when it is full, so a long run of small appends costs O(rows appended) in total.
int hrs_append_shadow(hrs_ctx* ctx, hrs_corpus* c, const float* C, int64_t N_new, hipStream_t st) {
    const int64_t N_old = c->N;
    if (N_new == N_old) {
        c->C = C;
        return HRS_OK;
    }
    if (N_old == 0 || (!c->shadow && !c->norm_stats)) {
        c->C = C;
        c->N = N_new;
        return hrs_build_shadow(ctx, c, c->want_kind, st);
    }
    const int64_t Npad_new = shadow_npad(N_new);
    if (!c->shadow) {  // no shadow by choice or depth: the statistics alone
        c->C = C;
        c->N = N_new;
        c->Npad = Npad_new;
        hipLaunchKernelGGL(k_row_norm_stats, dim3((unsigned)((N_new - N_old + 15) / 16)), dim3(256), 0, st, c->C, c->N, c->D,
                           (unsigned long long*)c->norm_stats, N_old);
        HRS_CHECK_HIP(hipGetLastError());
        return HRS_OK;
    }
    const int64_t r_from = N_old / 32 * 32;
    if (Npad_new > c->cap) {
        // The corpus keeps describing its old rows until the larger shadow exists and holds them: a failure below leaves
        // the handle as it was (N_old rows over the old matrix), never one that claims rows its shadow does not cover.
        int64_t cap = c->cap * 2;
        if (cap < Npad_new) cap = Npad_new;
        void* ns = nullptr;
        float4* ng = nullptr;
        int rc = shadow_alloc(c, cap, &ns, &ng);
        if (rc) return rc;
        const size_t esz = c->kind == HRS_SHADOW_I8 ? 1 : sizeof(__bf16);
        hipError_t e = hipMemcpyAsync(ns, c->shadow, (size_t)r_from * c->Dpad * esz, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && ng) e = hipMemcpyAsync(ng, c->group_pre, (size_t)(r_from / 32) * sizeof(float4), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && ng)
            e = hipMemcpyAsync(ng + cap / 32 + 32, c->group_main, (size_t)(r_from / 32) * sizeof(float4), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // the copies are done (and so is whatever read the old shadow on st)
        if (e != hipSuccess) {
            (void)hipFree(ns);
            if (ng) (void)hipFree(ng);
            hrs_set_error("hrs_corpus_append: copying the shadow into its larger allocation failed: %s", hipGetErrorString(e));
            return HRS_ERR_HIP;
        }
        (void)hipFree(c->shadow);
        if (c->group_pre) (void)hipFree(c->group_pre);
        c->shadow = ns;
        c->group_pre = ng;
        c->group_main = ng ? ng + cap / 32 + 32 : nullptr;
        c->cap = cap;
    }
    c->C = C;
    c->N = N_new;
    c->Npad = Npad_new;
    return shadow_fill(c, r_from, st);
}

// The few control words a call starts from zero (queue lengths, statistics) are cleared by the first
// kernel of the call instead of by memset nodes of their own: three launches less per call.
struct ZeroList {
    int32_t* p[3];
    int n[3];  // 32-bit words each (<= 64)
};
__device__ __forceinline__ void zero_words(const ZeroList& z) {
    if (blockIdx.x == 0 && threadIdx.x < 192) {
        const int w = threadIdx.x >> 6, t = threadIdx.x & 63;
        if (z.p[w] && t < z.n[w]) z.p[w][t] = 0;
    }
}

// Queries fp32 [B, D] -> int8 [Bpad, Dpad] with one scale per query (max |q| / 127); squared norms in
// fp64, the scale, and the two norms of the error bound: of the int8 values and of the residuals
// (q / scale - int8, in real arithmetic), rounded up.  One wave per query.
__global__ __launch_bounds__(256) void k_convert_q_i8(const float* __restrict__ Q, int B, int D, int Bpad, int Dpad,
                                                      int8_t* __restrict__ qb, int8_t* __restrict__ qb16, double* __restrict__ qn2,
                                                      float* __restrict__ qscale, float2* __restrict__ kq, ZeroList z) {
    zero_words(z);
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Bpad) return;
    float mxv = 0.f;
    if (q < B)
        for (int c = lane; c < D; c += 64) mxv = fmaxf(mxv, fabsf(Q[(size_t)q * D + c]));
    for (int o = 32; o > 0; o >>= 1) mxv = fmaxf(mxv, __shfl_xor(mxv, o));
    const float sq = mxv / 127.0f;
    double ss = 0.0, i2 = 0.0, e2 = 0.0;
    for (int c = lane; c < Dpad; c += 64) {
        const float v = (q < B && c < D) ? Q[(size_t)q * D + c] : 0.f;
        const int qi = sq > 0.f ? (int)rintf(v / sq) : 0;
        qb[qfrag_offset(q, c >> 5, (c >> 4) & 1, Dpad >> 5) + (c & 15)] = (int8_t)qi;
        if (qb16) qb16[qfrag16_offset(q, c, Dpad >> 6)] = (int8_t)qi;  // the main pass on the 16 x 16 x 64 shape reads this form
        const double res = sq > 0.f ? (double)v / (double)sq - qi : 0.0;
        i2 += (double)(qi * qi);
        e2 += res * res;
        ss += (double)v * (double)v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o);
        i2 += __shfl_xor(i2, o);
        e2 += __shfl_xor(e2, o);
    }
    if (lane == 0) {
        qn2[q] = ss;
        qscale[q] = sq;
        kq[q] = make_float2((float)(sqrt(i2) * 1.000002 + 1e-4), (float)(sqrt(e2) * 1.000002 + 1e-4));
    }
}

// Queries fp32 [B, D] -> bf16 [Bpad, Dpad] (zero padded), squared norms in fp64. One wave per query.
__global__ __launch_bounds__(256) void k_convert_q(const float* __restrict__ Q, int B, int D, int Bpad, int Dpad,
                                                   __bf16* __restrict__ qb, double* __restrict__ qn2, ZeroList z) {
    zero_words(z);
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Bpad) return;
    double ss = 0.0;
    for (int c = lane; c < Dpad; c += 64) {
        float v = (q < B && c < D) ? Q[(size_t)q * D + c] : 0.f;
        *(__bf16*)((char*)qb + qfrag_offset(q, c >> 4, (c >> 3) & 1, Dpad >> 4) + (c & 7) * 2) = (__bf16)v;
        ss += (double)v * (double)v;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) qn2[q] = ss;
}

// tau[q] = j-th largest of the query's M stream maxima -- for rows that score independently about
// `target` rows exceed it.  Also: the ladder step 16 / (largest maximum - tau) and cleared counters.
// The j-th largest comes from a radix select over the 32-bit order-preserving keys (four passes of a
// 256-bin histogram), not from a sort.  Streams without a sample hold -inf and rank last.
// Queries that must record nothing get a threshold no score reaches: the padding of a partial batch, and on the
// int8 shadow a query of zeros (scale 0) -- all their integer scores are 0, closer to any threshold than the integer
// compare of the main pass resolves, so every row of the corpus would be recorded.  (A query of zeros then has no
// candidates and is answered by the exact scan.)
__global__ __launch_bounds__(256) void k_pick_tau(const float* __restrict__ maxima, int M, int j, float* __restrict__ tau,
                                                  int* __restrict__ cnt, float* __restrict__ lad_inv, int* __restrict__ hist,
                                                  int nq, const float* __restrict__ qscale) {
    __shared__ uint32_t bins[256];
    __shared__ uint32_t sel[2];  // chosen bin, rank within it
    __shared__ uint32_t smin[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    constexpr int PER = HRS_MAX_STREAMS / 256;
    uint32_t key[PER];  // ascending key = descending value
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int t = tid + e * 256;
        key[e] = t < M ? ~f32_orderable(maxima[(size_t)q * HRS_MAX_STREAMS + t]) : 0xFFFFFFFFu;
        best = min(best, key[e]);
    }
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
    if ((tid & 63) == 0) smin[tid >> 6] = best;
    uint32_t prefix = 0, rank = (uint32_t)(j < M ? j : M);  // 1-based rank among the M keys
    for (int shift = 24; shift >= 0; shift -= 8) {
        bins[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e)
            if (tid + e * 256 < M && (shift == 24 || (key[e] >> (shift + 8)) == prefix)) atomicAdd(&bins[(key[e] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < 64) {  // wave 0: inclusive scan of the 256 bins, four per lane
            uint32_t c[4], run = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                run += bins[tid * 4 + e];
                c[e] = run;
            }
            uint32_t incl = run;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
                if (tid >= o) incl += v;
            }
            const uint32_t before = incl - run;  // keys in the bins of lower lanes
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t lo = before + (e ? c[e - 1] : 0u), hi = before + c[e];
                if (rank > lo && rank <= hi) {
                    sel[0] = (uint32_t)(tid * 4 + e);
                    sel[1] = rank - lo;
                }
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | sel[0];
        rank = sel[1];
    }
    if (tid < 16) hist[q * 16 + tid] = 0;
    if (tid == 0) {
        const float v = orderable_f32(~prefix);
        const float top = orderable_f32(~min(min(smin[0], smin[1]), min(smin[2], smin[3])));
        const bool off = q >= nq || (qscale && qscale[q] == 0.f);
        tau[q] = off ? 1e30f : v;
        cnt[q * MF_CNT_STRIDE] = 0;
        lad_inv[q] = (!off && top > v && top - v < 1e30f) ? 16.0f / (top - v) : 0.f;
    }
}

struct SelectArgs {
    const float* C;
    const float* Q;
    int D;
    int64_t N;
    int64_t row_offset;
    int B;
    int k;
    const int* cnt;
    const uint64_t* cand;
    const float* tau;
    const double* qn2;
    const double* norm_stats;  // min/max |c|^2
    double eps_rel;
    const float* qscale;       // int8 shadow only (else null): per-query scale; candidate scores are then
                               // upper bounds of score / scale
    uint64_t* keys_out;  // [B][k]
    float* dist_out;     // nullable [B][k]: the keys' distances ...
    int64_t* rows_out;   // nullable [B][k]: ... and rows (-1 and +inf where fewer than k rows exist)
    int32_t* qlist;      // queue of queries for the exact fallback
    int32_t* nflag;
    int large_first;     // every query goes to the large variant at once (deep rows and a large k: see the launcher)
    int32_t* qlist2;     // queue (chunk-local indices) of queries deferred to the large variant; [0] of nflag2 = count
    int32_t* nflag2;
    long long* stats;    // [0] fallbacks, [1] max candidates, [2] overflows
    int qbase;           // first query of this chunk
    int first_rows;      // rows re-scored in the first round
};

// Two sizes of the same kernel.  The small one (lists up to HRS_CAND_CAP, re-scored prefix up to HRS_RESCORE_MAX
// rows, 80 KB of LDS: two workgroups per CU) serves every query first; a query whose list is longer or whose
// prefix would have to grow further -- runs of near-duplicate rows around the top -- is deferred to the
// large one (lists up to HRS_CAND_CAP_BIG, prefix up to 4096, one workgroup per CU), and only what that
// cannot prove goes to the exact scan.

// Per query: sort the candidates by their recorded score (an upper bound of the true score up to the
// shadow's error term), re-score the best HRS_RESCORE exactly, and extend the re-scored prefix while any
// candidate outside it could still beat the current k-th exact distance.  Accept when nothing outside
// the prefix -- recorded or never recorded (score <= tau) -- can; otherwise queue the query for the
// exact scan.
template <int SPACE, bool BIG>
__device__ __forceinline__ void select_rescore_one(const SelectArgs& a, int ql) {  // ql: query within the chunk
    constexpr int SCAP = BIG ? HRS_CAND_CAP_BIG : HRS_CAND_CAP;  // candidates sorted in LDS
    constexpr int RMAX = BIG ? 4096 : HRS_RESCORE_MAX;           // largest re-scored prefix
    extern __shared__ __attribute__((aligned(16))) char sel_smem[];
    uint64_t* s = (uint64_t*)sel_smem;                       // [SCAP]
    uint64_t* ek = s + SCAP;                                 // [RMAX]
    float* qs = (float*)(ek + RMAX);                         // [1024] the query (D <= 1024 on this path)
    char* stage = (char*)(qs + 1024);                        // per wave: 64 rows x one 128-byte line (+16 pad)
    // two words that all threads share live in the padding of the first staged row (bytes 128..143 of a 144-byte row
    // are never written or read by the row fetches): the small variant's LDS is then exactly half a CU's
    int& need_s = *(int*)(stage + 128);
    int& ok = *(int*)(stage + 132);
    const int q = a.qbase + ql;      // query in the caller's batch
    const int tid = threadIdx.x;
    const int craw = a.cnt[ql * MF_CNT_STRIDE];
    const int c = craw < SCAP ? craw : SCAP;
    const bool overflow = craw > HRS_CAND_CAP_BIG;  // rows above tau were dropped: only the exact scan can answer
    // longer than this variant sorts, or a call whose proofs are expected to need a longer re-scored prefix than this
    // variant holds (it would re-score 512 rows and then hand the query over all the same): the large one takes it
    if (!BIG && !overflow && (craw > SCAP || a.large_first)) {
        if (tid == 0) a.qlist2[atomicAdd(a.nflag2, 1)] = ql;
        return;
    }
#ifdef HRS_SEL_PROF
    long long pt0 = wall_clock64(), pt_resc = 0, pt_sort = 0, pt_rounds = 0;
#endif
    int n2 = 256;
    while (n2 < c) n2 <<= 1;
    auto sort_keys = [&](uint64_t* p, int n) __attribute__((always_inline)) {
        if (n <= 2048) {
            bitonic_sort_u64_256t(p, n, tid);
        } else if (n == 4096) {  // 16 keys per thread in registers
            __syncthreads();
            bitonic_sort_u64_regs<16>(p, tid);
        } else if constexpr (BIG) {  // 32 (this variant runs one workgroup per CU)
            __syncthreads();
            bitonic_sort_u64_regs<32>(p, tid);
        }
    };
    // first round: the rows that must be re-scored for a proof are those whose recorded bound reaches the k-th best
    // exact score -- for the int8 shadow several times k, rarely more than 8 k + 32 (the launcher's first_rows)
    int first = a.first_rows;
    // The first round needs the `first` best candidates by recorded score as a SET, in no order, and nearly every
    // query is settled by it: instead of sorting the list, the small variant bins the recorded scores (1024 bins over
    // their range, in LDS), takes the bins that together hold at most `first` candidates, and moves those to the
    // front of the list.  A candidate's bin rises with its key, so the front is a prefix of the sorted order; the list
    // is sorted only if a second round turns out to be needed (sorted == false until then).
    bool sorted = true;
    int want = c < first ? c : first;
    const float* qr = a.Q + (size_t)q * a.D;
    for (int t = tid; t < a.D; t += 256) qs[t] = qr[t];
    if (!BIG && c > 0) {
        constexpr int KPT = SCAP / 256;
        uint64_t kk[KPT];
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            kk[u] = HRS_PAD_KEY;
            if (u * 256 < c) {
                const int t = tid + u * 256;
                if (t < c) {
                    kk[u] = a.cand[(size_t)ql * HRS_CAND_CAP_BIG + t];
                    lo = min(lo, (uint32_t)(kk[u] >> 32));
                    hi = max(hi, (uint32_t)(kk[u] >> 32));
                }
            }
        }
        int* hist = (int*)ek;  // [1024] bins, then [1024] their running sums; ek itself is filled further down
        uint32_t* ctl = (uint32_t*)(hist + 1024);  // [0] lowest, [1] highest recorded score, [2] last bin taken + 1, [3] front, [4] back
        for (int t = tid; t < 1024; t += 256) hist[t] = 0;
        if (tid == 0) {
            ctl[0] = 0xFFFFFFFFu;
            ctl[1] = 0u;
            ctl[2] = 0u;
            ctl[3] = 0u;
            ctl[4] = 0u;
        }
        __syncthreads();
        if (c > first) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {  // one atomic per wave, not per thread
                lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
                hi = max(hi, (uint32_t)__shfl_xor((int)hi, d));
            }
            if ((tid & 63) == 0) {
                atomicMin(&ctl[0], lo);
                atomicMax(&ctl[1], hi);
            }
            __syncthreads();
            const uint32_t base = ctl[0];
            const float scale = 1023.0f / (float)max(ctl[1] - base, 1u);
            int bin[KPT];
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                bin[u] = min(1023, (int)((float)((uint32_t)(kk[u] >> 32) - base) * scale));
                if (u * 256 < c && tid + u * 256 < c) atomicAdd(&hist[bin[u]], 1);
            }
            __syncthreads();
            // running sums of the bins (thread t: bins 4 t .. 4 t + 3); the last bin whose running sum is at most `first`
            int h4[4], run = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h4[i] = hist[4 * tid + i];
                run += h4[i];
            }
            int incl = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if ((tid & 63) >= d) incl += o;
            }
            int* wsum = hist + 1024;  // (behind ctl: [8..12))
            if ((tid & 63) == 63) wsum[8 + (tid >> 6)] = incl;
            __syncthreads();
            int before = incl - run;
            for (int w = 0; w < (tid >> 6); ++w) before += wsum[8 + w];
            int cum = before;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int prev = cum;
                cum += h4[i];
                if (prev <= first && cum > first) {  // exactly one (bin, thread): c > first
                    ctl[2] = (uint32_t)(4 * tid + i);  // bins below this one are taken
                    ctl[4] = (uint32_t)prev;           // ... they hold this many candidates
                }
            }
            __syncthreads();
            const int take = (int)ctl[2], front_n = (int)ctl[4];
            if (front_n >= first / 2) {
                // partition in place: every key is in a register by now
                __syncthreads();
                if (tid == 0) ctl[4] = 0u;
                __syncthreads();
#pragma unroll
                for (int u = 0; u < KPT; ++u) {
                    if (u * 256 < c && tid + u * 256 < c) {
                        const bool fr = bin[u] < take;
                        const int pos = fr ? (int)atomicAdd(&ctl[3], 1u) : front_n + (int)atomicAdd(&ctl[4], 1u);
                        s[pos] = kk[u];
                    }
                }
                want = front_n;
                sorted = false;
            } else {
                // (scores bunched into one bin -- runs of equal rows: the sorted list serves)
#pragma unroll
                for (int u = 0; u < KPT; ++u) s[tid + u * 256] = kk[u];
            }
        } else {
            // every candidate is re-scored in the first round: no order is needed at all
#pragma unroll
            for (int u = 0; u < KPT; ++u)
                if (u * 256 < c && tid + u * 256 < c) s[tid + u * 256] = kk[u];
            sorted = false;
        }
        __syncthreads();
    } else {
        for (int t = tid; t < n2; t += 256) s[t] = t < c ? a.cand[(size_t)ql * HRS_CAND_CAP_BIG + t] : HRS_PAD_KEY;
    }
    for (int t = tid; t < RMAX; t += 256) ek[t] = HRS_PAD_KEY;
#ifdef HRS_SEL_PROF
    __syncthreads();
    const long long pt_load = wall_clock64() - pt0;
#endif
    if (sorted) sort_keys(s, n2);
#ifdef HRS_SEL_PROF
    pt_sort = wall_clock64() - pt0 - pt_load;
#endif

    const double qn = a.qn2[ql], ncmin = a.norm_stats[0], ncmax = a.norm_stats[1];
    const double sq = a.qscale ? (double)a.qscale[ql] : 1.0;
    // smallest distance a row can have whose recorded score is at most A
    auto dist_floor = [&](float A) -> double {
        double sb;
        if (a.qscale) {
            // A bounds score / scale; allow for the fp32 roundings made while it was computed
            const double v = (double)A * sq;
            sb = v + 2e-6 * fabs(v) + 1e-6 * sqrt(qn) * sqrt(ncmax) + 1e-30;
        } else {
            sb = (double)A + a.eps_rel * sqrt(qn) * sqrt(ncmax) + 1e-30;
        }
        if (SPACE == HRS_SPACE_L2) return qn + ncmin - 2.0 * sb;
        if (SPACE == HRS_SPACE_IP) return 1.0 - sb;
        // (a zero query is at distance 1 from every row by definition: nothing can be proven, all rows tie)
        return qn > 0.0 ? 1.0 - sb / (sqrt(qn) * sqrt(sb > 0.0 ? ncmin : ncmax)) : 1.0;
    };
    double nq = 0.0;
    if (SPACE == HRS_SPACE_COSINE) {
        // the oracle's |q|^2 is a sequential fp64 sum: recompute it the same way (from the copy in LDS, four
        // elements per read: a loop over the row in memory costs a load latency per element)
        int d0 = 0;
        for (; d0 + 4 <= a.D; d0 += 4) {
            const float4 v = *(const float4*)(qs + d0);
            nq = fma((double)v.x, (double)v.x, nq);
            nq = fma((double)v.y, (double)v.y, nq);
            nq = fma((double)v.z, (double)v.z, nq);
            nq = fma((double)v.w, (double)v.w, nq);
        }
        for (; d0 < a.D; ++d0) nq = fma((double)qs[d0], (double)qs[d0], nq);
    }
    int done = 0;                       // candidates [0, done) are re-scored
    bool accept = false, defer = false;
    uint64_t* top = ek;                 // where the re-scored keys stand sorted
#ifdef HRS_SEL_PROF
    const long long pt_init = wall_clock64() - pt0;
#endif
    while (true) {
#ifdef HRS_SEL_PROF
        const long long pr0 = wall_clock64();
#endif
        // Re-score candidates [done, want): thread t owns candidate done + t (one fp64 chain per row, the
        // oracle's operation order).  The rows are fetched by the wave together -- eight lanes per row, a
        // 128-byte line per row and instruction -- into a per-wave LDS block and read back row-wise
        // (row stride 144 bytes: conflict-free), so every cache line is requested once, whole.
        {
            const int lane = tid & 63;
            const int idx = done + tid;
            const bool live = idx < want;
            // (a wave none of whose 64 candidates exists has nothing to fetch or to add up: its staging block is its own)
            const bool wave_live = __builtin_amdgcn_readfirstlane(done + (tid & ~63)) < want;
            const uint32_t grow = live ? (uint32_t)(s[idx] & 0xFFFFFFFFull) : 0u;
            // (a thread without a candidate still helps to fetch: it points at the first row)
            const float* cr = live ? a.C + (size_t)((int64_t)grow - a.row_offset) * a.D : a.C;
            // pointer of the row that this lane helps to fetch in instruction j: row (8j + lane/8) of the wave
            // (typed as global memory: a pointer rebuilt from integers would otherwise be loaded from with flat
            // instructions, which count as LDS operations too and would make every LDS wait a memory wait)
            typedef const __attribute__((address_space(1))) char* gptr;
            double acc = 0.0, nc = 0.0;
            if (wave_live) {
            gptr fp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int src = 8 * j + (lane >> 3);
                const uint64_t v = (uint64_t)cr;
                const uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
                fp[j] = (gptr)(((uint64_t)hi << 32) | lo);
            }
            char* wl = stage + (tid >> 6) * (64 * 144);
            char* wr_base = wl + (lane >> 3) * 144 + (lane & 7) * 16;
            const char* rd_base = wl + lane * 144;
            auto step4 = [&](const f32x4 cv, const float4 qv) __attribute__((always_inline)) {
                const double c0 = cv.x, c1 = cv.y, c2 = cv.z, c3 = cv.w;
                const double q0 = qv.x, q1 = qv.y, q2 = qv.z, q3 = qv.w;
                if (SPACE == HRS_SPACE_L2) {
                    const double e0 = q0 - c0, e1 = q1 - c1, e2 = q2 - c2, e3 = q3 - c3;
                    acc = fma(e0, e0, acc);
                    acc = fma(e1, e1, acc);
                    acc = fma(e2, e2, acc);
                    acc = fma(e3, e3, acc);
                } else {
                    acc = fma(q0, c0, acc);
                    acc = fma(q1, c1, acc);
                    acc = fma(q2, c2, acc);
                    acc = fma(q3, c3, acc);
                    if (SPACE == HRS_SPACE_COSINE) {
                        nc = fma(c0, c0, nc);
                        nc = fma(c1, c1, nc);
                        nc = fma(c2, c2, nc);
                        nc = fma(c3, c3, nc);
                    }
                }
            };
            // A window of SEL_PF 128-byte chunks per row is in flight (in registers) while one is consumed: a round
            // costs about one memory latency instead of one per chunk.  No request is conditional -- a piece beyond
            // the row's end is replaced by the row's last piece, which nobody reads back -- so the loop body has no
            // branch around a load and the compiler's waits stay counted.
            constexpr int SEL_PF = 3;
            const int nchunk = (a.D + 31) >> 5;
            const int last_piece = a.D * 4 - 16;
            f32x4 nxt[SEL_PF][8];
            auto fetch = [&](int c, f32x4 (&dst)[8]) __attribute__((always_inline)) {
                const int off = min(c * 128 + (lane & 7) * 16, last_piece);
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j] = *(const __attribute__((address_space(1))) f32x4*)(fp[j] + off);
            };
#pragma unroll
            for (int p = 0; p < SEL_PF; ++p) {
                fetch(p, nxt[p]);
                __builtin_amdgcn_sched_barrier(0);  // in this order: the first chunk is awaited first
            }
            for (int c0 = 0; c0 < nchunk; c0 += SEL_PF) {
#pragma unroll
                for (int p = 0; p < SEL_PF; ++p) {
                    const int c = c0 + p;
#pragma unroll
                    for (int j = 0; j < 8; ++j) *(f32x4*)(wr_base + j * (8 * 144)) = nxt[p][j];
                    fetch(c + SEL_PF, nxt[p]);
                    // only this wave touches its block: LDS operations of a wave complete in order
                    const int np = min(8, (a.D - c * 32) >> 2);
                    if (np == 8) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) step4(*(const f32x4*)(rd_base + u * 16), *(const float4*)(qs + c * 32 + 4 * u));
                    } else {
                        for (int u = 0; u < np; ++u) step4(*(const f32x4*)(rd_base + u * 16), *(const float4*)(qs + c * 32 + 4 * u));
                    }
                }
            }
            }  // waves with candidates
            if (live) {
                double d;
                if (SPACE == HRS_SPACE_L2)
                    d = acc;
                else if (SPACE == HRS_SPACE_IP)
                    d = 1.0 - acc;
                else
                    d = (nq == 0.0 || nc == 0.0) ? 1.0 : 1.0 - acc / (sqrt(nq) * sqrt(nc));
                ek[idx] = pack_key((float)d, grow);
            }
        }
#ifdef HRS_SEL_PROF
        const long long pr1 = wall_clock64();
        pt_resc += pr1 - pr0;
        pt_rounds += 1;
#endif
        done = want;
        if (tid == 0) need_s = 0;
        if (done <= 256) {
            // one re-scored key per thread: its rank among them by counting (keys are distinct), the sorted copy in
            // the second wave's staging block (free between rounds)
            __syncthreads();
            top = (uint64_t*)(stage + 64 * 144);
            const uint64_t mykey = tid < done ? ek[tid] : HRS_PAD_KEY;
            int rank = 0;
            for (int u = 0; u < done; ++u) rank += ek[u] < mykey;
            if (tid < done) top[rank] = mykey;
            __syncthreads();
        } else {
            int en = 256;
            while (en < done) en <<= 1;
            sort_keys(ek, en);  // (its barriers also publish need_s)
            top = ek;
        }
        if (overflow || done < a.k) break;
        const double dk = (double)orderable_f32((uint32_t)(top[a.k - 1] >> 32));
        auto could_beat = [&](uint64_t key) -> bool {
            const float A = orderable_f32(~(uint32_t)(key >> 32));
            const double dmin = dist_floor(A);
            return !(dk + 1e-6 * (1.0 + fabs(dmin)) < dmin);
        };
        if (!sorted) {
            // the list is partitioned, not sorted: is any candidate behind the re-scored front still in the race?
            int bad = 0;
            for (int t = done + tid; t < c; t += 256) bad |= could_beat(s[t]);
            if (bad) need_s = 1;
            __syncthreads();
            const int more = need_s;
            __syncthreads();
            if (more) {  // a second round: from here on the list is sorted (its first `done` are the re-scored ones)
                for (int t = c + tid; t < n2; t += 256) s[t] = HRS_PAD_KEY;
                sort_keys(s, n2);
                sorted = true;
                if (tid == 0) need_s = 0;
                __syncthreads();
            }
        }
        int need = done;
        if (sorted) {
            // recorded scores are sorted descending, so the candidates that could still beat dk are a prefix
            int mine = 0;
            for (int t = tid; t < c; t += 256)
                if (could_beat(s[t])) mine = t + 1;
            if (mine) atomicMax(&need_s, mine);
            __syncthreads();
            need = need_s;
            __syncthreads();
        }
        if (need <= done) {
            // and nothing that was never recorded (score <= tau) may beat it either
            const double dmin = dist_floor(a.tau[ql]);
            accept = dk + 1e-6 * (1.0 + fabs(dmin)) < dmin;
            break;
        }
        if (done >= RMAX) {
            defer = !BIG;  // the large variant may re-score four times as many
            break;
        }
        want = need < done + HRS_RESCORE ? need : done + HRS_RESCORE;
        if (want > RMAX) want = RMAX;
    }
#ifdef HRS_SEL_PROF
    if (tid == 0) {
        atomicAdd((unsigned long long*)&a.stats[12], (unsigned long long)pt_init);
        atomicAdd((unsigned long long*)&a.stats[13], (unsigned long long)pt_resc);
        atomicAdd((unsigned long long*)&a.stats[14], (unsigned long long)(wall_clock64() - pt0));
        atomicAdd((unsigned long long*)&a.stats[15], (unsigned long long)pt_rounds);
        atomicAdd((unsigned long long*)&a.stats[6], (unsigned long long)pt_load);
        atomicAdd((unsigned long long*)&a.stats[7], (unsigned long long)pt_sort);
    }
#endif
    if (tid == 0) {
        ok = accept;
        atomicMax((unsigned long long*)&a.stats[1], (unsigned long long)craw);
        if (overflow) atomicAdd((unsigned long long*)&a.stats[2], 1ull);
        if (done > a.first_rows) atomicAdd((unsigned long long*)&a.stats[3], 1ull);
        if (defer) {
            a.qlist2[atomicAdd(a.nflag2, 1)] = ql;
        } else if (!accept) {
            atomicAdd((unsigned long long*)&a.stats[0], 1ull);
            const int p = atomicAdd(a.nflag, 1);
            a.qlist[p] = q;
        }
    }
    __syncthreads();
    if (ok)
        for (int t = tid; t < a.k; t += 256) {
            const uint64_t key = top[t];
            a.keys_out[(size_t)q * a.k + t] = key;
            hrs_store_unpacked(key, (size_t)q * a.k + t, a.dist_out, a.rows_out);
        }
}

// The small variant: one workgroup per query.  The large one: a grid of at most one workgroup per CU walks the queue of
// deferred queries (device-side count, nearly always zero -- and a thousand workgroups of 100 KB of LDS each are slow
// to dispatch and retire beside a scan even when they leave at once).
template <int SPACE, bool BIG>
__global__ __launch_bounds__(256, BIG ? 1 : 2) void k_select_rescore(SelectArgs a) {  // (two small ones share a CU)
    if (!BIG) {
        select_rescore_one<SPACE, false>(a, blockIdx.x);
    } else {
        const int n = *a.nflag2;
        for (int slot = blockIdx.x; slot < n; slot += gridDim.x) {
            select_rescore_one<SPACE, true>(a, a.qlist2[slot]);
            __syncthreads();
        }
    }
}

template <int KS, int TR, bool I8, int QT, bool RS>
static int launch_scan_qt(const MfmaArgs& a, int mode, int device, hipStream_t st) {
    const size_t lds = scan_lds_bytes(TR * KS * 32, scan_pairs(QT) && !RS);
    static hrs_per_device_once attr;
    if (attr.need(device)) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_mfma_scan<KS, TR, 0, QT, I8, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_mfma_scan<KS, TR, 1, QT, I8, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_mfma_scan<KS, TR, 2, QT, I8, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.done(device);
    }
    if (mode == 1)
        hipLaunchKernelGGL((k_mfma_scan<KS, TR, 1, QT, I8, RS>), dim3(HRS_NUM_CU), dim3(512), lds, st, a);
    else if (mode == 2)
        hipLaunchKernelGGL((k_mfma_scan<KS, TR, 2, QT, I8, RS>), dim3(HRS_NUM_CU), dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL((k_mfma_scan<KS, TR, 0, QT, I8, RS>), dim3(HRS_NUM_CU), dim3(512), lds, st, a);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

template <int KS, int QG, int TR>
static int launch_scan16(MfmaArgs a, const char* qb16, const hrs_corpus* c, int device, hipStream_t st) {
    a.group = c->group_main;
    const size_t lds = scan_lds_bytes(TR * KS * 32, true);
    static hrs_per_device_once attr;
    if (attr.need(device)) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_mfma_scan16<KS, QG, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.done(device);
    }
    hipLaunchKernelGGL((k_mfma_scan16<KS, QG, TR>), dim3(HRS_NUM_CU), dim3(512), lds, st, a, qb16);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// Two 32-query tiles per wave (each corpus fragment read from LDS feeds two MFMAs) where the query
// fragments of 64 queries fit the wave's registers: the int8 shadow up to 12 k-steps (depth 384).
template <int KS, bool I8>
constexpr int scan_qt_max() { return (I8 && KS <= 12) ? 2 : 1; }

// qt: 1 one query tile per wave; 2 two; 3 two with the row split (batches of up to 256 queries)
template <int KS, int TR, bool I8>
static int launch_scan(MfmaArgs a, int mode, int qt, const hrs_corpus* c, int device, hipStream_t st) {
    a.group = mode == 0 ? c->group_main : c->group_pre;
    if constexpr (scan_qt_max<KS, I8>() == 2 && TR == 64) {
        if (qt == 3) return launch_scan_qt<KS, TR, I8, 2, true>(a, mode, device, st);
        if (qt == 2) return launch_scan_qt<KS, TR, I8, 2, false>(a, mode, device, st);
    }
    return launch_scan_qt<KS, TR, I8, 1, false>(a, mode, device, st);
}

template <int SPACE>
static int launch_select(const SelectArgs& s, int bc, int device, hipStream_t st) {
    const int lds_small = (HRS_CAND_CAP + HRS_RESCORE_MAX) * 8 + 1024 * 4 + 4 * 64 * 144;
    const int lds_big = (HRS_CAND_CAP_BIG + 4096) * 8 + 1024 * 4 + 4 * 64 * 144;
    static hrs_per_device_once attr;
    if (attr.need(device)) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_select_rescore<SPACE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_small));
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_select_rescore<SPACE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big));
        attr.done(device);
    }
    hipLaunchKernelGGL((k_select_rescore<SPACE, false>), dim3(bc), dim3(256), lds_small, st, s);
    // deferred queries (device-side count; workgroups beyond it leave at once)
    hipLaunchKernelGGL((k_select_rescore<SPACE, true>), dim3(bc < HRS_NUM_CU ? bc : HRS_NUM_CU), dim3(256), lds_big, st, s);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// Recorded right after the last scan launch of a call: the peer context's next scan waits for it.
static int scan_mark_done(hrs_ctx* ctx, hipStream_t st) {
    if (ctx->scan_done) {
        HRS_CHECK_HIP(hipEventRecord(ctx->scan_done, st));
        ctx->scan_recorded = true;
    }
    return HRS_OK;
}

int hrs_launch_mfma(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* mask,
                    uint64_t* keys_out, hipStream_t st, int phase, hrs_unpacked un) {
    const bool do_scan = phase != HRS_PHASE_TAIL, do_tail = phase != HRS_PHASE_SCAN;
    if (phase != HRS_PHASE_ALL && (!c->shadow || ctx->num_cu != HRS_NUM_CU || B > 1024)) {
        hrs_set_error("hrs_launch_mfma: a two-phase call needs a shadow and at most 1024 queries");
        return HRS_ERR_ARG;
    }
    if (!c->shadow || ctx->num_cu != HRS_NUM_CU) {
        // no matrix-core kernel for this shape: the exact scan gives the same answer
        return hrs_launch_exact(ctx, c, Q, B, k, space, mask, nullptr, nullptr, keys_out, st, nullptr, 0, -1, 0, un);
    }
    const bool i8 = c->kind == HRS_SHADOW_I8;
    const int Dpad = c->Dpad, KS = i8 ? Dpad / 32 : Dpad / 16;
    const bool wide = !i8 && Dpad == 1024;
    const int TR = i8 ? (KS <= 24 ? 64 : 32) : (Dpad <= 384 ? 64 : 32);
    const int BC = 1024;  // queries per chunk
    // workspace carve
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_ctrl = carve(256);
    const size_t o_qlist = carve((size_t)B * sizeof(int32_t));
    const size_t o_qb = carve((size_t)BC * Dpad * sizeof(__bf16));
    const size_t o_qb16 = carve((size_t)BC * Dpad);
    const size_t o_qn = carve((size_t)BC * sizeof(double));
    const size_t o_max = carve((size_t)BC * HRS_MAX_STREAMS * sizeof(float));
    const size_t o_tau = carve((size_t)BC * sizeof(float));
    const size_t o_tau2 = carve((size_t)BC * sizeof(float));
    const size_t o_cnt = carve((size_t)BC * MF_CNT_STRIDE * sizeof(int));
    const size_t o_qs = carve((size_t)BC * sizeof(float));
    const size_t o_kq = carve((size_t)BC * sizeof(float2));
    const size_t o_lad = carve((size_t)BC * sizeof(float));
    const size_t o_hist = carve((size_t)BC * 16 * sizeof(int));
    const size_t o_cand = carve((size_t)BC * HRS_CAND_CAP_BIG * sizeof(uint64_t));
    const size_t o_q2 = carve((size_t)BC * sizeof(int32_t) + 256);
    int rc = hrs_ws_reserve(ctx, &ctx->ws_mfma, off);
    if (rc) return rc;
    char* ws = (char*)ctx->ws_mfma.ptr;
    int32_t* nflag = (int32_t*)(ws + o_ctrl);
    int32_t* qlist = (int32_t*)(ws + o_qlist);

    const int64_t tiles = (c->N + TR - 1) / TR;
    // The tail of a two-phase call runs right behind its own main pass, beside the sampled passes of the scan that
    // waits for this one.  (HRS_TAIL_LATE=1 holds it until that scan has set its thresholds, so that it is queued
    // under that scan's main pass; it cannot share a SIMD with two scan waves -- 2 x 200 + 128 registers -- and when
    // it is made to fit, the main pass loses more than the tail takes by itself.)
    static const bool tail_late = getenv("HRS_TAIL_LATE") != nullptr;
    if (tail_late && phase == HRS_PHASE_TAIL && ctx->waiter && ctx->waiter->pre_recorded)
        HRS_CHECK_HIP(hipStreamWaitEvent(st, ctx->waiter->pre_done, 0));
    for (int qbase = 0; qbase < B; qbase += BC) {
        const int bc = B - qbase < BC ? B - qbase : BC;
        const int bpad = (wide && bc <= 128) ? 128 : (bc <= 256 ? 256 : (bc <= 512 ? 512 : 1024));
        // 64 queries per wave for large batches on the shallow int8 kernels
        // on the shallow int8 kernels: 64 queries per wave; batches of up to 256 split the rows of a tile as well
        const int qt = (i8 && KS <= 12 && !ctx->scan_qt1) ? (bpad >= 512 ? 2 : 3) : 1;
        // the main pass of the large int8 batches runs on the 16 x 16 x 64 MFMA shape (k_mfma_scan16)
        // (64 queries per wave up to depth 384 at batches of 512 and more; 32 queries per wave at depth 512: 3 % faster
        // there.  At depth 1024 -- instantiated below, reached with HRS_SCAN_SHAPE16_DEEP=1 -- the 32 x 32 x 32 kernel is
        // 5 % faster: 3.58 against 3.77 ms over 4M rows at batch 1024, top-100.)
        static const bool shape16_deep = getenv("HRS_SCAN_SHAPE16_DEEP") != nullptr;
        const bool shape16 = i8 && !ctx->scan_shape32 && ((qt == 2 && TR == 64 && KS % 2 == 0) || (qt == 1 && (KS == 16 || (KS == 32 && shape16_deep))));
        const int nqg = bpad / (wide ? 128 : MF_QPB * (qt == 2 ? 2 : 1));
        const int nslices = HRS_NUM_CU / nqg;
        // cleared by the convert kernel: this chunk's deferred-queue length; for the first chunk also the
        // fallback-queue length and the call's statistics
        ZeroList z;
        z.p[0] = (int32_t*)(ws + o_q2);
        z.n[0] = 1;
        z.p[1] = qbase == 0 ? nflag : nullptr;
        z.n[1] = 64;
        z.p[2] = qbase == 0 ? (int32_t*)ctx->stats_dev : nullptr;
        z.n[2] = 18;  // (32-bit words: the nine 64-bit counters 0..8)
        if (!do_scan) {
        } else if (i8)
            hipLaunchKernelGGL(k_convert_q_i8, dim3((bpad + 3) / 4), dim3(256), 0, st, Q + (size_t)qbase * c->D, bc, c->D,
                               bpad, Dpad, (int8_t*)(ws + o_qb), shape16 ? (int8_t*)(ws + o_qb16) : nullptr, (double*)(ws + o_qn),
                               (float*)(ws + o_qs), (float2*)(ws + o_kq), z);
        else
            hipLaunchKernelGGL(k_convert_q, dim3((bpad + 3) / 4), dim3(256), 0, st, Q + (size_t)qbase * c->D, bc, c->D,
                               bpad, Dpad, (__bf16*)(ws + o_qb), (double*)(ws + o_qn), z);
        // the scans of two contexts take turns -- but the conversion of the queries does not have to wait
        if (do_scan && qbase == 0 && ctx->peer) {
            if (ctx->peer->scan_recorded) HRS_CHECK_HIP(hipStreamWaitEvent(st, ctx->peer->scan_done, 0));
        }
        MfmaArgs a;
        a.shadow = (const char*)c->shadow;
        a.qb = (const char*)(ws + o_qb);
        a.group = nullptr;
        a.kq = i8 ? (const float2*)(ws + o_kq) : nullptr;
        a.N = c->N;
        a.tiles = tiles;
        a.row_offset = c->row_offset;
        a.nqg = nqg;
        a.nq = bc;
        const int64_t min_len = tiles / nslices;  // shortest slice, in tiles
        int ss = (int)(min_len / 4);  // at least four sampled tiles per slice
        static const int ss_max = getenv("HRS_SS_MAX") ? atoi(getenv("HRS_SS_MAX")) : HRS_SS_MAX;
        static const int j_min = getenv("HRS_J_MIN") ? atoi(getenv("HRS_J_MIN")) : HRS_J_MIN;
        if (ss > ss_max) ss = ss_max;
        // short slices (a shard of a split corpus): the sample is cheap against the rows its coarseness makes the main
        // pass record, which cost the same whatever the corpus' size
        if (min_len < 400 && ss > 24) ss = 24;
        if (ss < 1) ss = 1;
        a.ss = ss;
        a.maxima = (float*)(ws + o_max);
        a.tau = (const float*)(ws + o_tau);
        a.tau_out = (float*)(ws + o_tau2);
#ifdef HRS_SCAN_PROF
        a.prof = (long long*)ctx->stats_dev;
        a.prof_tau_shift = getenv("HRS_PROF_TAU_SHIFT") ? (float)atof(getenv("HRS_PROF_TAU_SHIFT")) : 0.f;
#endif
        a.cnt = (int*)(ws + o_cnt);
        a.cand = (uint64_t*)(ws + o_cand);
        a.mask = mask;
        a.lad_inv = (const float*)(ws + o_lad);
        a.hist = (int*)(ws + o_hist);
        // the ladder pass records its tiles and the main pass leaves them out where the main pass is the kernel that can
        // (HRS_LADDER_PLAIN: every main pass scores every tile, as the other kernels do)
        static const bool ladder_plain = getenv("HRS_LADDER_PLAIN") != nullptr;
        a.ladder_records = shape16 && !ladder_plain;
        a.work = (long long*)ctx->stats_dev + 8;
        if (do_scan) {  // (per chunk of queries, as the kernel's count; the tail half of a two-phase call leaves it alone)
            if (qbase == 0) ctx->last_tiles = 0;
            if (shape16) ctx->last_tiles += tiles;
        }
        const int M = nslices * 2 * (TR / 32);
        // aim at ~max(384, per_k * k) rows above tau; the expected count is ss * M * -ln(1 - j/M).  The int8 shadow
        // ranks by an upper bound that sits a third of a sigma above the score at depth 384 and half a sigma at depth
        // 1024: of the rows recorded one in four to seven scores above tau itself, and k of those are needed for the
        // proof, for every query of the batch -- hence 18k, and 27k for the deep rows.  (HRS_PER_K overrides.)
        static const int per_k_env = getenv("HRS_PER_K") ? atoi(getenv("HRS_PER_K")) : 0;
        const int per_k = per_k_env > 0 ? per_k_env : (i8 ? (Dpad > 512 ? 27 : 18) : 6);
        const int target = per_k * k > 384 ? per_k * k : 384;
        // lists are cut (thresholds raised) where the ladder estimates more than this many rows: a quarter of what a
        // list holds, and up to half of it where the target itself asks for more (the estimate is a coarse one)
        a.raise_limit = target * 3 / 2 < HRS_CAND_CAP_BIG / 4 ? HRS_CAND_CAP_BIG / 4
                      : (target * 3 / 2 < HRS_CAND_CAP_BIG / 2 ? target * 3 / 2 : HRS_CAND_CAP_BIG / 2);
        int j = (int)ceil((double)M * (1.0 - exp(-(double)target / ((double)ss * (double)M))));
        if (j < j_min) j = j_min;
        if (j > 3 * M / 4) j = 3 * M / 4;  // (the estimate of a low quantile of few maxima is a rough one)

        // sampled maxima -> first tau -> sampled ladder counts -> scan (tau raised where rows come in runs)
        auto tau_steps = [&](auto&& scan) -> int {
            int r = scan(1);
            if (r) return r;
            hipLaunchKernelGGL(k_pick_tau, dim3(bpad), dim3(256), 0, st, a.maxima, M, j, (float*)(ws + o_tau), a.cnt,
                               (float*)(ws + o_lad), (int*)(ws + o_hist), bc, i8 ? (const float*)(ws + o_qs) : nullptr);
            // the ladder only has to be right within a factor of two: half the sample of the first pass
            static const int ss2_env = getenv("HRS_SS2_MULT") ? atoi(getenv("HRS_SS2_MULT")) : 2;
            const int ss2 = (int64_t)ss2_env * ss * 4 <= min_len ? ss2_env * ss : ss;
            a.ss = ss2;
            a.raise_ss = ss2;
            r = scan(2);
            a.ss = ss;
            if (r) return r;
            if (phase == HRS_PHASE_SCAN && ctx->pre_done) {
                HRS_CHECK_HIP(hipEventRecord(ctx->pre_done, st));
                ctx->pre_recorded = true;
            }
            return HRS_OK;
        };
#ifdef HRS_SCAN_PROF
#define HRS_SCAN_PROF_RESET                                                    \
    (void)hipMemsetAsync(ctx->stats_dev + 8, 0xFF, 8, st);                     \
    (void)hipMemsetAsync(ctx->stats_dev + 9, 0, 8, st);                        \
    (void)hipMemsetAsync(ctx->stats_dev + 10, 0xFF, 8, st);                    \
    (void)hipMemsetAsync(ctx->stats_dev + 11, 0, 8, st);                       \
    (void)hipMemsetAsync(ctx->stats_dev + 12, 0, 32, st);
#else
#define HRS_SCAN_PROF_RESET
#endif
#define HRS_SCAN(KS_, TR_, I8_)                                                                      \
    case KS_: {                                                                                      \
        auto scan = [&](int mode) { return launch_scan<KS_, TR_, I8_>(a, mode, qt, c, ctx->device, st); };        \
        if ((rc = tau_steps(scan))) return rc;                                                       \
        HRS_SCAN_PROF_RESET                                                                          \
        hrs_prof_begin(ctx, st);                                                                     \
        if constexpr (I8_ && KS_ <= 12 && KS_ % 2 == 0 && TR_ == 64) {                               \
            rc = shape16 ? launch_scan16<KS_, 4, 64>(a, (const char*)(ws + o_qb16), c, ctx->device, st) : scan(0); \
        } else if constexpr (I8_ && (KS_ == 16 || KS_ == 32)) {                                      \
            rc = shape16 ? launch_scan16<KS_, 2, TR_>(a, (const char*)(ws + o_qb16), c, ctx->device, st) : scan(0); \
        } else {                                                                                     \
            rc = scan(0);                                                                            \
        }                                                                                            \
        hrs_prof_end(ctx, st);                                                                       \
        if (rc) return rc;                                                                           \
    } break;
        if (!do_scan) {
        } else if (wide) {
            static hrs_per_device_once attr;
            const int lds = 4 * 32 * 1024;
            if (attr.need(ctx->device)) {
                HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_mfma_scan_wide<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_mfma_scan_wide<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_mfma_scan_wide<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                attr.done(ctx->device);
            }
            auto scan = [&](int mode) {
                if (mode == 1)
                    hipLaunchKernelGGL(k_mfma_scan_wide<1>, dim3(HRS_NUM_CU), dim3(256), lds, st, a);
                else if (mode == 2)
                    hipLaunchKernelGGL(k_mfma_scan_wide<2>, dim3(HRS_NUM_CU), dim3(256), lds, st, a);
                else
                    hipLaunchKernelGGL(k_mfma_scan_wide<0>, dim3(HRS_NUM_CU), dim3(256), lds, st, a);
                return (int)HRS_OK;
            };
            if ((rc = tau_steps(scan))) return rc;
            hrs_prof_begin(ctx, st);
            scan(0);
            hrs_prof_end(ctx, st);
            HRS_CHECK_HIP(hipGetLastError());
        } else if (i8) {
            switch (KS) {
                HRS_SCAN(4, 64, true)
                HRS_SCAN(8, 64, true)
                HRS_SCAN(12, 64, true)
                HRS_SCAN(16, 64, true)
                HRS_SCAN(24, 64, true)
                HRS_SCAN(32, 32, true)
                default:
                    hrs_set_error("hrs_launch_mfma: no int8 kernel for padded depth %d", Dpad);
                    return HRS_ERR_UNSUPPORTED;
            }
        } else {
            switch (KS) {
                HRS_SCAN(8, 64, false)
                HRS_SCAN(16, 64, false)
                HRS_SCAN(24, 64, false)
                HRS_SCAN(32, 32, false)
                default:
                    hrs_set_error("hrs_launch_mfma: no kernel for padded depth %d", Dpad);
                    return HRS_ERR_UNSUPPORTED;
            }
        }
#undef HRS_SCAN
        if (do_scan && qbase + BC >= B && (rc = scan_mark_done(ctx, st))) return rc;
        if (!do_tail) continue;
        SelectArgs s;
        s.C = c->C;
        s.Q = Q;
        s.D = c->D;
        s.N = c->N;
        s.row_offset = c->row_offset;
        s.B = bc;
        s.k = k;
        s.cnt = a.cnt;
        s.cand = a.cand;
        s.tau = a.tau_out;
        s.qn2 = (const double*)(ws + o_qn);
        s.norm_stats = c->norm_stats;
        s.eps_rel = MF_EPS_BASE + MF_EPS_PER_D * Dpad;
        s.qscale = i8 ? (const float*)(ws + o_qs) : nullptr;
        s.keys_out = keys_out;
        s.dist_out = un.dist;
        s.rows_out = un.rows;
        s.qlist = qlist;
        s.nflag = nflag;
        s.stats = (long long*)ctx->stats_dev;
        s.qbase = qbase;
        {
            static const int first_env = getenv("HRS_SEL_FIRST") ? atoi(getenv("HRS_SEL_FIRST")) : 0;
            // (whole waves of 64: a wave without candidates skips the round)
            int first = first_env > 0 ? first_env : ((8 * k + 32 + 63) & ~63);
            s.first_rows = first > HRS_RESCORE ? HRS_RESCORE : first;
        }
        // a proof needs about a quarter of the target re-scored (the rows whose bound reaches the k-th exact score); for
        // deep rows and a large k that is often more than the small variant's prefix, and what it has re-scored by
        // then (4 KB a row) is read again by the large one
        s.large_first = i8 && Dpad > 512 && per_k * k / 4 > HRS_RESCORE_MAX / 2;
        s.qlist2 = (int32_t*)(ws + o_q2) + 64;
        s.nflag2 = (int32_t*)(ws + o_q2);
        if (space == HRS_SPACE_L2)
            rc = launch_select<HRS_SPACE_L2>(s, bc, ctx->device, st);
        else if (space == HRS_SPACE_COSINE)
            rc = launch_select<HRS_SPACE_COSINE>(s, bc, ctx->device, st);
        else
            rc = launch_select<HRS_SPACE_IP>(s, bc, ctx->device, st);
        if (rc) return rc;
        HRS_CHECK_HIP(hipGetLastError());
    }
    if (!do_tail) return HRS_OK;
    // queries that could not be proven take the exact scan (device-side count; usually zero)
    return hrs_launch_exact(ctx, c, Q, B, k, space, mask, qlist, nflag, keys_out, st, nullptr, 0, -1, 0, un);
}
