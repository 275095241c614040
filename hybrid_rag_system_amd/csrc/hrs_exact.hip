// Exact streaming scan + top-k (the small-batch path and the fallback of the MFMA path).
//
// Every (query, row) distance is accumulated in fp64 from the fp32 inputs in index order, one
// correctly rounded operation per step, then rounded once to fp32 -- the same operations the CPU
// oracle performs -- so the (distance, row) order is exact by construction.  The kernel is bound
// by the HBM stream of the fp32 corpus: one workgroup per CU walks a contiguous slice of rows;
// 256-row x 32-float segments (32 KB) arrive in a 4-deep LDS ring by LDS-DMA (three segments in
// flight per CU, counted vmcnt, one barrier per segment).  A thread owns one row of the tile: the
// per-row fp64 fma chain is latency bound, so a segment is wide in rows (256 independent chains, one
// wave on every SIMD) and short in depth.  The LDS image is lane-linear as the DMA requires; bank
// conflicts of the per-row reads are avoided by permuting the 16-byte pieces of a row on the SOURCE
// side (piece p of row r is stored at p ^ ESWZ(r)).  Nothing in the loop touches global memory
// except the DMA: the pass's queries are copied to LDS once, candidates live in LDS.
#include "hrs_internal.h"

#ifndef ER
#define ER 256   // rows per tile
#endif
#ifndef EDC
#define EDC 32   // floats per row segment
#endif
#ifndef ENBUF
#define ENBUF 4  // ring depth
#endif
#define EPP (EDC / 4)             // 16-byte pieces per row segment = DMA lanes per row
#define ERPI (64 / EPP)           // rows per DMA instruction
#define ESWZ(r) (((r) / (16 / EPP)) & (EPP - 1))  // piece p of row r is stored at p ^ ESWZ(r)
#define ESEG_BYTES (ER * EDC * 4)
#define EIPW (ESEG_BYTES / 4096)  // LDS-DMA instructions per wave per segment (1 KB each, 4 waves)
#define ESLOTS (256 / ER)         // query slots per row (threads that share a row, one query subset each)
#define EQ_FLOATS 2048  // fp32 query staging: BQ * D <= 2048

struct ExactArgs {
    const float* C;
    int64_t N;
    int D;
    int64_t row_offset;
    const float* Q;
    const int32_t* qlist;
    const int32_t* nq_dev;
    int nq;
    int k;
    const uint8_t* mask;
    int64_t mask_qstride;   // 0: one mask for all queries; else query q reads mask[q * mask_qstride + row]
    int mask_eq;            // < 0: a row is kept when its mask byte is non-zero; else when it equals this value
    const uint64_t* after;  // nullable, per query: only keys strictly above it are eligible (a search resumed behind its last result)
    uint64_t* partial;  // [nq][gridDim.x][k]
    int64_t tiles;
    const double* qd;   // SQ kernels: the queries as fp64, [nq][D] (written by k_queries_f64 before the scan)
    // LV kernels: one list per (query, class).  Row r belongs to class level[q * level_pitch + r] of query q; this launch
    // serves the classes lv0 .. lv0 + nlv - 1 of `levels`, and partial is [nq][levels][gridDim.x][k]
    const uint8_t* level;
    int64_t level_pitch;  // a multiple of 4, >= N
    int lv0, nlv, levels;
};
#define ELV_MAX 64  // classes one launch of an LV kernel serves (nlv * k <= HRS_EXACT_CB - ER as well)

// fp32 -> fp64 copy of the queries for the scan kernels that take them through the scalar cache
__global__ void k_queries_f64(const float* __restrict__ Q, int64_t n, double* __restrict__ qd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) qd[i] = (double)Q[i];
}

// SQ: the queries' values come from a.qd through the scalar cache instead of from an fp64 copy in LDS.  With four
// queries per pass the LDS copy costs 64 of the 72 LDS reads per thread and 32-float segment -- every lane reads the
// same 16 bytes, but a read instruction takes its share of the LDS pipe all the same -- and held the scan at 4.1 TB/s
// where the same loop without them streams 5.8.
// EXT: the forms only the extended entry points use (a resume key per query, a mask per query, a mask compared for
// equality with a value).  They are compiled out of the plain scan: their conditional global loads at the end of a tile
// make the compiler place a vmcnt(0) in front of the loop's counted wait, which drains the DMA ring at every segment.
// LV: the nearest k rows of EVERY class of every query in one pass over the corpus (the full-corpus hybrid mode: the
// class of a row is the number of the query's keywords it holds).  A list in LDS holds the candidates of all classes as
// (class : 8 | distance : 32 | row within this workgroup's slice : 24), so that one sort groups them by class; what a
// class keeps is its first k, and each (query, class) has its own threshold (the distance half of a key: rows that tie
// with it are admitted and the sort settles them).  The class bytes of a tile come in by LDS-DMA with the tile's
// first segment -- 256 bytes per query, one 4-byte piece per lane -- so the loop still holds no load of the compiler's.
template <int SPACE, int BQ, bool SQ = false, bool EXT = false, bool LV = false>
__global__ __launch_bounds__(256, 1) void k_exact_scan(ExactArgs a) {
    static_assert(!(EXT && LV), "a scan by classes takes no resume key and no other mask");
    typedef const __attribute__((address_space(4))) double* qd_ptr;
    constexpr int NQT = BQ > ESLOTS ? BQ / ESLOTS : 1;  // queries per thread
    constexpr int NSL = BQ < ESLOTS ? BQ : ESLOTS;      // query slots in use
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: ring [ENBUF][ESEG_BYTES] | qf32 [EQ_FLOATS] | qbuf [2][BQ][EDC] double | cand [BQ][CB] | misc
    char* ring = smem;
    float* qf32 = (float*)(smem + ENBUF * ESEG_BYTES);
    double* qbuf = (double*)(qf32 + EQ_FLOATS);
    uint64_t* cand = (uint64_t*)(qbuf + 2 * BQ * EDC);
    uint64_t* tau = cand + BQ * HRS_EXACT_CB;
    double* qnorm = (double*)(tau + BQ);
    uint64_t* aft = (uint64_t*)(qnorm + BQ);
    int* cnt = (int*)(aft + BQ);
    int* qidx = cnt + BQ;
    // LV: [ENBUF][BQ][ER] class bytes of a tile (requested with the tile's first segment, up to ENBUF - 1 tiles ahead)
    uint8_t* lvb = (uint8_t*)(qidx + BQ);
    uint32_t* tau32 = (uint32_t*)(lvb + ENBUF * BQ * ER);  // LV: [BQ][ELV_MAX]
    int* lstart = (int*)(tau32 + BQ * ELV_MAX);        // LV: [ELV_MAX] where a class begins in the sorted list

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row_t = tid & (ER - 1);   // this thread's row inside a tile
    const int qslot = tid / ER;
    int nq = a.nq_dev ? *a.nq_dev : a.nq;
    if (nq > a.nq) nq = a.nq;
    if (nq <= 0) return;

    const int D = a.D;
    const int nchunks = (D + EDC - 1) / EDC;
    const int64_t tile_begin = a.tiles * blockIdx.x / gridDim.x;
    const int64_t tile_end = a.tiles * (blockIdx.x + 1) / gridDim.x;
    const int64_t steps = (tile_end - tile_begin) * nchunks;
    const int npass = (nq + BQ - 1) / BQ;
    const uint32_t ring0 = lds_addr_of(ring);

    for (int pass = 0; pass < npass; ++pass) {
        __syncthreads();
        if (tid < BQ) {
            int j = pass * BQ + tid;
            if (j >= nq) j = pass * BQ;
            qidx[tid] = a.qlist ? a.qlist[j] : j;
            cnt[tid] = 0;
            tau[tid] = HRS_PAD_KEY;
            if (EXT) aft[tid] = a.after ? a.after[qidx[tid]] : 0ull;
        }
        if (LV)
            for (int e = tid; e < BQ * ELV_MAX; e += 256) tau32[e] = 0xFFFFFFFFu;
        __syncthreads();
        for (int e = tid; e < BQ * D; e += 256) qf32[e] = a.Q[(size_t)qidx[e / D] * D + e % D];
        __syncthreads();
        if (SPACE == HRS_SPACE_COSINE && tid < BQ) {
            const float* q = qf32 + tid * D;
            double s = 0.0;
            for (int i = 0; i < D; ++i) s = fma((double)q[i], (double)q[i], s);
            qnorm[tid] = s;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the compiler's is in flight past here
        qd_ptr qrow[NQT];  // SQ: this thread's queries (wave-uniform: a wave lies within one query slot)
#pragma unroll
        for (int b = 0; b < NQT; ++b)
            qrow[b] = SQ ? (qd_ptr)(uintptr_t)uniform_ptr(a.qd + (size_t)__builtin_amdgcn_readfirstlane(qidx[qslot + NSL * b]) * D) : (qd_ptr)(uintptr_t)0;

        auto stage_q = [&](int ch, int buf) {  // LDS fp32 -> LDS fp64, this chunk of every query of the pass
            if (SQ) return;
            const int w = min(EDC, D - ch * EDC);
            for (int e = tid; e < BQ * EDC; e += 256) {
                const int b = e / EDC, i = e % EDC;
                qbuf[(buf * BQ + b) * EDC + i] = i < w ? (double)qf32[b * D + ch * EDC + i] : 0.0;
            }
        };
        // per-lane source offsets of this wave's 8 pieces for a full-width segment of a full tile
        int fastoff[EIPW];
#pragma unroll
        for (int i = 0; i < EIPW; ++i) {
            const int r = wv * (ER / 4) + i * ERPI + lane / EPP;
            fastoff[i] = (r * D + (((lane & (EPP - 1)) ^ ESWZ(r)) << 2)) * 4;
        }
        auto issue_dma = [&](int64_t step) {
            if (step >= steps) step = steps - 1;  // keep the vmcnt bookkeeping uniform at the tail
            const int64_t tile = tile_begin + step / nchunks;
            const int ch = (int)(step % nchunks);
            const int wpieces = min(EDC, D - ch * EDC) >> 2;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)(step % ENBUF) * ESEG_BYTES + wv * (EIPW * 1024));
            const char* base = uniform_ptr(a.C + tile * ER * (int64_t)D + ch * EDC);
            const int64_t rows_left = a.N - tile * ER;
            if (LV && ch == 0 && wv < BQ) {
                // this wave's query: the class bytes of the tile's rows (requested ahead of the segment, so they have
                // landed when it has; lanes past the end of the byte row repeat its last piece)
                const int64_t left = a.level_pitch - tile * ER;
                const int voff = (int)min((int64_t)lane * 4, left - 4);
                const char* lb = uniform_ptr(a.level + (int64_t)__builtin_amdgcn_readfirstlane(qidx[wv]) * a.level_pitch + tile * ER);
                glds4_asm(lb, voff, __builtin_amdgcn_readfirstlane(lds_addr_of(lvb) + (uint32_t)(((tile % ENBUF) * BQ + wv) * ER)));
            }
            if (wpieces == EDC / 4 && rows_left >= ER) {
#pragma unroll
                for (int i = 0; i < EIPW; ++i) glds16_asm(base, fastoff[i], dst + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < EIPW; ++i) {
                    int r = wv * (ER / 4) + i * ERPI + lane / EPP;
                    const int p = lane & (EPP - 1);
                    int sp = p ^ ESWZ(r);
                    if (sp >= wpieces) sp = 0;
                    if (r >= rows_left) r = (int)rows_left - 1;
                    glds16_asm(base, (r * D + sp * 4) * 4, dst + i * 1024);
                }
            }
        };

        double acc[NQT];
        double ncacc = 0.0;
        if (steps > 0) {
            stage_q(0, 0);
#pragma unroll
            for (int g = 0; g < ENBUF - 1; ++g) issue_dma(g);
        }
        for (int64_t step = 0; step < steps; ++step) {
            const int ch = (int)(step % nchunks);
            // segments step+1 .. step+ENBUF-2 may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ENBUF - 2) * EIPW) : "memory");
            __syncthreads();
            issue_dma(step + ENBUF - 1);
            if (step + 1 < steps) stage_q((int)((step + 1) % nchunks), (int)((step + 1) & 1));
            if (ch == 0) {
#pragma unroll
                for (int b = 0; b < NQT; ++b) acc[b] = 0.0;
                ncacc = 0.0;
            }
            if (qslot < NSL) {
                const int pieces = min(EDC, D - ch * EDC) >> 2;
                const char* rowp = ring + (step % ENBUF) * ESEG_BYTES + row_t * (EDC * 4);
                const double* qb = qbuf + (step & 1) * BQ * EDC;
                const int sw = ESWZ(row_t);
                auto piece = [&](int s) {
                    const float4 cv = *(const float4*)(rowp + ((s ^ sw) << 4));
                    const double c0 = (double)cv.x, c1 = (double)cv.y, c2 = (double)cv.z, c3 = (double)cv.w;
                    if (SPACE == HRS_SPACE_COSINE) {
                        ncacc = fma(c0, c0, ncacc);
                        ncacc = fma(c1, c1, ncacc);
                        ncacc = fma(c2, c2, ncacc);
                        ncacc = fma(c3, c3, ncacc);
                    }
#pragma unroll
                    for (int b = 0; b < NQT; ++b) {
                        const double* qp = qb + (qslot + NSL * b) * EDC + s * 4;
                        double q0, q1, q2, q3;
                        if (SQ) {
                            // wave-uniform values through the scalar cache: operands of the fp64 instructions straight
                            // from scalar registers
                            const qd_ptr qs = qrow[b] + ch * EDC + s * 4;
                            q0 = qs[0], q1 = qs[1], q2 = qs[2], q3 = qs[3];
                        } else {
                            q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
                        }
                        if (SPACE == HRS_SPACE_L2) {
                            const double d0 = q0 - c0, d1 = q1 - c1, d2 = q2 - c2, d3 = q3 - c3;
                            acc[b] = fma(d0, d0, acc[b]);
                            acc[b] = fma(d1, d1, acc[b]);
                            acc[b] = fma(d2, d2, acc[b]);
                            acc[b] = fma(d3, d3, acc[b]);
                        } else {
                            acc[b] = fma(q0, c0, acc[b]);
                            acc[b] = fma(q1, c1, acc[b]);
                            acc[b] = fma(q2, c2, acc[b]);
                            acc[b] = fma(q3, c3, acc[b]);
                        }
                    }
                };
                if (SQ && pieces == EDC / 4) {
                    // Full segment, queries through the scalar cache.  Left alone, the compiler requests all 256 scalar
                    // registers' worth of a segment at once and parks them in vector lanes; here two pieces of four
                    // values per query are live: piece s + 1 is requested behind the point where piece s is known to
                    // have arrived and before it is used.  Scalar loads return in any order, so the compiler's wait for
                    // them is lgkmcnt(0) -- hence the row itself is read from LDS up front.  The loads are ordinary
                    // loads through the constant address space (hand-written s_load statements are not safe: the
                    // compiler may copy their destination registers before a hand-written wait); the empty asm that
                    // takes piece s as an input is what places the compiler's wait.
                    typedef double f64x4 __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(4))) f64x4* qd4_ptr;
                    float4 cv[EDC / 4];
#pragma unroll
                    for (int s = 0; s < EDC / 4; ++s) cv[s] = *(const float4*)(rowp + ((s ^ sw) << 4));
                    f64x4 nxt[NQT];
#pragma unroll
                    for (int b = 0; b < NQT; ++b) nxt[b] = *(qd4_ptr)(qrow[b] + ch * EDC);
#pragma unroll
                    for (int s = 0; s < EDC / 4; ++s) {
                        f64x4 cur[NQT];
#pragma unroll
                        for (int b = 0; b < NQT; ++b) {
                            cur[b] = nxt[b];
                            asm volatile("" ::"s"(cur[b]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (s + 1 < EDC / 4) {
#pragma unroll
                            for (int b = 0; b < NQT; ++b) nxt[b] = *(qd4_ptr)(qrow[b] + ch * EDC + (s + 1) * 4);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const double c0 = (double)cv[s].x, c1 = (double)cv[s].y, c2 = (double)cv[s].z, c3 = (double)cv[s].w;
                        if (SPACE == HRS_SPACE_COSINE) {
                            ncacc = fma(c0, c0, ncacc);
                            ncacc = fma(c1, c1, ncacc);
                            ncacc = fma(c2, c2, ncacc);
                            ncacc = fma(c3, c3, ncacc);
                        }
#pragma unroll
                        for (int b = 0; b < NQT; ++b) {
                            const double q0 = cur[b][0], q1 = cur[b][1], q2 = cur[b][2], q3 = cur[b][3];
                            if (SPACE == HRS_SPACE_L2) {
                                const double d0 = q0 - c0, d1 = q1 - c1, d2 = q2 - c2, d3 = q3 - c3;
                                acc[b] = fma(d0, d0, acc[b]);
                                acc[b] = fma(d1, d1, acc[b]);
                                acc[b] = fma(d2, d2, acc[b]);
                                acc[b] = fma(d3, d3, acc[b]);
                            } else {
                                acc[b] = fma(q0, c0, acc[b]);
                                acc[b] = fma(q1, c1, acc[b]);
                                acc[b] = fma(q2, c2, acc[b]);
                                acc[b] = fma(q3, c3, acc[b]);
                            }
                        }
                    }
                } else if (pieces == EDC / 4) {
                    // full segment: unrolled, so the LDS reads of all 16 pieces are in flight together
#pragma unroll
                    for (int s = 0; s < EDC / 4; ++s) piece(s);
                } else {
                    for (int s = 0; s < pieces; ++s) piece(s);
                }
            }
            if (ch == nchunks - 1) {
                const int64_t tile = tile_begin + step / nchunks;
                const int64_t row = tile * ER + row_t;
                bool live = row < a.N && qslot < NSL;
                if (EXT) {
                    if (live && a.mask && a.mask_qstride == 0) live = a.mask_eq < 0 ? a.mask[row] != 0 : a.mask[row] == a.mask_eq;
                } else if (!LV) {
                    if (live && a.mask) live = a.mask[row] != 0;
                }
                if (live) {
                    const uint32_t grow = (uint32_t)(row + a.row_offset);
#pragma unroll
                    for (int b = 0; b < NQT; ++b) {
                        const int qb_i = qslot + NSL * b;
                        int lv = 0;
                        if (LV) {
                            lv = (int)lvb[((tile % ENBUF) * BQ + qb_i) * ER + row_t] - a.lv0;
                            if ((unsigned)lv >= (unsigned)a.nlv) continue;  // a class this launch does not serve
                        }
                        if (EXT && a.mask_qstride) {  // a mask per query (rows grouped by their keyword match count)
                            const uint8_t mv = a.mask[(int64_t)qidx[qb_i] * a.mask_qstride + row];
                            if (a.mask_eq < 0 ? mv == 0 : mv != a.mask_eq) continue;
                        }
                        double d;
                        if (SPACE == HRS_SPACE_L2) {
                            d = acc[b];
                        } else if (SPACE == HRS_SPACE_IP) {
                            d = 1.0 - acc[b];
                        } else {
                            const double nqv = qnorm[qb_i];
                            d = (nqv == 0.0 || ncacc == 0.0) ? 1.0 : 1.0 - acc[b] / (sqrt(nqv) * sqrt(ncacc));
                        }
                        if (LV) {
                            const uint32_t d32 = f32_orderable((float)d);
                            if (d32 <= tau32[qb_i * ELV_MAX + lv]) {
                                const int pos = atomicAdd(&cnt[qb_i], 1);
                                cand[qb_i * HRS_EXACT_CB + pos] = ((uint64_t)lv << 56) | ((uint64_t)d32 << 24) | (uint32_t)(row - tile_begin * ER);
                            }
                            continue;
                        }
                        const uint64_t key = pack_key((float)d, grow);
                        if (key < tau[qb_i] && (!EXT || !a.after || key > aft[qb_i])) {
                            const int pos = atomicAdd(&cnt[qb_i], 1);
                            cand[qb_i * HRS_EXACT_CB + pos] = key;  // pos < CB: cnt <= CB - ER before a tile
                        }
                    }
                }
                if (!LV && a.mask) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // compact any list that could overflow during the next tile
                for (int b = 0; b < BQ; ++b) {
                    const int c = cnt[b];
                    if (c > HRS_EXACT_CB - ER) {
                        uint64_t* cb = cand + b * HRS_EXACT_CB;
                        for (int i = c + tid; i < HRS_EXACT_CB; i += 256) cb[i] = HRS_PAD_KEY;
                        bitonic_sort_u64(cb, HRS_EXACT_CB, tid, 256);
                        if (LV) {
                            // sorted by (class, distance, row): every class keeps its first k, the k-th sets its
                            // threshold, the rest become padding and a second sort moves them behind what is kept
                            for (int i = tid; i < c; i += 256) {
                                const int l = (int)(cb[i] >> 56);
                                if (i == 0 || (int)(cb[i - 1] >> 56) != l) lstart[l] = i;
                            }
                            if (tid == 0) cnt[b] = 0;
                            __syncthreads();
                            uint64_t nk[HRS_EXACT_CB / 256];
#pragma unroll
                            for (int u = 0; u < HRS_EXACT_CB / 256; ++u) {
                                const int i = tid + u * 256;
                                const uint64_t key = cb[i];
                                bool keep = false;
                                if (i < c) {
                                    const int l = (int)(key >> 56), r = i - lstart[l];
                                    keep = r < a.k;
                                    if (r == a.k - 1) tau32[b * ELV_MAX + l] = (uint32_t)(key >> 24);
                                }
                                nk[u] = keep ? key : HRS_PAD_KEY;
                                const int kept = __builtin_popcountll(__builtin_amdgcn_ballot_w64(keep));
                                if (lane == 0 && kept) atomicAdd(&cnt[b], kept);
                            }
#pragma unroll
                            for (int u = 0; u < HRS_EXACT_CB / 256; ++u) cb[tid + u * 256] = nk[u];
                            bitonic_sort_u64(cb, HRS_EXACT_CB, tid, 256);
                            continue;
                        }
                        if (tid == 0) {
                            cnt[b] = min(c, a.k);
                            tau[b] = c >= a.k ? cb[a.k - 1] : HRS_PAD_KEY;
                        }
                        __syncthreads();
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // final: sort each list, emit k keys
        for (int b = 0; b < BQ; ++b) {
            const int j = pass * BQ + b;
            if (j >= nq) break;
            const int c = cnt[b];
            uint64_t* cb = cand + b * HRS_EXACT_CB;
            for (int i = c + tid; i < HRS_EXACT_CB; i += 256) cb[i] = HRS_PAD_KEY;
            bitonic_sort_u64(cb, HRS_EXACT_CB, tid, 256);
            if (LV) {
                for (int l = tid; l < a.nlv; l += 256) lstart[l] = -1;
                __syncthreads();
                for (int i = tid; i < c; i += 256) {
                    const int l = (int)(cb[i] >> 56);
                    if (i == 0 || (int)(cb[i - 1] >> 56) != l) lstart[l] = i;
                }
                __syncthreads();
                // slot (class, rank) is written by one thread: the class's rank-th entry, or padding
                for (int t = tid; t < a.nlv * a.k; t += 256) {
                    const int l = t / a.k, r = t % a.k, at = lstart[l] + r;
                    uint64_t key = HRS_PAD_KEY;
                    if (lstart[l] >= 0 && at < c && (int)(cb[at] >> 56) == l) {
                        const uint64_t e = cb[at];
                        key = (((e >> 24) & 0xFFFFFFFFull) << 32) | (uint32_t)(tile_begin * ER + (int64_t)(e & 0xFFFFFFull) + a.row_offset);
                    }
                    a.partial[(((size_t)j * a.levels + a.lv0 + l) * gridDim.x + blockIdx.x) * a.k + r] = key;
                }
                __syncthreads();
                continue;
            }
            uint64_t* out = a.partial + ((size_t)j * gridDim.x + blockIdx.x) * a.k;
            for (int i = tid; i < a.k; i += 256) out[i] = cb[i];
            __syncthreads();
        }
    }
}

// Merge: each block takes `per` consecutive lists of k keys of one query and writes their best k.
// in: [nq][nlists][k]; out: [nq][ceil(nlists/per)][k].  When dst_rows != null (final level, one
// group) the result goes to final_out[dst_rows[q]] instead (dst_rows may be the identity = null).
struct MergeArgs {
    const uint64_t* in;
    uint64_t* out;
    const int32_t* nq_dev;
    int nq;
    int nlists;
    int per;
    int k;
    const int32_t* qlist;  // final level only: scatter to out[qlist[q]]
    int final_level;
    int64_t out_stride;    // final level: keys of query q go to out + q * out_stride (k when the lists are dense)
    float* dist_out;       // final level, nullable: the keys unpacked, [nq][k] ...
    int64_t* rows_out;     // ... (only with out_stride == k)
};

__global__ __launch_bounds__(256) void k_merge_lists(MergeArgs a) {
    __shared__ uint64_t s[2048];
    int nq = a.nq_dev ? min(*a.nq_dev, a.nq) : a.nq;
    const int ngroups = (a.nlists + a.per - 1) / a.per;
    const int q = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    if (q >= nq) return;
    const int l0 = g * a.per, l1 = min(a.nlists, l0 + a.per);
    const int n = (l1 - l0) * a.k;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    const uint64_t* src = a.in + ((size_t)q * a.nlists + l0) * a.k;
    for (int i = threadIdx.x; i < n2; i += 256) s[i] = i < n ? src[i] : HRS_PAD_KEY;
    bitonic_sort_u64(s, n2, threadIdx.x, 256);
    uint64_t* dst;
    if (a.final_level)
        dst = a.out + (size_t)(a.qlist ? a.qlist[q] : q) * a.out_stride;
    else
        dst = a.out + ((size_t)q * ngroups + g) * a.k;
    for (int i = threadIdx.x; i < a.k; i += 256) {
        const uint64_t key = i < n2 ? s[i] : HRS_PAD_KEY;
        dst[i] = key;
        if (a.final_level) hrs_store_unpacked(key, (size_t)(a.qlist ? a.qlist[q] : q) * a.k + i, a.dist_out, a.rows_out);
    }
}

// Both levels in one launch, one workgroup per query: the groups of `per` lists one after the other, then their
// winners.  For the queue of queries that the matrix-core path could not prove (device-side count, nearly always
// zero): the launch saved matters on every call, the serial groups only for a query that is actually queued.
__global__ __launch_bounds__(256) void k_merge_lists_serial(MergeArgs a) {
    __shared__ uint64_t s[2048], top[2048];
    const int nq = a.nq_dev ? min(*a.nq_dev, a.nq) : a.nq;
    const int q = blockIdx.x;
    if (q >= nq) return;
    const int ngroups = (a.nlists + a.per - 1) / a.per;  // at most 2048 / k
    for (int g = 0; g < ngroups; ++g) {
        const int l0 = g * a.per, l1 = min(a.nlists, l0 + a.per);
        const int n = (l1 - l0) * a.k;
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        const uint64_t* src = a.in + ((size_t)q * a.nlists + l0) * a.k;
        for (int i = threadIdx.x; i < n2; i += 256) s[i] = i < n ? src[i] : HRS_PAD_KEY;
        bitonic_sort_u64(s, n2, threadIdx.x, 256);
        for (int i = threadIdx.x; i < a.k; i += 256) top[g * a.k + i] = i < n2 ? s[i] : HRS_PAD_KEY;
        __syncthreads();
    }
    const int n = ngroups * a.k;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 256) s[i] = i < n ? top[i] : HRS_PAD_KEY;
    bitonic_sort_u64(s, n2, threadIdx.x, 256);
    const size_t qo = (size_t)(a.qlist ? a.qlist[q] : q);
    for (int i = threadIdx.x; i < a.k; i += 256) {
        const uint64_t key = i < n2 ? s[i] : HRS_PAD_KEY;
        a.out[qo * a.out_stride + i] = key;
        hrs_store_unpacked(key, qo * a.k + i, a.dist_out, a.rows_out);
    }
}

template <int SPACE, int BQ, bool SQ = false, bool EXT = false, bool LV = false>
static int launch_exact_t(const ExactArgs& a, int grid, int device, hipStream_t st) {
    size_t lds = ENBUF * ESEG_BYTES + EQ_FLOATS * sizeof(float) + 2 * BQ * EDC * sizeof(double) +
                 BQ * HRS_EXACT_CB * sizeof(uint64_t) + BQ * (2 * sizeof(uint64_t) + sizeof(double) + 2 * sizeof(int)) + 64;
    if (LV) lds += ENBUF * BQ * ER + BQ * ELV_MAX * sizeof(uint32_t) + ELV_MAX * sizeof(int);
    static hrs_per_device_once attr;  // the attribute belongs to the function on ONE device
    if (attr.need(device)) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_exact_scan<SPACE, BQ, SQ, EXT, LV>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.done(device);
    }
    hipLaunchKernelGGL((k_exact_scan<SPACE, BQ, SQ, EXT, LV>), dim3(grid), dim3(256), lds, st, a);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}
template <int SPACE>
static int launch_exact_lv(const ExactArgs& a, int grid, int bq, int device, hipStream_t st) {
    if (bq >= 4 && a.qd) return launch_exact_t<SPACE, 4, true, false, true>(a, grid, device, st);
    if (bq >= 4) return launch_exact_t<SPACE, 4, false, false, true>(a, grid, device, st);
    if (bq >= 2) return launch_exact_t<SPACE, 2, false, false, true>(a, grid, device, st);
    return launch_exact_t<SPACE, 1, false, false, true>(a, grid, device, st);
}

template <int SPACE, bool EXT>
static int launch_exact_e(const ExactArgs& a, int grid, int bq, int device, hipStream_t st) {
    if (bq >= 4 && a.qd) return launch_exact_t<SPACE, 4, true, EXT>(a, grid, device, st);
    if (bq >= 4) return launch_exact_t<SPACE, 4, false, EXT>(a, grid, device, st);
    if (bq >= 2) return launch_exact_t<SPACE, 2, false, EXT>(a, grid, device, st);
    return launch_exact_t<SPACE, 1, false, EXT>(a, grid, device, st);
}
template <int SPACE>
static int launch_exact_s(const ExactArgs& a, int grid, int bq, int device, hipStream_t st) {
    if (a.after || a.mask_qstride || a.mask_eq >= 0) return launch_exact_e<SPACE, true>(a, grid, bq, device, st);
    return launch_exact_e<SPACE, false>(a, grid, bq, device, st);
}

// Exact top-k of queries Q[qlist[0..nq)] (or Q[0..B) when qlist is null); nq may live on the device
// (nq_dev), in which case B is the capacity.  Results go to keys_out[q] (q = qlist entry or index).
// p0: the scratch of the call -- per-workgroup lists [B][grid][k], then the first merge level
// [B][ceil(grid/16)][k]; every word that is read has been written by this call.
static int exact_run(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* mask,
                     const int32_t* qlist, const int32_t* nq_dev, uint64_t* keys_out, uint64_t* p0, int grid, bool timed,
                     hipStream_t st, const uint64_t* after = nullptr, int64_t mask_qstride = 0, int mask_eq = -1,
                     int64_t out_stride = 0, hrs_unpacked un = hrs_unpacked(), double* qd = nullptr) {
    const int64_t tiles = (c->N + ER - 1) / ER;
    const int per = 16;
    const int g1 = (grid + per - 1) / per;
    uint64_t* p1 = p0 + (size_t)B * grid * k;
    int rc = HRS_OK;

    ExactArgs a;
    a.C = c->C;
    a.N = c->N;
    a.D = c->D;
    a.row_offset = c->row_offset;
    a.Q = Q;
    a.qlist = qlist;
    a.nq_dev = nq_dev;
    a.nq = B;
    a.k = k;
    a.mask = mask;
    a.mask_qstride = mask_qstride;
    a.mask_eq = mask_eq;
    a.after = after;
    a.partial = p0;
    a.tiles = tiles;
    a.level = nullptr;
    a.level_pitch = 0;
    a.lv0 = a.nlv = a.levels = 0;
    int bq = nq_dev ? 4 : B;
    if (bq * c->D > EQ_FLOATS) bq = EQ_FLOATS / c->D;  // the pass's queries are staged in LDS as fp32
    // passes of four queries take the queries' values through the scalar cache, from an fp64 copy (qd: room for B x D
    // doubles; not for the queue of the matrix-core path, whose scan must cost nothing when the queue is empty)
    a.qd = nullptr;
    if (qd && !nq_dev && !qlist && bq >= 4 && c->N > 0) {
        const int64_t n = (int64_t)B * c->D;
        hipLaunchKernelGGL(k_queries_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Q, n, qd);
        a.qd = qd;
    }
    if (bq < 1) {
        hrs_set_error("exact scan: D=%d exceeds %d", c->D, EQ_FLOATS);
        return HRS_ERR_UNSUPPORTED;
    }
    if (c->N > 0) {
        if (timed) hrs_prof_begin(ctx, st);
        if (space == HRS_SPACE_L2)
            rc = launch_exact_s<HRS_SPACE_L2>(a, grid, bq, ctx->device, st);
        else if (space == HRS_SPACE_COSINE)
            rc = launch_exact_s<HRS_SPACE_COSINE>(a, grid, bq, ctx->device, st);
        else
            rc = launch_exact_s<HRS_SPACE_IP>(a, grid, bq, ctx->device, st);
        if (timed) hrs_prof_end(ctx, st);
        if (rc) return rc;
    } else {
        HRS_CHECK_HIP(hipMemsetAsync(p0, 0xFF, (size_t)B * grid * k * sizeof(uint64_t), st));
    }
    MergeArgs m;
    m.nq_dev = nq_dev;
    m.nq = B;
    m.k = k;
    m.per = per;
    m.qlist = qlist;
    m.out_stride = out_stride ? out_stride : k;
    m.dist_out = m.out_stride == k ? un.dist : nullptr;
    m.rows_out = m.out_stride == k ? un.rows : nullptr;
    if (g1 > 1 && nq_dev && (int64_t)g1 * k <= 2048) {
        m.in = p0;
        m.out = keys_out;
        m.nlists = grid;
        m.final_level = 1;
        hipLaunchKernelGGL(k_merge_lists_serial, dim3(B), dim3(256), 0, st, m);
    } else if (g1 > 1) {
        m.in = p0;
        m.out = p1;
        m.nlists = grid;
        m.final_level = 0;
        hipLaunchKernelGGL(k_merge_lists, dim3(B * g1), dim3(256), 0, st, m);
        m.in = p1;
        m.out = keys_out;
        m.nlists = g1;
        m.final_level = 1;
        hipLaunchKernelGGL(k_merge_lists, dim3(B), dim3(256), 0, st, m);
    } else {
        m.in = p0;
        m.out = keys_out;
        m.nlists = grid;
        m.final_level = 1;
        hipLaunchKernelGGL(k_merge_lists, dim3(B), dim3(256), 0, st, m);
    }
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

static int exact_grid(const hrs_ctx* ctx, const hrs_corpus* c) {
    const int64_t tiles = (c->N + ER - 1) / ER;
    return (int)(tiles < ctx->num_cu ? (tiles > 0 ? tiles : 1) : ctx->num_cu);
}
static size_t exact_scratch_words(int B, int grid, int k) {
    return (size_t)B * grid * k + (size_t)B * ((grid + 15) / 16) * k;
}

int hrs_launch_exact(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* mask,
                     const int32_t* qlist, const int32_t* nq_dev, uint64_t* keys_out, hipStream_t st, const uint64_t* after,
                     int64_t mask_qstride, int mask_eq, int64_t out_stride, hrs_unpacked un) {
    if (B <= 0) return HRS_OK;
    const int grid = exact_grid(ctx, c);
    const size_t words = exact_scratch_words(B, grid, k), qd_words = nq_dev ? 0 : (size_t)B * c->D;
    int rc = hrs_ws_reserve(ctx, &ctx->ws_partial, (words + qd_words) * sizeof(uint64_t));
    if (rc) return rc;
    uint64_t* p0 = (uint64_t*)ctx->ws_partial.ptr;
    return exact_run(ctx, c, Q, B, k, space, mask, qlist, nq_dev, keys_out, p0, grid, !nq_dev, st, after, mask_qstride, mask_eq,
                     out_stride, un, qd_words ? (double*)(p0 + words) : nullptr);
}

// The nearest k rows of every class 0 .. levels-1 of every query (level[q * pitch + row] = the class of a row for query
// q) -> keys_out [B][levels][k].  One pass over the corpus per four queries and per group of classes whose lists fit the
// kernel's buffer (256 / k classes, at most ELV_MAX), then the merges of the plain scan with (query, class) in the
// place of the query.  Returns HRS_ERR_UNSUPPORTED (nothing launched) where the kernel's layout does not apply.
int hrs_launch_exact_levels(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* level,
                            int64_t pitch, int levels, uint64_t* keys_out, hipStream_t st) {
    const int grid = exact_grid(ctx, c);
    const int64_t tiles = (c->N + ER - 1) / ER;
    const int64_t slice_rows = ((tiles + grid - 1) / grid + 1) * ER;
    if (pitch % 4 != 0 || k > HRS_EXACT_CB - ER || slice_rows > (1 << 24) || c->N <= 0 || ((uintptr_t)level & 3)) return HRS_ERR_UNSUPPORTED;
    int per = (HRS_EXACT_CB - ER) / k;
    if (per > ELV_MAX) per = ELV_MAX;
    const int BL = B * levels;
    const int g1 = (grid + 15) / 16;
    const size_t words = exact_scratch_words(BL, grid, k), qd_words = (size_t)B * c->D;
    int rc = hrs_ws_reserve(ctx, &ctx->ws_partial, (words + qd_words) * sizeof(uint64_t));
    if (rc) return rc;
    uint64_t* p0 = (uint64_t*)ctx->ws_partial.ptr;
    uint64_t* p1 = p0 + (size_t)BL * grid * k;
    ExactArgs a;
    a.C = c->C;
    a.N = c->N;
    a.D = c->D;
    a.row_offset = c->row_offset;
    a.Q = Q;
    a.qlist = nullptr;
    a.nq_dev = nullptr;
    a.nq = B;
    a.k = k;
    a.mask = nullptr;
    a.mask_qstride = 0;
    a.mask_eq = -1;
    a.after = nullptr;
    a.partial = p0;
    a.tiles = tiles;
    a.level = level;
    a.level_pitch = pitch;
    a.levels = levels;
    int bq = B;
    if (bq * c->D > EQ_FLOATS) bq = EQ_FLOATS / c->D;
    if (bq < 1) return HRS_ERR_UNSUPPORTED;
    a.qd = nullptr;
    if (bq >= 4) {
        const int64_t n = (int64_t)B * c->D;
        double* qd = (double*)(p0 + words);
        hipLaunchKernelGGL(k_queries_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Q, n, qd);
        a.qd = qd;
    }
    hrs_prof_begin(ctx, st);
    for (int lv0 = 0; lv0 < levels; lv0 += per) {
        a.lv0 = lv0;
        a.nlv = levels - lv0 < per ? levels - lv0 : per;
        if (space == HRS_SPACE_L2)
            rc = launch_exact_lv<HRS_SPACE_L2>(a, grid, bq, ctx->device, st);
        else if (space == HRS_SPACE_COSINE)
            rc = launch_exact_lv<HRS_SPACE_COSINE>(a, grid, bq, ctx->device, st);
        else
            rc = launch_exact_lv<HRS_SPACE_IP>(a, grid, bq, ctx->device, st);
        if (rc) return rc;
    }
    hrs_prof_end(ctx, st);
    MergeArgs m;
    m.nq_dev = nullptr;
    m.nq = BL;
    m.k = k;
    m.per = 16;
    m.qlist = nullptr;
    m.out_stride = k;
    m.dist_out = nullptr;
    m.rows_out = nullptr;
    m.in = p0;
    m.nlists = grid;
    if (g1 > 1) {
        m.out = p1;
        m.final_level = 0;
        hipLaunchKernelGGL(k_merge_lists, dim3(BL * g1), dim3(256), 0, st, m);
        m.in = p1;
        m.nlists = g1;
    }
    m.out = keys_out;
    m.final_level = 1;
    hipLaunchKernelGGL(k_merge_lists, dim3(BL), dim3(256), 0, st, m);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// ---- the entry point that allocates nothing ------------------------------------------------------
// Its scratch (what hrs_launch_exact takes from a lazily grown workspace) is a file-scope device array: every
// GPU the library runs on holds its own copy at its own address until the process ends.  The array's name is
// only a host-side stand-in; the address on a device comes from hipGetSymbolAddress with that device current
// (hrs_exact_static_init, called by hrs_ctx_create) and only that pointer is ever handed to a kernel.
#define HRS_STATIC_GRID_MAX 256
#define HRS_STATIC_WORDS ((size_t)HRS_STATIC_MAX_B * HRS_MAX_K * (HRS_STATIC_GRID_MAX + HRS_STATIC_GRID_MAX / 16))
static __device__ __align__(256) unsigned long long hrs_static_scratch[HRS_STATIC_WORDS];  // 4.25 MiB
#define HRS_MAX_DEVICES 64
static uint64_t* g_static_ptr[HRS_MAX_DEVICES];  // per device: where hrs_static_scratch lives there

int hrs_exact_static_init(int device) {
    if (device < 0 || device >= HRS_MAX_DEVICES) return HRS_OK;  // such a device simply has no static entry point
    if (g_static_ptr[device]) return HRS_OK;
    void* p = nullptr;
    HRS_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(hrs_static_scratch)));  // the caller made `device` current
    g_static_ptr[device] = (uint64_t*)p;
    return HRS_OK;
}

extern "C" int hrs_score_topk_static(hrs_ctx* ctx, hrs_corpus* c, const float* Q_dev, int B, int k, int space,
                                     const uint8_t* row_mask_dev, uint64_t* keys_dev, float* dist_dev,
                                     int64_t* rows_dev, void* stream) {
    HRS_REQUIRE(ctx && c, "hrs_score_topk_static: null handle");
    HRS_REQUIRE(keys_dev, "hrs_score_topk_static: keys_dev is required (the library holds no result buffer here)");
    HRS_REQUIRE(B >= 0 && (B == 0 || Q_dev), "hrs_score_topk_static: B=%d with null queries", B);
    HRS_REQUIRE(B <= HRS_STATIC_MAX_B, "hrs_score_topk_static: B=%d exceeds the static scratch (at most %d queries per call)", B,
                HRS_STATIC_MAX_B);
    HRS_REQUIRE(k >= 1 && k <= HRS_MAX_K, "hrs_score_topk_static: k=%d outside [1, %d]", k, HRS_MAX_K);
    HRS_REQUIRE(space >= 0 && space <= 2, "hrs_score_topk_static: unknown space %d", space);
    if (B == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    const int grid = exact_grid(ctx, c);
    uint64_t* scratch = ctx->device < HRS_MAX_DEVICES ? g_static_ptr[ctx->device] : nullptr;
    if (!scratch || grid > HRS_STATIC_GRID_MAX || exact_scratch_words(B, grid, k) > HRS_STATIC_WORDS) {
        hrs_set_error("hrs_score_topk_static: no static scratch for this device (device %d, %d workgroups)", ctx->device, grid);
        return HRS_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    hrs_unpacked un;
    un.dist = dist_dev;
    un.rows = rows_dev;
    return exact_run(ctx, c, Q_dev, B, k, space, row_mask_dev, nullptr, nullptr, keys_dev, scratch, grid, false, st, nullptr, 0, -1, 0, un);
}
