// Exact streaming scan + top-k (the small-batch path and the fallback of the MFMA path).
//
// Every (query, row) distance is accumulated in fp64 from the fp32 inputs in index order, one
// correctly rounded operation per step, then rounded once to fp32 -- the same operations the CPU
// oracle performs -- so the (distance, row) order is exact by construction.  The kernel is bound
// by the HBM stream of the fp32 corpus: one workgroup per CU walks a contiguous slice of rows;
// 256-row x 64-float segments are brought into LDS by LDS-DMA (global_load_lds, 16 B per lane),
// double buffered, and each thread owns one row of the tile.  The LDS image is lane-linear as the
// DMA requires; bank conflicts of the per-row reads are avoided by permuting the 16-byte pieces
// of a row on the SOURCE side (piece p of row r is stored at p ^ (r & 15)).
#include "hrs_internal.h"

#define ER 256   // rows per tile (one per thread)
#define EDC 64   // floats per row segment
#define ETILE_BYTES (ER * EDC * 4)

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
    uint64_t* partial;  // [nq][gridDim.x][k]
    int64_t tiles;
};

template <int SPACE, int BQ>
__global__ __launch_bounds__(256, 1) void k_exact_scan(ExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [2][ETILE_BYTES] tiles | [2][BQ][EDC] double q-chunks | [BQ][CB] keys | misc
    char* tilebuf = smem;
    double* qbuf = (double*)(smem + 2 * ETILE_BYTES);
    uint64_t* cand = (uint64_t*)(qbuf + 2 * BQ * EDC);
    uint64_t* tau = cand + BQ * HRS_EXACT_CB;
    double* qnorm = (double*)(tau + BQ);
    int* cnt = (int*)(qnorm + BQ);
    int* qidx = cnt + BQ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nq = a.nq_dev ? *a.nq_dev : a.nq;
    if (nq > a.nq) nq = a.nq;
    if (nq <= 0) return;

    const int D = a.D;
    const int nchunks = (D + EDC - 1) / EDC;
    const int64_t tile_begin = a.tiles * blockIdx.x / gridDim.x;
    const int64_t tile_end = a.tiles * (blockIdx.x + 1) / gridDim.x;
    const int64_t steps = (tile_end - tile_begin) * nchunks;
    const int npass = (nq + BQ - 1) / BQ;

    for (int pass = 0; pass < npass; ++pass) {
        __syncthreads();
        if (tid < BQ) {
            int j = pass * BQ + tid;
            if (j >= nq) j = pass * BQ;
            qidx[tid] = a.qlist ? a.qlist[j] : j;
            cnt[tid] = 0;
            tau[tid] = HRS_PAD_KEY;
        }
        __syncthreads();
        if (SPACE == HRS_SPACE_COSINE && tid < BQ) {
            const float* q = a.Q + (size_t)qidx[tid] * D;
            double s = 0.0;
            for (int i = 0; i < D; ++i) s = fma((double)q[i], (double)q[i], s);
            qnorm[tid] = s;
        }

        auto stage_q = [&](int ch, int buf) {
            // BQ x EDC doubles for the chunk; 256 threads
            const int w = min(EDC, D - ch * EDC);
            for (int e = tid; e < BQ * EDC; e += 256) {
                int b = e / EDC, i = e % EDC;
                double v = 0.0;
                if (i < w) v = (double)a.Q[(size_t)qidx[b] * D + ch * EDC + i];
                qbuf[(buf * BQ + b) * EDC + i] = v;
            }
        };
        auto issue_dma = [&](int64_t step, int buf) {
            const int64_t tile = tile_begin + step / nchunks;
            const int ch = (int)(step % nchunks);
            const int wpieces = min(EDC, D - ch * EDC) >> 2;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr_of(tilebuf) + buf * ETILE_BYTES + wv * 16384);
            // scalar base = first row of the tile at this chunk; per-lane offsets stay below 2^31:
            // 256 rows x D x 4 bytes <= 8 MB
            const char* base = uniform_ptr(a.C + tile * ER * (int64_t)D + ch * EDC);
            const int64_t rows_left = a.N - tile * ER;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int r = wv * 64 + i * 4 + (lane >> 4);
                int p = lane & 15;
                int sp = p ^ (r & 15);
                if (sp >= wpieces) sp = 0;
                if (r >= rows_left) r = (int)rows_left - 1;
                glds16_asm(base, (r * D + sp * 4) * 4, dst + i * 1024);
            }
        };

        double acc[BQ];
        double ncacc = 0.0;
        if (steps > 0) {
            stage_q(0, 0);
            issue_dma(0, 0);
        }
        for (int64_t step = 0; step < steps; ++step) {
            const int cur = (int)(step & 1);
            const int ch = (int)(step % nchunks);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step + 1 < steps) {
                issue_dma(step + 1, cur ^ 1);
                stage_q((int)((step + 1) % nchunks), cur ^ 1);
            }
            if (ch == 0) {
#pragma unroll
                for (int b = 0; b < BQ; ++b) acc[b] = 0.0;
                ncacc = 0.0;
            }
            const int pieces = min(EDC, D - ch * EDC) >> 2;
            const char* rowp = tilebuf + cur * ETILE_BYTES + tid * (EDC * 4);
            const double* qb = qbuf + cur * BQ * EDC;
            const int sw = tid & 15;
            for (int s = 0; s < pieces; ++s) {
                float4 cv = *(const float4*)(rowp + ((s ^ sw) << 4));
                double c0 = (double)cv.x, c1 = (double)cv.y, c2 = (double)cv.z, c3 = (double)cv.w;
                if (SPACE == HRS_SPACE_COSINE) {
                    ncacc = fma(c0, c0, ncacc);
                    ncacc = fma(c1, c1, ncacc);
                    ncacc = fma(c2, c2, ncacc);
                    ncacc = fma(c3, c3, ncacc);
                }
#pragma unroll
                for (int b = 0; b < BQ; ++b) {
                    const double* qp = qb + b * EDC + s * 4;
                    double q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
                    if (SPACE == HRS_SPACE_L2) {
                        double d0 = q0 - c0, d1 = q1 - c1, d2 = q2 - c2, d3 = q3 - c3;
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
            if (ch == nchunks - 1) {
                const int64_t tile = tile_begin + step / nchunks;
                const int64_t row = tile * ER + tid;
                bool live = row < a.N;
                if (live && a.mask) live = a.mask[row] != 0;
                if (live) {
                    const uint32_t grow = (uint32_t)(row + a.row_offset);
#pragma unroll
                    for (int b = 0; b < BQ; ++b) {
                        double d;
                        if (SPACE == HRS_SPACE_L2) {
                            d = acc[b];
                        } else if (SPACE == HRS_SPACE_IP) {
                            d = 1.0 - acc[b];
                        } else {
                            double nqv = qnorm[b];
                            d = (nqv == 0.0 || ncacc == 0.0) ? 1.0 : 1.0 - acc[b] / (sqrt(nqv) * sqrt(ncacc));
                        }
                        uint64_t key = pack_key((float)d, grow);
                        if (key < tau[b]) {
                            int pos = atomicAdd(&cnt[b], 1);
                            cand[b * HRS_EXACT_CB + pos] = key;  // pos < CB: cnt <= CB-ER before a tile
                        }
                    }
                }
                __syncthreads();
                // compact any list that could overflow during the next tile
                for (int b = 0; b < BQ; ++b) {
                    int c = cnt[b];
                    if (c > HRS_EXACT_CB - ER) {
                        uint64_t* cb = cand + b * HRS_EXACT_CB;
                        for (int i = c + tid; i < HRS_EXACT_CB; i += 256) cb[i] = HRS_PAD_KEY;
                        bitonic_sort_u64(cb, HRS_EXACT_CB, tid, 256);
                        if (tid == 0) {
                            cnt[b] = min(c, a.k);
                            tau[b] = c >= a.k ? cb[a.k - 1] : HRS_PAD_KEY;
                        }
                        __syncthreads();
                    }
                }
            }
        }
        __syncthreads();
        // final: sort each list, emit k keys
        for (int b = 0; b < BQ; ++b) {
            int j = pass * BQ + b;
            if (j >= nq) break;
            int c = cnt[b];
            uint64_t* cb = cand + b * HRS_EXACT_CB;
            for (int i = c + tid; i < HRS_EXACT_CB; i += 256) cb[i] = HRS_PAD_KEY;
            bitonic_sort_u64(cb, HRS_EXACT_CB, tid, 256);
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
        dst = a.out + (size_t)(a.qlist ? a.qlist[q] : q) * a.k;
    else
        dst = a.out + ((size_t)q * ngroups + g) * a.k;
    for (int i = threadIdx.x; i < a.k; i += 256) dst[i] = i < n2 ? s[i] : HRS_PAD_KEY;
}

template <int SPACE, int BQ>
static int launch_exact_t(const ExactArgs& a, int grid, hipStream_t st) {
    size_t lds = 2 * ETILE_BYTES + 2 * BQ * EDC * sizeof(double) + BQ * HRS_EXACT_CB * sizeof(uint64_t) +
                 BQ * (sizeof(uint64_t) + sizeof(double) + 2 * sizeof(int)) + 64;
    static bool attr_set = false;
    if (!attr_set) {
        HRS_CHECK_HIP(hipFuncSetAttribute((const void*)k_exact_scan<SPACE, BQ>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_exact_scan<SPACE, BQ>), dim3(grid), dim3(256), lds, st, a);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

template <int SPACE>
static int launch_exact_s(const ExactArgs& a, int grid, int bq, hipStream_t st) {
    if (bq >= 4) return launch_exact_t<SPACE, 4>(a, grid, st);
    if (bq >= 2) return launch_exact_t<SPACE, 2>(a, grid, st);
    return launch_exact_t<SPACE, 1>(a, grid, st);
}

// Exact top-k of queries Q[qlist[0..nq)] (or Q[0..B) when qlist is null); nq may live on the device
// (nq_dev), in which case B is the capacity.  Results go to keys_out[q] (q = qlist entry or index).
int hrs_launch_exact(hrs_ctx* ctx, hrs_corpus* c, const float* Q, int B, int k, int space, const uint8_t* mask,
                     const int32_t* qlist, const int32_t* nq_dev, uint64_t* keys_out, hipStream_t st) {
    if (B <= 0) return HRS_OK;
    const int64_t tiles = (c->N + ER - 1) / ER;
    int grid = (int)(tiles < ctx->num_cu ? (tiles > 0 ? tiles : 1) : ctx->num_cu);
    // ping-pong: partial [B][grid][k] then [B][ceil(grid/16)][k]
    const int per = 16;
    const int g1 = (grid + per - 1) / per;
    size_t need = ((size_t)B * grid * k + (size_t)B * g1 * k) * sizeof(uint64_t);
    int rc = hrs_ws_reserve(ctx, &ctx->ws_partial, need);
    if (rc) return rc;
    uint64_t* p0 = (uint64_t*)ctx->ws_partial.ptr;
    uint64_t* p1 = p0 + (size_t)B * grid * k;

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
    a.partial = p0;
    a.tiles = tiles;
    const int bq = nq_dev ? 4 : B;
    if (c->N > 0) {
        if (!nq_dev) hrs_prof_begin(ctx, st);
        if (space == HRS_SPACE_L2)
            rc = launch_exact_s<HRS_SPACE_L2>(a, grid, bq, st);
        else if (space == HRS_SPACE_COSINE)
            rc = launch_exact_s<HRS_SPACE_COSINE>(a, grid, bq, st);
        else
            rc = launch_exact_s<HRS_SPACE_IP>(a, grid, bq, st);
        if (!nq_dev) hrs_prof_end(ctx, st);
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
    if (g1 > 1) {
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
