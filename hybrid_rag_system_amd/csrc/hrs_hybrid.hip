// Keyword presence counts over forward-CSR postings, score fusion + stable order, cross-shard merge.
#include <new>

#include "hrs_internal.h"

// One wave per (query, candidate): the 64 lanes read the row's term list coalesced, 64 terms at a time, and every
// query term (wave-uniform, at most 64 per round) is compared with all of them at once: one compare and one ballot
// per query term and block, the terms found collected as bits of a scalar mask.  Integer work only.
__global__ __launch_bounds__(256) void k_keyword_counts(const int64_t* __restrict__ doc_offs,
                                                        const int32_t* __restrict__ term_ids, int64_t N,
                                                        int64_t row_offset, const int32_t* __restrict__ q_terms,
                                                        const int32_t* __restrict__ q_term_offs,
                                                        const int64_t* __restrict__ cand_rows, int B, int c,
                                                        int32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)B * c) return;
    const int b = (int)(pair / c);
    int64_t row = cand_rows[pair] - row_offset;
    int n = 0;
    if (cand_rows[pair] >= 0 && row >= 0 && row < N) {
        const int64_t p0 = doc_offs[row], p1 = doc_offs[row + 1];
        const int t0 = q_term_offs[b], t1 = q_term_offs[b + 1];
        for (int tb = t0; tb < t1; tb += 64) {
            const int nt = min(64, t1 - tb);
            const int my = lane < nt ? q_terms[tb + lane] : -1;  // the query terms of this pass over them, one per lane
            uint64_t found = 0;                                    // bit j: query term tb + j occurs in the row
            for (int64_t p = p0; p < p1; p += 64) {
                const bool have = p + lane < p1;  // lanes beyond the row compare nothing (a query term may be any int32)
                const int dt = have ? term_ids[p + lane] : 0;
                for (int j = 0; j < nt; ++j) {
                    const int qt = __builtin_amdgcn_readlane(my, j);
                    if (__builtin_amdgcn_ballot_w64(have && dt == qt) != 0) found |= 1ull << j;
                }
            }
            n += __builtin_popcountll(found);
        }
    }
    if (lane == 0) counts[pair] = n;
}

extern "C" int hrs_keyword_counts(hrs_ctx* ctx, hrs_postings* p, const int32_t* q_terms_dev,
                                  const int32_t* q_term_offs_dev, const int64_t* cand_rows_dev, int B, int c,
                                  int32_t* counts_dev, void* stream) {
    HRS_REQUIRE(ctx && p && q_term_offs_dev && cand_rows_dev && counts_dev, "hrs_keyword_counts: null argument");
    HRS_REQUIRE(B >= 0 && c >= 0, "hrs_keyword_counts: negative size");
    if ((int64_t)B * c == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    const int64_t pairs = (int64_t)B * c;
    hipLaunchKernelGGL(k_keyword_counts, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       p->doc_offs, p->term_ids, p->N, p->row_offset, q_terms_dev, q_term_offs_dev, cand_rows_dev, B,
                       c, counts_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// BM25 keyword score of every (query, candidate) pair, normalised to [0, 1): one wave per pair.  Lane j keeps the
// occurrence count of query term j in the row (an integer: ballots over the row's term list, 64 entries at a time);
// lane 0 then adds the terms' contributions in query order, every operation a separately rounded fp64 one:
//   score = sum_j idf[t_j] * tf_j * (k1 + 1) / (tf_j + k1 * (1 - b + b * len / avg_len)),  kw = score / sum_j idf[t_j] * (k1 + 1)
// (the denominator is the score of a row that holds every term infinitely often).  Terms outside [0, V) weigh nothing.
__global__ __launch_bounds__(256) void k_bm25_scores(const int64_t* __restrict__ doc_offs, const int32_t* __restrict__ term_ids,
                                                     int64_t N, int64_t row_offset, const double* __restrict__ idf, int64_t V,
                                                     double avg_len, double k1, double bb, const int32_t* __restrict__ q_terms,
                                                     const int32_t* __restrict__ q_term_offs, const int64_t* __restrict__ cand_rows,
                                                     int B, int c, double* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)B * c) return;
    const int b = (int)(pair / c);
    const int64_t row = cand_rows[pair] - row_offset;
    double score = 0.0, best = 0.0;
    if (cand_rows[pair] >= 0 && row >= 0 && row < N) {
        const int64_t p0 = doc_offs[row], p1 = doc_offs[row + 1];
        const int t0 = q_term_offs[b], t1 = q_term_offs[b + 1];
        const double norm = __dmul_rn(k1, __dadd_rn(__dsub_rn(1.0, bb), __dmul_rn(bb, __ddiv_rn((double)(p1 - p0), avg_len))));
        const double k1p = __dadd_rn(k1, 1.0);
        for (int tb = t0; tb < t1; tb += 64) {
            const int nt = min(64, t1 - tb);
            const int my = lane < nt ? q_terms[tb + lane] : -1;
            int tf = 0;  // lane j: occurrences of query term tb + j
            for (int64_t p = p0; p < p1; p += 64) {
                const bool have = p + lane < p1;
                const int dt = have ? term_ids[p + lane] : 0;
                for (int j = 0; j < nt; ++j) {
                    const int qt = __builtin_amdgcn_readlane(my, j);
                    const int hits = __builtin_popcountll(__builtin_amdgcn_ballot_w64(have && dt == qt));
                    if (lane == j) tf += hits;
                }
            }
            // lane j works out the contribution of term j (the division: once per wave, not once per term), then the
            // contributions are added in query order; a term that is absent or outside the vocabulary adds an exact 0.0
            double wk = 0.0, contrib = 0.0;
            if (my >= 0 && my < V) {
                const double w = idf[my];
                wk = __dmul_rn(w, k1p);
                if (tf > 0) contrib = __ddiv_rn(__dmul_rn(__dmul_rn(w, (double)tf), k1p), __dadd_rn((double)tf, norm));
            }
            for (int j = 0; j < nt; ++j) {  // (wave-uniform sums; lane 0's copy is the one kept)
                best = __dadd_rn(best, __shfl(wk, j));
                score = __dadd_rn(score, __shfl(contrib, j));
            }
        }
    }
    if (lane == 0) scores[pair] = best > 0.0 ? __ddiv_rn(score, best) : 0.0;
}

extern "C" int hrs_bm25_scores(hrs_ctx* ctx, hrs_postings* p, const double* idf_dev, int64_t V, double avg_len, double k1, double b,
                               const int32_t* q_terms_dev, const int32_t* q_term_offs_dev, const int64_t* cand_rows_dev, int B,
                               int c, double* scores_dev, void* stream) {
    HRS_REQUIRE(ctx && p && idf_dev && q_term_offs_dev && cand_rows_dev && scores_dev, "hrs_bm25_scores: null argument");
    HRS_REQUIRE(B >= 0 && c >= 0 && V >= 0, "hrs_bm25_scores: negative size");
    HRS_REQUIRE(avg_len > 0.0 && k1 >= 0.0 && b >= 0.0 && b <= 1.0, "hrs_bm25_scores: avg_len=%g k1=%g b=%g", avg_len, k1, b);
    if ((int64_t)B * c == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    const int64_t pairs = (int64_t)B * c;
    hipLaunchKernelGGL(k_bm25_scores, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p->doc_offs,
                       p->term_ids, p->N, p->row_offset, idf_dev, V, avg_len, k1, b, q_terms_dev, q_term_offs_dev, cand_rows_dev,
                       B, c, scores_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// One block per query, one thread per candidate (c <= 256).  Python-float arithmetic: every product
// and the sum are separately rounded fp64 operations (no contraction into an fma).
// kw_scores (nullable): the keyword score of every candidate given directly (BM25) instead of counts / terms.
__global__ __launch_bounds__(256) void k_fuse_sort(const float* __restrict__ dist, const int32_t* __restrict__ counts,
                                                   const int32_t* __restrict__ q_term_offs, const double* __restrict__ kw_scores,
                                                   int c, double boost, int n, double* __restrict__ combined,
                                                   int32_t* __restrict__ perm) {
    __shared__ double comb[256];
    __shared__ int valid[256];
    const int b = blockIdx.x, j = threadIdx.x;
    double mine = 0.0;
    int ok = 0;
    if (j < c) {
        const float d32 = dist[(size_t)b * c + j];
        ok = !(d32 == __builtin_inff());
        if (ok) {
            const double d = (double)d32;
            double sem = __dsub_rn(1.0, d);
            if (!(sem > 0.0)) sem = 0.0;
            double kw;
            if (kw_scores) {
                kw = kw_scores[(size_t)b * c + j];
            } else {
                const int m = q_term_offs[b + 1] - q_term_offs[b];
                kw = m > 0 ? __ddiv_rn((double)counts[(size_t)b * c + j], (double)m) : 0.0;
            }
            const double w = __dsub_rn(1.0, boost);
            mine = __dadd_rn(__dmul_rn(sem, w), __dmul_rn(kw, boost));
        }
    }
    comb[j] = mine;
    valid[j] = ok;
    __syncthreads();
    if (j < c && ok) {
        // rank under (combined descending, candidate index ascending) == a stable descending sort
        int rank = 0;
        for (int i = 0; i < c; ++i) {
            if (!valid[i]) continue;
            const double o = comb[i];
            rank += (o > mine) || (o == mine && i < j);
        }
        if (rank < n) {
            combined[(size_t)b * n + rank] = mine;
            perm[(size_t)b * n + rank] = j;
        }
    }
    // slots past the number of valid candidates
    __syncthreads();
    int nv = 0;
    for (int i = 0; i < c; ++i) nv += valid[i];
    for (int i = nv + j; i < n; i += 256) {
        combined[(size_t)b * n + i] = -1.0;
        perm[(size_t)b * n + i] = -1;
    }
}

extern "C" int hrs_fuse_sort(hrs_ctx* ctx, const float* dist_dev, const int32_t* counts_dev,
                             const int32_t* q_term_offs_dev, int B, int c, double boost, int n, double* combined_dev,
                             int32_t* perm_dev, void* stream) {
    HRS_REQUIRE(ctx && dist_dev && counts_dev && q_term_offs_dev && combined_dev && perm_dev,
                "hrs_fuse_sort: null argument");
    HRS_REQUIRE(c >= 1 && c <= 256, "hrs_fuse_sort: c=%d outside [1, 256]", c);
    HRS_REQUIRE(n >= 1, "hrs_fuse_sort: n=%d < 1", n);
    if (B <= 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fuse_sort, dim3(B), dim3(256), 0, (hipStream_t)stream, dist_dev, counts_dev, q_term_offs_dev,
                       (const double*)nullptr, c, boost, n, combined_dev, perm_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

extern "C" int hrs_fuse_sort_scores(hrs_ctx* ctx, const float* dist_dev, const double* kw_scores_dev, int B, int c, double boost,
                                    int n, double* combined_dev, int32_t* perm_dev, void* stream) {
    HRS_REQUIRE(ctx && dist_dev && kw_scores_dev && combined_dev && perm_dev, "hrs_fuse_sort_scores: null argument");
    HRS_REQUIRE(c >= 1 && c <= 256, "hrs_fuse_sort_scores: c=%d outside [1, 256]", c);
    HRS_REQUIRE(n >= 1, "hrs_fuse_sort_scores: n=%d < 1", n);
    if (B <= 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fuse_sort, dim3(B), dim3(256), 0, (hipStream_t)stream, dist_dev, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, kw_scores_dev, c, boost, n, combined_dev, perm_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// Full-corpus hybrid mode, second half.  One block per query, one thread per candidate: the nearest k rows of every
// class (class index = keyword match count) are put nearest first (keys are unique: distance, then row), scored as
// k_fuse_sort scores them -- the same separately rounded fp64 operations -- and ranked by (fused score descending,
// position in the nearest-first list ascending), which is a stable descending sort of that list.
__global__ __launch_bounds__(256) void k_classes_finish(const uint64_t* __restrict__ keys, const int32_t* __restrict__ q_term_offs,
                                                        int levels, int k, double boost, int n, int64_t* __restrict__ rows,
                                                        float* __restrict__ dist, int32_t* __restrict__ counts,
                                                        double* __restrict__ combined) {
    __shared__ uint64_t sk[256];
    __shared__ double comb[256];
    __shared__ int nvalid;
    const int b = blockIdx.x, j = threadIdx.x, c = levels * k;
    if (j == 0) nvalid = 0;
    const uint64_t key = j < c ? keys[(size_t)b * c + j] : HRS_PAD_KEY;
    const bool ok = key != HRS_PAD_KEY;
    double mine = 0.0;
    if (ok) {
        const double d = (double)orderable_f32((uint32_t)(key >> 32));
        double sem = __dsub_rn(1.0, d);
        if (!(sem > 0.0)) sem = 0.0;
        const int m = q_term_offs[b + 1] - q_term_offs[b];
        const double kw = m > 0 ? __ddiv_rn((double)(j / k), (double)m) : 0.0;
        const double w = __dsub_rn(1.0, boost);
        mine = __dadd_rn(__dmul_rn(sem, w), __dmul_rn(kw, boost));
    }
    sk[j] = key;
    comb[j] = mine;
    __syncthreads();
    if (ok) {
        int rank = 0;
        for (int i = 0; i < c; ++i) {
            const uint64_t o = sk[i];
            if (o == HRS_PAD_KEY) continue;
            const double oc = comb[i];
            rank += (oc > mine) || (oc == mine && o < key);
        }
        if (rank < n) {
            const size_t at = (size_t)b * n + rank;
            rows[at] = (int64_t)(key & 0xFFFFFFFFull);
            dist[at] = orderable_f32((uint32_t)(key >> 32));
            counts[at] = j / k;
            combined[at] = mine;
        }
        atomicAdd(&nvalid, 1);
    }
    __syncthreads();
    for (int i = nvalid + j; i < n; i += 256) {
        const size_t at = (size_t)b * n + i;
        rows[at] = -1;
        dist[at] = __builtin_inff();
        counts[at] = 0;
        combined[at] = -1.0;
    }
}

extern "C" int hrs_classes_finish(hrs_ctx* ctx, const uint64_t* keys_dev, const int32_t* q_term_offs_dev, int B, int levels,
                                  int k, double boost, int n, int64_t* rows_dev, float* dist_dev, int32_t* counts_dev,
                                  double* combined_dev, void* stream) {
    HRS_REQUIRE(ctx && keys_dev && q_term_offs_dev && rows_dev && dist_dev && counts_dev && combined_dev,
                "hrs_classes_finish: null argument");
    HRS_REQUIRE(levels >= 1 && k >= 1 && (int64_t)levels * k <= 256, "hrs_classes_finish: levels * k = %lld outside [1, 256]",
                (long long)levels * k);
    HRS_REQUIRE(n >= 1, "hrs_classes_finish: n=%d < 1", n);
    if (B <= 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_classes_finish, dim3(B), dim3(256), 0, (hipStream_t)stream, keys_dev, q_term_offs_dev, levels, k,
                       boost, n, rows_dev, dist_dev, counts_dev, combined_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// Cross-shard merge: keys are globally unique (the row is part of the key), so the payload of each
// output key is found again by position after sorting (key, source slot) pairs.  VT: the payload's type (int32 keyword
// counts, or 64 bits that are moved and never looked at: BM25 keyword scores).
template <typename VT>
__global__ __launch_bounds__(256) void k_merge_topk(const uint64_t* __restrict__ keys, const VT* __restrict__ vals,
                                                    int G, int B, int k, uint64_t* __restrict__ out_keys,
                                                    VT* __restrict__ out_vals) {
    __shared__ uint64_t s[2048];
    __shared__ VT v[2048];
    const int b = blockIdx.x;
    const int n = G * k;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 256) {
        if (i < n) {
            const int g = i / k, j = i % k;
            const size_t src = ((size_t)g * B + b) * k + j;
            s[i] = keys[src];
            v[i] = vals ? vals[src] : 0;
        } else {
            s[i] = HRS_PAD_KEY;
            v[i] = 0;
        }
    }
    __syncthreads();
    // rank-based selection of the k smallest: n <= 2048, each thread ranks n/256 entries
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint64_t mine = s[i];
        if (mine == HRS_PAD_KEY) continue;
        int rank = 0;
        for (int t = 0; t < n; ++t) {
            const uint64_t o = s[t];
            rank += (o < mine) || (o == mine && t < i);
        }
        if (rank < k) {
            out_keys[(size_t)b * k + rank] = mine;
            if (out_vals) out_vals[(size_t)b * k + rank] = v[i];
        }
    }
    __syncthreads();
    int nv = 0;
    for (int t = threadIdx.x; t < n; t += 256) nv += s[t] != HRS_PAD_KEY;
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    atomicAdd(&total, nv);
    __syncthreads();
    for (int i = total + threadIdx.x; i < k; i += 256) {
        out_keys[(size_t)b * k + i] = HRS_PAD_KEY;
        if (out_vals) out_vals[(size_t)b * k + i] = 0;
    }
}

extern "C" int hrs_merge_topk(hrs_ctx* ctx, const uint64_t* keys_dev, const int32_t* vals_dev, int G, int B, int k,
                              uint64_t* out_keys_dev, int32_t* out_vals_dev, void* stream) {
    HRS_REQUIRE(ctx && keys_dev && out_keys_dev, "hrs_merge_topk: null argument");
    HRS_REQUIRE(G >= 1 && k >= 1 && (int64_t)G * k <= 2048, "hrs_merge_topk: G*k=%lld outside [1, 2048]",
                (long long)G * k);
    HRS_REQUIRE(!(out_vals_dev && !vals_dev), "hrs_merge_topk: out_vals without vals");
    if (B <= 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_merge_topk<int32_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, keys_dev, vals_dev, G, B, k,
                       out_keys_dev, out_vals_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

extern "C" int hrs_merge_topk_i64(hrs_ctx* ctx, const uint64_t* keys_dev, const int64_t* vals_dev, int G, int B, int k,
                                  uint64_t* out_keys_dev, int64_t* out_vals_dev, void* stream) {
    HRS_REQUIRE(ctx && keys_dev && vals_dev && out_keys_dev && out_vals_dev, "hrs_merge_topk_i64: null argument");
    HRS_REQUIRE(G >= 1 && k >= 1 && (int64_t)G * k <= 2048, "hrs_merge_topk_i64: G*k=%lld outside [1, 2048]",
                (long long)G * k);
    if (B <= 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_merge_topk<int64_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, keys_dev, vals_dev, G, B, k,
                       out_keys_dev, out_vals_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// Row masks of the `where` filter from per-key value codes (integer compares only).
// t_i = codes ? ((codes[i] is one of values[0..m)) != negate) : (other[i] != 0);  mask[i] = t_i (combine 0),
// mask[i] & t_i (1) or mask[i] | t_i (2).
__global__ __launch_bounds__(256) void k_row_mask_update(int64_t N, const int32_t* __restrict__ codes,
                                                         const int32_t* __restrict__ values, int m, int negate,
                                                         const uint8_t* __restrict__ other, int combine,
                                                         uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    bool t;
    if (codes) {
        const int32_t c = codes[i];
        bool in = false;
        for (int j = 0; j < m; ++j) in |= (values[j] == c);
        t = in != (negate != 0);
    } else {
        t = other[i] != 0;
    }
    const bool old = combine ? mask[i] != 0 : false;
    mask[i] = (uint8_t)(combine == 0 ? t : (combine == 1 ? (old && t) : (old || t)));
}

extern "C" int hrs_row_mask_update(hrs_ctx* ctx, int64_t N, const int32_t* codes_dev, const int32_t* values_dev, int m,
                                   int negate, const uint8_t* other_dev, int combine, uint8_t* mask_dev, void* stream) {
    HRS_REQUIRE(ctx && mask_dev, "hrs_row_mask_update: null argument");
    HRS_REQUIRE(N >= 0 && m >= 0 && combine >= 0 && combine <= 2, "hrs_row_mask_update: bad size or combine %d", combine);
    HRS_REQUIRE((codes_dev != nullptr) != (other_dev != nullptr), "hrs_row_mask_update: exactly one of codes_dev / other_dev");
    HRS_REQUIRE(!codes_dev || m == 0 || values_dev, "hrs_row_mask_update: m=%d with null values", m);
    if (N == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_row_mask_update, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, codes_dev,
                       values_dev, m, negate, other_dev, combine, mask_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}

// ---- full-corpus keyword side: inverted CSR postings -------------------------------------------------
// counts[b][row] = how many of query b's terms occur in `row`, for EVERY row: the posting list of each query term
// is streamed (coalesced reads of ascending rows) and the row's byte is incremented.  A row occurs at most once in
// a term's list, but two terms of a query may meet in one row from different workgroups: the increment is a 32-bit
// atomic on the word that holds the byte (counts stay below 256, so no carry crosses a byte).
__global__ __launch_bounds__(256) void k_match_counts(const int64_t* __restrict__ term_offs, const int32_t* __restrict__ doc_ids,
                                                      int64_t V, int64_t N, const int32_t* __restrict__ q_terms,
                                                      const int32_t* __restrict__ q_term_offs, uint8_t* __restrict__ counts) {
    const int b = blockIdx.y;
    uint32_t* row0 = (uint32_t*)(counts + (int64_t)b * N);  // N is a multiple of 4 here (the caller pads the row pitch)
    // a row's count is one byte: at most 255 terms of a query are counted (the rest could carry into the next row's byte)
    const int t_begin = q_term_offs[b], t_end = min(q_term_offs[b + 1], t_begin + 255);
    for (int t = t_begin; t < t_end; ++t) {
        const int term = q_terms[t];
        if (term < 0 || term >= V) continue;
        const int64_t p0 = term_offs[term], p1 = term_offs[term + 1];
        for (int64_t p = p0 + (int64_t)blockIdx.x * 256 + threadIdx.x; p < p1; p += (int64_t)gridDim.x * 256) {
            const int32_t r = doc_ids[p];
            atomicAdd(row0 + (r >> 2), 1u << (8 * (r & 3)));
        }
    }
}

extern "C" int hrs_postings_register_inverted(hrs_ctx* ctx, const int64_t* term_offs_dev, const int32_t* doc_ids_dev, int64_t V,
                                              int64_t N, hrs_inverted** out) {
    HRS_REQUIRE(ctx && out && term_offs_dev, "hrs_postings_register_inverted: null argument");
    HRS_REQUIRE(V >= 0 && N >= 0 && N < (1ll << 31), "hrs_postings_register_inverted: bad size");
    hrs_inverted* p = new (std::nothrow) hrs_inverted();
    if (!p) {
        hrs_set_error("hrs_postings_register_inverted: out of host memory");
        return HRS_ERR_NOMEM;
    }
    p->term_offs = term_offs_dev;
    p->doc_ids = doc_ids_dev;
    p->V = V;
    p->N = N;
    *out = p;
    return HRS_OK;
}

extern "C" int hrs_postings_release_inverted(hrs_ctx* ctx, hrs_inverted* p) {
    (void)ctx;
    delete p;
    return HRS_OK;
}

extern "C" int hrs_match_counts(hrs_ctx* ctx, hrs_inverted* p, const int32_t* q_terms_dev, const int32_t* q_term_offs_dev, int B,
                                int64_t row_pitch, uint8_t* counts_dev, void* stream) {
    HRS_REQUIRE(ctx && p && q_term_offs_dev && counts_dev, "hrs_match_counts: null argument");
    HRS_REQUIRE(B >= 0 && B <= 65535, "hrs_match_counts: B=%d outside [0, 65535]", B);
    HRS_REQUIRE(row_pitch >= p->N && row_pitch % 4 == 0 && ((uintptr_t)counts_dev & 3) == 0,
                "hrs_match_counts: row_pitch=%lld must be a multiple of 4 that holds the %lld rows (and counts_dev 4-byte aligned)",
                (long long)row_pitch, (long long)p->N);
    if (B == 0 || p->N == 0) return HRS_OK;
    HRS_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    HRS_CHECK_HIP(hipMemsetAsync(counts_dev, 0, (size_t)B * row_pitch, st));
    // enough workgroups per query to stream a long posting list at full bandwidth, few enough that short ones cost little
    int gx = (int)((p->N / 256 / 8) < 1 ? 1 : (p->N / 256 / 8));
    if (gx > 1024 / (B < 4 ? 1 : 4)) gx = 1024 / (B < 4 ? 1 : 4);
    hipLaunchKernelGGL(k_match_counts, dim3(gx, B), dim3(256), 0, st, p->term_offs, p->doc_ids, p->V, row_pitch, q_terms_dev,
                       q_term_offs_dev, counts_dev);
    HRS_CHECK_HIP(hipGetLastError());
    return HRS_OK;
}
