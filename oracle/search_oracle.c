/*
 * search_oracle.c -- CPU oracle for everything below the reference's
 * `collection.query(...)` seam.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py):
 * the shipped library never links or loads this file.
 *
 * What it restates
 * ----------------
 * The reference (scripts/chromadb_helper.py:259-269 and 289-293) hands one query
 * embedding to a ChromaDB server and gets back the n nearest rows by ascending
 * distance; the collection is created without an `hnsw:space` key
 * (scripts/chromadb_helper.py:82-86), so the space is ChromaDB's default, squared L2.
 * ChromaDB (requirements.txt:3, `chromadb>=0.4.0`, unpinned) is neither vendored in
 * the reference nor installed in the build container, and the reference holds no
 * test vectors for it: PARITY UNPINNED against ChromaDB itself.  The semantics below
 * are the ones this build defines and the HIP path is held to, bit for bit:
 *
 *   distance (fp64, from fp32 inputs, index order i = 0..D-1, one rounding per step):
 *     l2     : acc = fma(q_i - c_i, q_i - c_i, acc)          (the hnswlib L2Sqr sum)
 *     ip     : 1 - dot,      dot: acc = fma(q_i, c_i, acc)
 *     cosine : 1 - dot / (sqrt(|q|^2) * sqrt(|c|^2)); 1 when either norm is 0
 *   reported distance = that fp64 value rounded once to fp32 (ChromaDB returns fp32)
 *   total order       = (reported fp32 distance ascending, row index ascending)
 *
 * The hybrid re-rank follows scripts/chromadb_helper.py:296-325 with the keyword
 * presence test (scripts/chromadb_helper.py:352-359) taken over integer term ids:
 * "keyword in document" <=> "term id in the document's term list".
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { SPACE_L2 = 0, SPACE_COSINE = 1, SPACE_IP = 2 };

static inline uint32_t f32_orderable(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static inline float orderable_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* Exact distance in fp64; every step is a single correctly rounded IEEE operation. */
double hrs_oracle_dist64(const float* q, const float* c, int D, int space) {
    double acc = 0.0;
    if (space == SPACE_L2) {
        for (int i = 0; i < D; ++i) {
            double d = (double)q[i] - (double)c[i];
            acc = fma(d, d, acc);
        }
        return acc;
    }
    for (int i = 0; i < D; ++i) acc = fma((double)q[i], (double)c[i], acc);
    if (space == SPACE_IP) return 1.0 - acc;
    double nq = 0.0, nc = 0.0;
    for (int i = 0; i < D; ++i) nq = fma((double)q[i], (double)q[i], nq);
    for (int i = 0; i < D; ++i) nc = fma((double)c[i], (double)c[i], nc);
    if (nq == 0.0 || nc == 0.0) return 1.0;
    return 1.0 - acc / (sqrt(nq) * sqrt(nc));
}

/* key = (orderable fp32 distance << 32) | row : ascending key order == the total order. */
uint64_t hrs_oracle_key(const float* q, const float* c, int D, int space, uint32_t row) {
    float d32 = (float)hrs_oracle_dist64(q, c, D, space);
    return ((uint64_t)f32_orderable(d32) << 32) | row;
}

static void insert_key(uint64_t* best, int k, uint64_t key) {
    /* best[] ascending, length k, best[k-1] is the worst kept */
    if (key >= best[k - 1]) return;
    int j = k - 1;
    while (j > 0 && best[j - 1] > key) {
        best[j] = best[j - 1];
        --j;
    }
    best[j] = key;
}

/*
 * Exact top-k of every query over rows [0, N) (row ids reported as row_offset + local).
 * mask (may be NULL): rows with mask[row] == 0 are skipped (the `where` filter,
 * scripts/chromadb_helper.py:265-266).  out_keys[B][k], padded with UINT64_MAX.
 */
void hrs_oracle_topk(const float* Q, const float* C, int64_t N, int D, int B, int k, int space,
                     int64_t row_offset, const uint8_t* mask, uint64_t* out_keys) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        uint64_t* best = out_keys + (size_t)b * k;
        for (int j = 0; j < k; ++j) best[j] = UINT64_MAX;
        const float* q = Q + (size_t)b * D;
        for (int64_t r = 0; r < N; ++r) {
            if (mask && !mask[r]) continue;
            insert_key(best, k, hrs_oracle_key(q, C + (size_t)r * D, D, space, (uint32_t)(row_offset + r)));
        }
    }
}

/*
 * Same result as hrs_oracle_topk, OQB queries per pass over the rows: every (query, row) chain is
 * the same sequence of fp64 operations as in hrs_oracle_dist64 (so the keys are bit-identical), but
 * OQB independent chains keep the FMA pipes busy and a row is read once for OQB queries.
 * tests/test_search_oracle.py holds it to hrs_oracle_topk.
 */
#define OQB 32
void hrs_oracle_topk_blocked(const float* Q, const float* C, int64_t N, int D, int B, int k, int space,
                             int64_t row_offset, const uint8_t* mask, uint64_t* out_keys) {
    const int nblk = (B + OQB - 1) / OQB;
    /* cosine: the rows' squared norms (one sequential chain per row, as in hrs_oracle_dist64) once per call */
    double* ncs = NULL;
    if (space == SPACE_COSINE) {
        ncs = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
#pragma omp parallel for schedule(static)
        for (int64_t r4 = 0; r4 < N; r4 += 4) { /* four rows side by side: four independent chains */
            double s[4] = {0, 0, 0, 0};
            const int nr = N - r4 < 4 ? (int)(N - r4) : 4;
            const float* c0 = C + (size_t)r4 * D;
            for (int i = 0; i < D; ++i)
                for (int u = 0; u < nr; ++u) {
                    const double v = (double)c0[(size_t)u * D + i];
                    s[u] = fma(v, v, s[u]);
                }
            for (int u = 0; u < nr; ++u) ncs[r4 + u] = s[u];
        }
    }
#pragma omp parallel
    {
        /* the block's queries widened to fp64 once, depth-major: qd[i][u] */
        double* qd = (double*)aligned_alloc(64, sizeof(double) * OQB * (size_t)D);
#pragma omp for schedule(dynamic, 1)
        for (int blk = 0; blk < nblk; ++blk) {
            const int b0 = blk * OQB;
            const int nb = B - b0 < OQB ? B - b0 : OQB;
            double nq[OQB];
            for (int u = 0; u < OQB; ++u) {
                const float* q = Q + (size_t)(b0 + (u < nb ? u : 0)) * D;
                nq[u] = 0.0;
                for (int i = 0; i < D; ++i) {
                    qd[(size_t)i * OQB + u] = (double)q[i];
                    nq[u] = fma((double)q[i], (double)q[i], nq[u]);
                }
                if (u < nb)
                    for (int j = 0; j < k; ++j) out_keys[(size_t)(b0 + u) * k + j] = UINT64_MAX;
            }
            for (int64_t r = 0; r < N; ++r) {
                if (mask && !mask[r]) continue;
                const float* c = C + (size_t)r * D;
                double acc[OQB];
                for (int u = 0; u < OQB; ++u) acc[u] = 0.0;
                const double nc = ncs ? ncs[r] : 0.0;
                if (space == SPACE_L2) {
                    for (int i = 0; i < D; ++i) {
                        const double ci = (double)c[i];
                        const double* qi = qd + (size_t)i * OQB;
#pragma omp simd
                        for (int u = 0; u < OQB; ++u) {
                            const double d = qi[u] - ci;
                            acc[u] = fma(d, d, acc[u]);
                        }
                    }
                } else {
                    for (int i = 0; i < D; ++i) {
                        const double ci = (double)c[i];
                        const double* qi = qd + (size_t)i * OQB;
#pragma omp simd
                        for (int u = 0; u < OQB; ++u) acc[u] = fma(qi[u], ci, acc[u]);
                    }
                }
                for (int u = 0; u < nb; ++u) {
                    double d = acc[u];
                    if (space == SPACE_IP)
                        d = 1.0 - acc[u];
                    else if (space == SPACE_COSINE)
                        d = (nq[u] == 0.0 || nc == 0.0) ? 1.0 : 1.0 - acc[u] / (sqrt(nq[u]) * sqrt(nc));
                    const uint64_t key = ((uint64_t)f32_orderable((float)d) << 32) | (uint32_t)(row_offset + r);
                    insert_key(out_keys + (size_t)(b0 + u) * k, k, key);
                }
            }
        }
        free(qd);
    }
    free(ncs);
}

/* Same result, parallel over rows instead of queries (for B smaller than the core count). */
void hrs_oracle_topk_rows(const float* Q, const float* C, int64_t N, int D, int B, int k, int space,
                          int64_t row_offset, const uint8_t* mask, uint64_t* out_keys) {
    for (int b = 0; b < B; ++b) {
        uint64_t* best = out_keys + (size_t)b * k;
        for (int j = 0; j < k; ++j) best[j] = UINT64_MAX;
        const float* q = Q + (size_t)b * D;
#pragma omp parallel
        {
            uint64_t* mine = (uint64_t*)malloc(sizeof(uint64_t) * k);
            for (int j = 0; j < k; ++j) mine[j] = UINT64_MAX;
#pragma omp for schedule(static)
            for (int64_t r = 0; r < N; ++r) {
                if (mask && !mask[r]) continue;
                insert_key(mine, k, hrs_oracle_key(q, C + (size_t)r * D, D, space, (uint32_t)(row_offset + r)));
            }
#pragma omp critical
            for (int j = 0; j < k; ++j) insert_key(best, k, mine[j]);
            free(mine);
        }
    }
}

void hrs_oracle_unpack(const uint64_t* keys, int64_t n, float* dist, int64_t* rows) {
    for (int64_t i = 0; i < n; ++i) {
        if (keys[i] == UINT64_MAX) {
            dist[i] = INFINITY;
            rows[i] = -1;
        } else {
            dist[i] = orderable_f32((uint32_t)(keys[i] >> 32));
            rows[i] = (int64_t)(keys[i] & 0xFFFFFFFFu);
        }
    }
}

/* Union of G per-shard lists [G][B][k] -> best k per query (cross-shard merge). */
void hrs_oracle_merge(const uint64_t* keys, int G, int B, int k, uint64_t* out) {
    for (int b = 0; b < B; ++b) {
        uint64_t* best = out + (size_t)b * k;
        for (int j = 0; j < k; ++j) best[j] = UINT64_MAX;
        for (int g = 0; g < G; ++g)
            for (int j = 0; j < k; ++j) insert_key(best, k, keys[((size_t)g * B + b) * k + j]);
    }
}

/*
 * Keyword presence counts over a forward CSR (doc_offs[row]..doc_offs[row+1] into term_ids).
 * counts[b][j] = number of the query's terms present in candidate row rows[b][j]; 0 for rows < 0.
 * Follows scripts/chromadb_helper.py:358 with "kw in doc_lower" read as "term id in the doc's list".
 */
void hrs_oracle_keyword_counts(const int64_t* doc_offs, const int32_t* term_ids, const int32_t* q_term_offs,
                               const int32_t* q_terms, const int64_t* rows, int B, int c, int64_t row_offset,
                               int32_t* counts) {
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < c; ++j) {
            int64_t r = rows[(size_t)b * c + j];
            int32_t n = 0;
            if (r >= 0) {
                r -= row_offset;
                for (int32_t t = q_term_offs[b]; t < q_term_offs[b + 1]; ++t)
                    for (int64_t p = doc_offs[r]; p < doc_offs[r + 1]; ++p)
                        if (term_ids[p] == q_terms[t]) {
                            ++n;
                            break;
                        }
            }
            counts[(size_t)b * c + j] = n;
        }
}

/*
 * Fusion and final order, scripts/chromadb_helper.py:302-325, in Python-float (fp64) arithmetic:
 *   sem  = max(0, 1 - d)            d = the reported fp32 distance widened to fp64
 *   kw   = count / n_terms          0.0 when the query has no keywords
 *   comb = sem * (1 - boost) + kw * boost     (two products and one sum, each rounded)
 * stable sort by comb descending, first n kept.  nvalid[b] = candidates the query really has.
 * perm[b][i] = index into the candidate list (-1 past the end), combined[b][i] = its score.
 */
void hrs_oracle_fuse_sort(const float* dist, const int32_t* counts, const int32_t* n_terms, const int32_t* nvalid,
                          int B, int c, double boost, int n, double* combined, int32_t* perm, double* sem_out,
                          double* kw_out) {
    double* comb = (double*)malloc(sizeof(double) * c);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * c);
    for (int b = 0; b < B; ++b) {
        int nv = nvalid ? nvalid[b] : c;
        for (int j = 0; j < nv; ++j) {
            double d = (double)dist[(size_t)b * c + j];
            double sem = 1.0 - d;
            if (!(sem > 0.0)) sem = 0.0;
            double kw = n_terms[b] > 0 ? (double)counts[(size_t)b * c + j] / (double)n_terms[b] : 0.0;
            double w = 1.0 - boost;
            volatile double p1 = sem * w;
            volatile double p2 = kw * boost;
            comb[j] = p1 + p2;
            idx[j] = j;
            if (sem_out) sem_out[(size_t)b * c + j] = sem;
            if (kw_out) kw_out[(size_t)b * c + j] = kw;
        }
        for (int i = 1; i < nv; ++i) { /* stable insertion sort, descending */
            double v = comb[i];
            int32_t vi = idx[i];
            int j = i;
            while (j > 0 && comb[j - 1] < v) {
                comb[j] = comb[j - 1];
                idx[j] = idx[j - 1];
                --j;
            }
            comb[j] = v;
            idx[j] = vi;
        }
        for (int i = 0; i < n; ++i) {
            combined[(size_t)b * n + i] = i < nv ? comb[i] : -1.0;
            perm[(size_t)b * n + i] = i < nv ? idx[i] : -1;
        }
    }
    free(comb);
    free(idx);
}

/*
 * Full-corpus hybrid mode -- an EXTENSION with no counterpart in the reference (which fuses only the 2n nearest
 * rows, scripts/chromadb_helper.py:289-325): PARITY UNPINNED by construction.  Brute force over every row: the
 * reported fp32 distance, the keyword fraction of scripts/chromadb_helper.py:352-359 over integer term ids (forward
 * CSR), the fused score in the arithmetic of hrs_oracle_fuse_sort, then the n best under
 * (fused score descending, distance key ascending) -- what a stable descending sort of ALL rows, listed nearest first,
 * would keep.  out_rows / out_combined / out_counts [B][n]; missing entries -1 / -1.0 / 0.
 */
typedef struct {
    double comb;
    uint64_t key;
    int32_t cnt;
} full_item;

static int full_better(const full_item* a, const full_item* b) { /* does a come before b */
    if (a->comb != b->comb) return a->comb > b->comb;
    return a->key < b->key;
}

void hrs_oracle_hybrid_full(const float* Q, const float* C, int64_t N, int D, int B, int space, const int64_t* doc_offs,
                            const int32_t* term_ids, const int32_t* q_term_offs, const int32_t* q_terms, int n, double boost,
                            int64_t* out_rows, double* out_combined, int32_t* out_counts, float* out_dist) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        full_item* best = (full_item*)malloc(sizeof(full_item) * (size_t)(n > 0 ? n : 1));
        int have = 0;
        const int m = q_term_offs[b + 1] - q_term_offs[b];
        for (int64_t r = 0; r < N; ++r) {
            full_item it;
            it.key = hrs_oracle_key(Q + (size_t)b * D, C + (size_t)r * D, D, space, (uint32_t)r);
            it.cnt = 0;
            for (int32_t t = q_term_offs[b]; t < q_term_offs[b + 1]; ++t)
                for (int64_t p = doc_offs[r]; p < doc_offs[r + 1]; ++p)
                    if (term_ids[p] == q_terms[t]) {
                        ++it.cnt;
                        break;
                    }
            const double d = (double)orderable_f32((uint32_t)(it.key >> 32));
            double sem = 1.0 - d;
            if (!(sem > 0.0)) sem = 0.0;
            const double kw = m > 0 ? (double)it.cnt / (double)m : 0.0;
            const double w = 1.0 - boost;
            volatile double p1 = sem * w;
            volatile double p2 = kw * boost;
            it.comb = p1 + p2;
            if (have == n && !full_better(&it, &best[n - 1])) continue;
            int j = have < n ? have++ : n - 1;
            while (j > 0 && full_better(&it, &best[j - 1])) {
                best[j] = best[j - 1];
                --j;
            }
            best[j] = it;
        }
        for (int i = 0; i < n; ++i) {
            const size_t o = (size_t)b * n + i;
            out_rows[o] = i < have ? (int64_t)(best[i].key & 0xFFFFFFFFu) : -1;
            out_combined[o] = i < have ? best[i].comb : -1.0;
            out_counts[o] = i < have ? best[i].cnt : 0;
            out_dist[o] = i < have ? orderable_f32((uint32_t)(best[i].key >> 32)) : INFINITY;
        }
        free(best);
    }
}
