/*
 * hrs.h -- C ABI of libhrs.so, the MI355X (gfx950) retrieval engine that sits behind the
 * `chromadb_helper.py` embed / index / query surface of mxioi/hybrid-rag-system.
 *
 * The reference has no FFI of its own: its helper calls five methods of a ChromaDB collection
 * object over HTTP.  Each entry point below names the reference call site it stands in for
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns 0 on success and a negative hrs_status on failure; the message of
 *     the last failure on the calling thread is read with hrs_last_error(); no C++ exception
 *     crosses this boundary;
 *   - all `*_dev` pointers are device pointers owned by the caller (torch-ROCm tensors: pass
 *     data_ptr()); the library never frees them and keeps them only inside the handles that are
 *     documented to (a registered corpus / postings / encoder keeps its inputs);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all
 *     work is enqueued on it and no entry point waits for the device unless it says so;
 *   - one hrs_ctx per device; a ctx is not thread-safe, different ctxs are independent;
 *   - result keys: uint64 = (orderable fp32 distance << 32) | global row, ascending = better;
 *     HRS_PAD_KEY marks a missing entry (fewer rows than k).
 */
#ifndef HRS_H
#define HRS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hrs_ctx hrs_ctx;
typedef struct hrs_corpus hrs_corpus;
typedef struct hrs_postings hrs_postings;
typedef struct hrs_inverted hrs_inverted;
typedef struct hrs_encoder hrs_encoder;

enum hrs_status {
    HRS_OK = 0,
    HRS_ERR_ARG = -1,     /* bad argument (shape, range, null) */
    HRS_ERR_HIP = -2,     /* a HIP runtime call failed */
    HRS_ERR_NOMEM = -3,   /* device allocation failed */
    HRS_ERR_UNSUPPORTED = -4
};

/* ChromaDB's `hnsw:space`; the reference creates its collection without one -> l2
 * (scripts/chromadb_helper.py:82-86). */
enum hrs_space { HRS_SPACE_L2 = 0, HRS_SPACE_COSINE = 1, HRS_SPACE_IP = 2 };

/* How hrs_score_topk scores. AUTO picks EXACT for corpora under 65536 rows and, for batches of up to
 * four queries, under the size where streaming the shadow becomes faster than streaming the fp32 rows
 * (measured: 1.5e8 - 3e8 elements over an int8 shadow, 2e8 - 6e8 over a bf16 one), and MFMA otherwise.  Both return identical results (MFMA re-scores its candidates exactly and falls back
 * to EXACT for any query it cannot prove). */
enum hrs_score_mode { HRS_MODE_AUTO = 0, HRS_MODE_EXACT = 1, HRS_MODE_MFMA = 2 };

/* The reduced-precision copy of the corpus that the MFMA scan streams.  I8: int8 with one fp32 scale
 * per 32-row group, half the bytes and twice the matrix-core rate of BF16.  The choice never changes
 * results (candidates are re-scored exactly and the top-k is proven against the shadow's rigorous
 * error bound); it only changes how many rows are recorded per query and how many queries need the
 * exact fallback.  AUTO (what hrs_corpus_register builds) is I8 for every depth the MFMA scan serves
 * (D <= 1024). */
enum hrs_shadow_kind { HRS_SHADOW_NONE = 0, HRS_SHADOW_BF16 = 1, HRS_SHADOW_I8 = 2, HRS_SHADOW_AUTO = 3 };

#define HRS_PAD_KEY 0xFFFFFFFFFFFFFFFFull
#define HRS_MAX_K 128

int hrs_version(void);
/* Copies the calling thread's last error message (NUL terminated) into buf; returns its length. */
int hrs_last_error(char* buf, size_t n);

/* ---- context ------------------------------------------------------------------------------ */
/* Replaces chromadb.HttpClient(host, port) (scripts/chromadb_helper.py:74-76): the "server" is
 * the GPU `device` of this process. */
int hrs_ctx_create(int device, hrs_ctx** out);
int hrs_ctx_destroy(hrs_ctx* ctx);
/* Two contexts on one device that serve alternate query batches on two streams: after this call the
 * matrix-core scan of a hrs_score_topk on `ctx` first waits (on its stream) for the last scan launched
 * through `peer`, so the scans run one after the other while everything behind a scan (candidate sort,
 * exact re-score) overlaps the peer's next scan.  No counterpart in the reference (a single synchronous
 * process, scripts/chromadb_helper.py:591-684).  peer may be NULL to undo. */
int hrs_ctx_set_peer(hrs_ctx* ctx, hrs_ctx* peer);

/* ---- tokenizer (host code, no GPU) -------------------------------------------------------- */
/* The tokenisation sentence-transformers performs inside SentenceTransformerEmbeddingFunction
 * (scripts/chromadb_helper.py:64-66): BERT basic pre-tokenisation + WordPiece over `vocab_utf8`
 * ('\n'-separated tokens, id = line number; must hold [CLS], [SEP], [UNK]), or, with vocab_utf8 == NULL,
 * the hash mapping word -> 1000 + FNV-1a(word) % (hash_vocab_size - 1000) used with generated weights.
 * Rows start with [CLS], end with [SEP] and hold at most max_len ids. */
typedef struct hrs_tokenizer hrs_tokenizer;
int hrs_tokenizer_create(const char* vocab_utf8, int64_t vocab_bytes, int hash_vocab_size, int lower, int max_len,
                         hrs_tokenizer** out);
int hrs_tokenizer_release(hrs_tokenizer* tok);
/* texts: UTF-8 bytes of B texts back to back, text b = [offs[b], offs[b+1]).  Host outputs: ids int32
 * [B, row_stride] (row_stride >= max_len, padded with the pad id), lens int32 [B], fallback uint8 [B]:
 * 1 marks a text holding a code point whose lower-casing or normalisation depends on context (final
 * sigma, U+0130, reorderable marks); its row is empty and the caller tokenises it with the Python
 * implementation.  `threads` host threads share the texts. */
int hrs_tokenize(hrs_tokenizer* tok, const char* texts, const int64_t* offs, int B, int row_stride, int32_t* ids,
                 int32_t* lens, uint8_t* fallback, int threads);

/* ---- corpus: the collection's embedding matrix --------------------------------------------- */
/* Stands in for the server side of collection.add(...) (scripts/chromadb_helper.py:397-401,
 * 462-466, 474-478): C_dev is the fp32 [N, D] row-major embedding matrix of this shard, row i
 * being global row row_offset + i.  The corpus handle keeps C_dev (caller must keep it alive)
 * and owns a shadow copy (HRS_SHADOW_AUTO) plus row statistics that it builds on `stream`.
 * D must be a multiple of 4; N + row_offset must stay below 2^32. */
int hrs_corpus_register(hrs_ctx* ctx, const float* C_dev, int64_t N, int D, int64_t row_offset, void* stream,
                        hrs_corpus** out);
int hrs_corpus_release(hrs_ctx* ctx, hrs_corpus* corpus);
/* A further collection.add(...) on the same collection (scripts/chromadb_helper.py:397-401, 462-466, 474-478):
 * the matrix now holds N >= the registered rows; the rows registered before are unchanged but may have moved
 * (C_dev replaces the pointer the handle keeps).  Only the new rows, and the last partly filled 32-row group,
 * are quantised into the shadow, whose allocation doubles when full -- the cost of an append does not depend
 * on the size of the collection (a doubling copies the shadow and waits for `stream`).  Work is enqueued on
 * `stream`; calls that read the corpus must be ordered behind it. */
int hrs_corpus_append(hrs_ctx* ctx, hrs_corpus* corpus, const float* C_dev, int64_t N, void* stream);
/* Rebuilds the corpus' shadow as `kind` (hrs_shadow_kind) on `stream`; waits for the device first.
 * No counterpart in the reference (ChromaDB's index parameters are fixed at collection creation,
 * scripts/chromadb_helper.py:82-86); NONE leaves only the exact scan. */
int hrs_corpus_shadow(hrs_ctx* ctx, hrs_corpus* corpus, int kind, void* stream);
/* Waits for `stream`, then reports min/max squared row norm (0/0 for an empty corpus). */
int hrs_corpus_stats(hrs_ctx* ctx, hrs_corpus* corpus, void* stream, double* min_norm2, double* max_norm2);

/* Replaces collection.query(query_embeddings, n_results) on the server
 * (scripts/chromadb_helper.py:259-269 with n, 289-293 with 2n): exact top-k of every query row of
 * Q_dev (fp32 [B, D]) over the corpus, ordered by (fp32 distance ascending, row ascending).
 * row_mask_dev (nullable, uint8 [N]): rows with 0 are skipped -- the `where` filter
 * (scripts/chromadb_helper.py:265-266).  The MFMA scan never records a masked row; its thresholds come
 * from all rows, so a mask that keeps few rows leaves fewer than k candidates and the query takes the
 * exact scan (same result).
 * Outputs (each nullable): keys_dev uint64 [B, k]; dist_dev fp32 [B, k] (+inf when missing);
 * rows_dev int64 [B, k] global rows (-1 when missing).  1 <= k <= HRS_MAX_K. */
int hrs_score_topk(hrs_ctx* ctx, hrs_corpus* corpus, const float* Q_dev, int B, int k, int space,
                   const uint8_t* row_mask_dev, int mode, uint64_t* keys_dev, float* dist_dev, int64_t* rows_dev,
                   void* stream);

/* hrs_score_topk resumed behind earlier results, for n_results beyond HRS_MAX_K (collection.query does not bound
 * n_results, scripts/chromadb_helper.py:259-269): only rows whose key is strictly greater than after_keys_dev[b]
 * (uint64 [B]: the last key query b has received; HRS_PAD_KEY = nothing left) are eligible.  The order is total,
 * so consecutive calls concatenate to the exact top-n.  Exact streaming scan; other arguments as hrs_score_topk. */
int hrs_score_topk_after(hrs_ctx* ctx, hrs_corpus* corpus, const float* Q_dev, int B, int k, int space,
                         const uint8_t* row_mask_dev, const uint64_t* after_keys_dev, uint64_t* keys_dev, float* dist_dev,
                         int64_t* rows_dev, void* stream);

/* hrs_score_topk in two halves, for callers that pipeline batches over two contexts joined by
 * hrs_ctx_set_peer (no counterpart in the reference, which serves one query per process,
 * scripts/chromadb_helper.py:591-684).  _begin launches the scan (thresholds from sampled passes, then
 * the main pass); _finish, on the same stream, launches the rest (candidate sort, exact re-score, proof,
 * fallback) and writes the outputs of hrs_score_topk.  When batch i+1 is begun on the peer context before
 * batch i is finished, the tail of batch i waits for the thresholds of batch i+1 and then runs under that
 * scan's main pass, whose matrix-core work it barely disturbs.  One call may be open per context; Q_dev and
 * the mask must stay valid until _finish; calls that cannot be split (exact mode, more than 1024 queries)
 * run whole in _finish.  Results are those of hrs_score_topk. */
int hrs_score_topk_begin(hrs_ctx* ctx, hrs_corpus* corpus, const float* Q_dev, int B, int k, int space,
                         const uint8_t* row_mask_dev, int mode, void* stream);
int hrs_score_topk_finish(hrs_ctx* ctx, uint64_t* keys_dev, float* dist_dev, int64_t* rows_dev, void* stream);

/* hrs_score_topk for callers that cannot allocate device memory -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference (one synchronous
 * process per query, scripts/chromadb_helper.py:591-684).  Same results as hrs_score_topk in HRS_MODE_EXACT
 * (the fp64 streaming scan), same arguments except that there is no mode and keys_dev is required; dist_dev
 * and rows_dev stay optional.  The per-workgroup candidate lists live in a static device array of 4.25 MiB
 * that every GPU the library runs on holds until the process ends, hence the limits: at most
 * HRS_STATIC_MAX_B queries per call (more is HRS_ERR_ARG), devices with at most 256 compute units.  The call
 * performs no hipMalloc / hipFree.  All calls on one GPU share that array: calls that overlap on one GPU
 * (two streams, two threads, two contexts) race; calls on one stream are ordered and need no clearing (every
 * word a call reads it has written itself). */
#define HRS_STATIC_MAX_B 16
int hrs_score_topk_static(hrs_ctx* ctx, hrs_corpus* corpus, const float* Q_dev, int B, int k, int space,
                          const uint8_t* row_mask_dev, uint64_t* keys_dev, float* dist_dev, int64_t* rows_dev,
                          void* stream);

/* Counters of the last MFMA-mode call on this ctx, valid after the stream has been waited for:
 * out[0] = queries that took the exact fallback, out[1] = largest candidate count of a query,
 * out[2] = queries whose candidate list overflowed. Waits for `stream`. */
int hrs_score_stats(hrs_ctx* ctx, void* stream, int64_t out[3]);
/* Measurement aid (no counterpart in the reference): the share of the corpus' tiles that the main pass of the last
 * matrix-core call on this context scored itself -- 1.0, or a little less where the second sampled pass has recorded
 * the rows of its own tiles and the main pass leaves those out.  Waits for the stream. */
int hrs_scan_work(hrs_ctx* ctx, void* stream, double* main_pass_share);

/* Kernel timing for benchmarks: while enabled, every launch of the dominant scan kernel (the MFMA
 * main pass in MFMA mode, the fp64 streaming scan in a direct EXACT-mode call) is bracketed by HIP
 * events on the caller's stream.  hrs_profile_read waits for them, returns their summed duration
 * and count, and resets the tally. */
int hrs_profile_enable(hrs_ctx* ctx, int on);
int hrs_profile_read(hrs_ctx* ctx, double* total_ms, int64_t* launches);

/* ---- encoder ------------------------------------------------------------------------------- */
/* Replaces the embedding function the reference builds with
 * SentenceTransformerEmbeddingFunction(model_name="all-MiniLM-L6-v2") (scripts/chromadb_helper.py:58-71)
 * and that ChromaDB calls inside collection.add / collection.query: BERT-style post-LN encoder,
 * erf-GELU, masked mean pooling, L2 normalisation.  Built for hidden = 384, heads = 12. */
typedef struct hrs_encoder_config {
    int vocab, hidden, layers, heads, inter, max_pos;
    float eps;
} hrs_encoder_config;
/* weights_dev: 5 + 16 * layers fp32 DEVICE tensors in this order: word_emb [vocab, H], pos_emb
 * [max_pos, H], type_emb [2, H], emb_ln_gamma, emb_ln_beta, then per layer q_w [H,H], q_b, k_w, k_b,
 * v_w, v_b, o_w [H,H], o_b, ln1_gamma, ln1_beta, i_w [I,H], i_b, d_w [H,I], d_b, ln2_gamma, ln2_beta
 * (linear weights [out, in]).  The encoder owns bf16 / fp32 copies; the inputs may be freed once
 * `stream` has been waited for. */
int hrs_encoder_load(hrs_ctx* ctx, const hrs_encoder_config* cfg, const float* const* weights_dev, void* stream,
                     hrs_encoder** out);
int hrs_encoder_release(hrs_ctx* ctx, hrs_encoder* enc);
/* ids_dev int32 [B, S] token ids (padded), lens_dev int32 [B] valid tokens per row, S a multiple of
 * 32 in [32, 256]; out_dev fp32 [B, hidden] unit-norm sentence embeddings. */
int hrs_encode(hrs_ctx* ctx, hrs_encoder* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int S,
               float* out_dev, void* stream);

/* ---- the `where` filter -------------------------------------------------------------------- */
/* Builds the row mask that hrs_score_topk takes from per-key value codes kept on the device, for the equality
 * forms of ChromaDB's `where` (scripts/chromadb_helper.py:265-266: {"system": S}; also $eq/$ne/$in/$nin/$and/$or).
 * t_i = codes_dev ? ((codes_dev[i] is one of values_dev[0..m)) != negate) : (other_dev[i] != 0)  -- exactly one of
 * codes_dev (int32 [N], the host's code of row i's value under one metadata key) and other_dev (uint8 [N],
 * another mask) is given; mask_dev[i] = t_i (combine 0), mask_dev[i] AND t_i (1), mask_dev[i] OR t_i (2). */
int hrs_row_mask_update(hrs_ctx* ctx, int64_t N, const int32_t* codes_dev, const int32_t* values_dev, int m, int negate,
                        const uint8_t* other_dev, int combine, uint8_t* mask_dev, void* stream);

/* ---- keyword side -------------------------------------------------------------------------- */
/* Forward CSR postings of this shard: the term ids of row i are
 * term_ids_dev[doc_offs_dev[i] .. doc_offs_dev[i+1]).  The handle keeps both pointers. */
int hrs_postings_register(hrs_ctx* ctx, const int64_t* doc_offs_dev, const int32_t* term_ids_dev, int64_t N,
                          int64_t row_offset, hrs_postings** out);
int hrs_postings_release(hrs_ctx* ctx, hrs_postings* p);

/* Replaces the numerator of calculate_keyword_score (scripts/chromadb_helper.py:352-359) with
 * "keyword in document" read as "term id in the row's term list": counts_dev[b][j] = how many of
 * query b's terms q_terms_dev[q_term_offs_dev[b] .. q_term_offs_dev[b+1]) occur in candidate row
 * cand_rows_dev[b][j] (global row; rows < 0 or outside this shard count 0). */
int hrs_keyword_counts(hrs_ctx* ctx, hrs_postings* p, const int32_t* q_terms_dev, const int32_t* q_term_offs_dev,
                       const int64_t* cand_rows_dev, int B, int c, int32_t* counts_dev, void* stream);

/* Replaces the scoring loop, sort and cut of hybrid_query (scripts/chromadb_helper.py:302-325),
 * in Python-float (fp64) arithmetic: sem = max(0, 1 - dist), kw = count / n_terms (0 without
 * keywords), combined = sem * (1 - boost) + kw * boost; stable sort by combined descending; first
 * n kept.  dist_dev fp32 [B, c] ascending per query, +inf = missing candidate.  Outputs:
 * combined_dev fp64 [B, n] (-1 when missing), perm_dev int32 [B, n] index into the candidate list
 * (-1 when missing).  c <= 2 * HRS_MAX_K. */
int hrs_fuse_sort(hrs_ctx* ctx, const float* dist_dev, const int32_t* counts_dev, const int32_t* q_term_offs_dev,
                  int B, int c, double boost, int n, double* combined_dev, int32_t* perm_dev, void* stream);

/* ---- BM25 keyword weighting (an extension: the reference's keyword score is the fraction above) ------ */
/* Where calculate_keyword_score (scripts/chromadb_helper.py:352-359) counts the keywords present, this weighs them:
 * scores_dev fp64 [B, c] = sum over query b's terms t of idf_dev[t] * tf * (k1 + 1) / (tf + k1 * (1 - b + b * len / avg_len)),
 * divided by sum idf_dev[t] * (k1 + 1) so that it lies in [0, 1) like the fraction it replaces.  tf = occurrences of t in
 * the candidate row's term list, len = the list's length; idf_dev fp64 [V] comes from the caller (terms outside
 * [0, V) weigh nothing).  Every operation is a separately rounded fp64 one, in query-term order. */
int hrs_bm25_scores(hrs_ctx* ctx, hrs_postings* p, const double* idf_dev, int64_t V, double avg_len, double k1, double b,
                    const int32_t* q_terms_dev, const int32_t* q_term_offs_dev, const int64_t* cand_rows_dev, int B, int c,
                    double* scores_dev, void* stream);
/* hrs_fuse_sort with the keyword score of every candidate given (kw_scores_dev fp64 [B, c]) instead of counts / terms. */
int hrs_fuse_sort_scores(hrs_ctx* ctx, const float* dist_dev, const double* kw_scores_dev, int B, int c, double boost, int n,
                         double* combined_dev, int32_t* perm_dev, void* stream);

/* ---- full-corpus hybrid mode (an extension: NOT what the reference computes) ------------------------ */
/* The reference re-ranks the 2n nearest rows (scripts/chromadb_helper.py:289-325); these entry points fuse the keyword
 * fraction of scripts/chromadb_helper.py:352-359 with the semantic score over EVERY row instead, exactly: within the
 * rows that match the same number c of a query's terms the fused score falls as the distance grows, so the best n of
 * each class c = 0 .. m (by distance, then row) contain the best n overall.
 *
 * Inverted CSR of this shard: the rows that hold term t are doc_ids_dev[term_offs_dev[t] .. term_offs_dev[t+1])
 * (local rows, each at most once per term); the handle keeps both pointers. */
int hrs_postings_register_inverted(hrs_ctx* ctx, const int64_t* term_offs_dev, const int32_t* doc_ids_dev, int64_t V,
                                   int64_t N, hrs_inverted** out);
int hrs_postings_release_inverted(hrs_ctx* ctx, hrs_inverted* p);
/* counts_dev uint8 [B, row_pitch] (row_pitch >= N, a multiple of 4): counts_dev[b][r] = how many of query b's terms
 * (distinct; only the first 255 terms of a query are counted, a count is one byte) occur in local row r -- the
 * numerator of calculate_keyword_score for every row. */
int hrs_match_counts(hrs_ctx* ctx, hrs_inverted* p, const int32_t* q_terms_dev, const int32_t* q_term_offs_dev, int B,
                     int64_t row_pitch, uint8_t* counts_dev, void* stream);
/* keys_dev uint64 [B, levels, k]: for every query and every l < levels the exact top-k (as hrs_score_topk, exact
 * streaming scan) among the rows with level_dev[b * row_pitch + r] == l.  One pass over the corpus per four queries
 * serves 256 / k classes at once (row_pitch a multiple of 4, level_dev 4-byte aligned; otherwise one pass per class). */
int hrs_score_topk_levels(hrs_ctx* ctx, hrs_corpus* corpus, const float* Q_dev, int B, int k, int space,
                          const uint8_t* level_dev, int64_t row_pitch, int levels, uint64_t* keys_dev, void* stream);

/* Second half of the full-corpus mode: keys_dev uint64 [B, levels, k] (of one shard, or merged per class over all
 * shards; the class index is the keyword match count) -> the n best rows under the fused score of
 * scripts/chromadb_helper.py:302-325 (arithmetic and order as hrs_fuse_sort; ties by distance, then row):
 * rows_dev int64 [B, n] (-1 when missing), dist_dev fp32 [B, n] (+inf), counts_dev int32 [B, n] (0), combined_dev fp64
 * [B, n] (-1).  levels * k <= 256. */
int hrs_classes_finish(hrs_ctx* ctx, const uint64_t* keys_dev, const int32_t* q_term_offs_dev, int B, int levels, int k,
                       double boost, int n, int64_t* rows_dev, float* dist_dev, int32_t* counts_dev, double* combined_dev,
                       void* stream);

/* ---- cross-shard merge (no counterpart in the single-process reference) -------------------- */
/* keys_dev uint64 [G, B, k]: per-shard results gathered from G ranks (all-gather over RCCL is done
 * by the caller).  vals_dev (nullable) int32 [G, B, k] rides along (per-candidate keyword counts).
 * Writes the best k of each query's G*k keys (and their vals).  G * k <= 2048. */
int hrs_merge_topk(hrs_ctx* ctx, const uint64_t* keys_dev, const int32_t* vals_dev, int G, int B, int k,
                   uint64_t* out_keys_dev, int32_t* out_vals_dev, void* stream);
/* The same merge with a 64-bit payload per candidate, moved and never interpreted (the bits of the fp64 keyword scores
 * of hrs_bm25_scores on a row-sharded corpus; slots without a key get 0). */
int hrs_merge_topk_i64(hrs_ctx* ctx, const uint64_t* keys_dev, const int64_t* vals_dev, int G, int B, int k,
                       uint64_t* out_keys_dev, int64_t* out_vals_dev, void* stream);
/* keys -> (fp32 distance, int64 row); either output nullable. */
int hrs_unpack_keys(hrs_ctx* ctx, const uint64_t* keys_dev, int64_t n, float* dist_dev, int64_t* rows_dev,
                    void* stream);

/* ---- synthetic inputs for benchmarks and tests --------------------------------------------- */
/* Rows row0 .. row0+nrows-1 of the i.i.d. N(0,1) matrix keyed (seed, row, column), L2-normalised
 * when normalize != 0; any shard of any size reproduces the same rows. */
int hrs_synth_rows(hrs_ctx* ctx, uint64_t seed, int64_t row0, int64_t nrows, int D, int normalize, float* out_dev,
                   void* stream);
/* L distinct ascending term ids per row, drawn by inverse CDF from cdf_dev (fp32 [V], ascending,
 * last = 1) keyed (seed, row, draw); out_dev int32 [nrows, L]. */
int hrs_synth_postings(hrs_ctx* ctx, uint64_t seed, int64_t row0, int64_t nrows, int V, int L, const float* cdf_dev,
                       int32_t* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HRS_H */
