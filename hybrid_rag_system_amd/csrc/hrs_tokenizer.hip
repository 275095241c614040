// Host-side tokenisation in front of the encoder: BERT "basic" pre-tokenisation (clean, CJK spacing,
// lower-case, accent strip, punctuation split) + greedy WordPiece over a vocabulary, or the hash
// word -> id mapping used with generated weights.  It is the native form of
// hybrid_rag_system_amd/tokenizer.py and produces the same ids; sentence-transformers does this step
// inside SentenceTransformerEmbeddingFunction (scripts/chromadb_helper.py:64-66).  No GPU code here:
// texts are tokenised by a pool of host threads so that the encoder is not left waiting for Python.
//
// Code-point properties come from tables generated out of the Python interpreter's unicodedata
// (tools/gen_unicode_tables.py).  A text that contains one of the few code points whose handling
// depends on context (final sigma, U+0130, marks that canonical reordering could move) is flagged and
// left to the Python implementation.
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <exception>

#include "hrs_internal.h"
#include "hrs_unicode_tables.h"

namespace {

bool in_ranges(const uint32_t (*r)[2], int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < r[mid][0])
            hi = mid - 1;
        else if (cp > r[mid][1])
            lo = mid + 1;
        else
            return true;
    }
    return false;
}

uint32_t to_lower(uint32_t cp) {
    if (cp < 128) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
    int lo = 0, hi = uc_lower_n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < uc_lower[mid][0])
            hi = mid - 1;
        else if (cp > uc_lower[mid][0])
            lo = mid + 1;
        else
            return uc_lower[mid][1];
    }
    return cp;
}

// canonical decomposition of one code point appended to out (Hangul syllables algorithmically)
void decompose(uint32_t cp, std::vector<uint32_t>& out) {
    if (cp < 0xC0) {
        out.push_back(cp);
        return;
    }
    if (cp >= 0xAC00 && cp <= 0xD7A3) {
        const uint32_t s = cp - 0xAC00;
        out.push_back(0x1100 + s / 588);
        out.push_back(0x1161 + (s % 588) / 28);
        if (s % 28) out.push_back(0x11A7 + s % 28);
        return;
    }
    int lo = 0, hi = uc_decomp_n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < uc_decomp_index[mid][0])
            hi = mid - 1;
        else if (cp > uc_decomp_index[mid][0])
            lo = mid + 1;
        else {
            for (uint32_t j = 0; j < uc_decomp_index[mid][2]; ++j) out.push_back(uc_decomp_data[uc_decomp_index[mid][1] + j]);
            return;
        }
    }
    out.push_back(cp);
}

bool is_cjk(uint32_t cp) {
    return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
           (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
           (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}

bool is_punct(uint32_t cp) {
    if ((cp >= 33 && cp <= 47) || (cp >= 58 && cp <= 64) || (cp >= 91 && cp <= 96) || (cp >= 123 && cp <= 126)) return true;
    if (cp < 128) return false;
    return in_ranges(uc_punct, uc_punct_n, cp);
}

void put_utf8(uint32_t cp, std::string& s) {
    if (cp < 0x80) {
        s.push_back((char)cp);
    } else if (cp < 0x800) {
        s.push_back((char)(0xC0 | (cp >> 6)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18)));
        s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

}  // namespace

struct hrs_tokenizer {
    bool hash_mode = true;
    int hash_vocab = 30522;
    bool lower = true;
    int max_len = 256;
    int cls_id = 101, sep_id = 102, pad_id = 0, unk_id = 100;
    std::unordered_map<std::string, int> vocab;

    // ids of one (already split) word, appended to out
    void word_ids(const std::vector<uint32_t>& w, std::vector<int32_t>& out, std::string& scratch) const {
        if (hash_mode) {
            uint64_t h = 0xCBF29CE484222325ull;
            scratch.clear();
            for (uint32_t cp : w) put_utf8(cp, scratch);
            for (unsigned char b : scratch) h = (h ^ b) * 0x100000001B3ull;
            out.push_back((int32_t)(1000 + h % (uint64_t)(hash_vocab - 1000)));
            return;
        }
        if (w.size() > 100) {
            out.push_back(unk_id);
            return;
        }
        const size_t first = out.size();
        size_t start = 0;
        while (start < w.size()) {
            size_t end = w.size();
            int cur = -1;
            while (start < end) {
                scratch.clear();
                if (start > 0) scratch.append("##");
                for (size_t j = start; j < end; ++j) put_utf8(w[j], scratch);
                auto it = vocab.find(scratch);
                if (it != vocab.end()) {
                    cur = it->second;
                    break;
                }
                --end;
            }
            if (cur < 0) {
                out.resize(first);
                out.push_back(unk_id);
                return;
            }
            out.push_back(cur);
            start = end;
        }
    }

    // -> false when the text must be tokenised by the Python implementation
    bool encode(const unsigned char* s, int64_t n, std::vector<int32_t>& ids) const {
        ids.clear();
        ids.push_back(cls_id);
        std::vector<uint32_t> tok, low, word;
        std::string scratch;
        bool full = false;
        auto flush_word = [&]() {
            if (word.empty() || full) return;
            word_ids(word, ids, scratch);
            word.clear();
            if ((int)ids.size() >= max_len - 1) full = true;
        };
        auto flush_token = [&]() {
            if (tok.empty()) return;
            const std::vector<uint32_t>* t = &tok;
            if (lower) {
                low.clear();
                for (uint32_t cp : tok) decompose(to_lower(cp), low);
                // strip nonspacing marks
                size_t k = 0;
                for (uint32_t cp : low)
                    if (cp < 0x300 || !in_ranges(uc_mn, uc_mn_n, cp)) low[k++] = cp;
                low.resize(k);
                t = &low;
            }
            for (uint32_t cp : *t) {
                if (is_punct(cp)) {
                    flush_word();
                    word.push_back(cp);
                    flush_word();
                } else {
                    word.push_back(cp);
                }
            }
            flush_word();
            tok.clear();
        };
        int64_t p = 0;
        while (p < n && !full) {
            // decode one code point (the caller passes well-formed UTF-8, surrogates included as 3 bytes)
            uint32_t cp = s[p];
            int len = 1;
            if (cp >= 0xF0) {
                len = 4;
                cp &= 0x07;
            } else if (cp >= 0xE0) {
                len = 3;
                cp &= 0x0F;
            } else if (cp >= 0xC0) {
                len = 2;
                cp &= 0x1F;
            }
            if (p + len > n) return false;
            for (int j = 1; j < len; ++j) cp = (cp << 6) | (s[p + j] & 0x3F);
            p += len;
            if (cp == 0 || cp == 0xFFFD) continue;
            if (cp < 128) {
                if (cp == '\t' || cp == '\n' || cp == '\r' || cp == ' ') {
                    flush_token();
                } else if (cp < 32 || cp == 127) {
                    continue;
                } else {
                    tok.push_back(cp);
                }
                continue;
            }
            if (in_ranges(uc_special, uc_special_n, cp)) return false;
            if (in_ranges(uc_space, uc_space_n, cp)) {
                flush_token();
            } else if (in_ranges(uc_ctrl, uc_ctrl_n, cp)) {
                continue;
            } else if (cp == 0x2028 || cp == 0x2029) {  // kept by the cleaning step, whitespace to str.split()
                flush_token();
            } else if (is_cjk(cp)) {
                flush_token();
                tok.push_back(cp);
                flush_token();
            } else {
                tok.push_back(cp);
            }
        }
        if (!full) flush_token();
        if ((int)ids.size() > max_len - 1) ids.resize(max_len - 1);
        ids.push_back(sep_id);
        return true;
    }
};

extern "C" int hrs_tokenizer_create(const char* vocab_utf8, int64_t vocab_bytes, int hash_vocab_size, int lower, int max_len,
                                    hrs_tokenizer** out) {
    HRS_REQUIRE(out, "hrs_tokenizer_create: null out");
    HRS_REQUIRE(max_len >= 2 && max_len <= 512, "hrs_tokenizer_create: max_len=%d outside [2, 512]", max_len);
    hrs_tokenizer* t = nullptr;
    try {  // no C++ exception crosses the C boundary
    t = new hrs_tokenizer();
    t->lower = lower != 0;
    t->max_len = max_len;
    if (vocab_utf8 && vocab_bytes > 0) {
        t->hash_mode = false;
        int id = 0;
        int64_t a = 0;
        for (int64_t p = 0; p <= vocab_bytes; ++p) {
            if (p == vocab_bytes || vocab_utf8[p] == '\n') {
                if (p > a || p < vocab_bytes) {
                    // later duplicates win, as in a Python dict comprehension
                    t->vocab[std::string(vocab_utf8 + a, (size_t)(p - a))] = id;
                    ++id;
                }
                a = p + 1;
            }
        }
        auto need = [&](const char* name, int* dst, bool required) -> bool {
            auto it = t->vocab.find(name);
            if (it == t->vocab.end()) return !required;
            *dst = it->second;
            return true;
        };
        t->pad_id = 0;
        if (!need("[CLS]", &t->cls_id, true) || !need("[SEP]", &t->sep_id, true) || !need("[UNK]", &t->unk_id, true)) {
            delete t;
            hrs_set_error("hrs_tokenizer_create: the vocabulary lacks [CLS], [SEP] or [UNK]");
            return HRS_ERR_ARG;
        }
        need("[PAD]", &t->pad_id, false);
    } else {
        if (hash_vocab_size <= 1000) {
            delete t;
            hrs_set_error("hrs_tokenizer_create: hash vocabulary of %d ids", hash_vocab_size);
            return HRS_ERR_ARG;
        }
        t->hash_vocab = hash_vocab_size;
    }
    } catch (const std::exception& ex) {
        delete t;
        hrs_set_error("hrs_tokenizer_create: %s", ex.what());
        return HRS_ERR_NOMEM;
    }
    *out = t;
    return HRS_OK;
}

extern "C" int hrs_tokenizer_release(hrs_tokenizer* t) {
    delete t;
    return HRS_OK;
}

extern "C" int hrs_tokenize(hrs_tokenizer* t, const char* texts, const int64_t* offs, int B, int row_stride, int32_t* ids,
                            int32_t* lens, uint8_t* fallback, int threads) {
    HRS_REQUIRE(t && offs && ids && lens && fallback, "hrs_tokenize: null argument");
    HRS_REQUIRE(B >= 0 && row_stride >= t->max_len, "hrs_tokenize: row stride %d below max_len %d", row_stride, t->max_len);
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (threads > B) threads = B > 0 ? B : 1;
    std::atomic<int> next(0);
    std::atomic<bool> failed(false);
    auto work = [&]() {
        try {  // an exception (out of host memory) must neither leave a thread nor cross the C boundary
        std::vector<int32_t> row;
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= B) break;
            const bool ok = t->encode((const unsigned char*)texts + offs[b], offs[b + 1] - offs[b], row);
            fallback[b] = ok ? 0 : 1;
            int32_t* dst = ids + (size_t)b * row_stride;
            const int n = ok ? (int)row.size() : 0;
            for (int j = 0; j < n; ++j) dst[j] = row[j];
            for (int j = n; j < row_stride; ++j) dst[j] = t->pad_id;
            lens[b] = n;
        }
        } catch (...) {
            failed = true;
        }
    };
    if (threads == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        try {
            for (int i = 0; i < threads; ++i) pool.emplace_back(work);
        } catch (...) {  // fewer threads than asked for: the ones that started share the texts with this one
            work();
        }
        for (auto& th : pool) th.join();
    }
    if (failed) {
        hrs_set_error("hrs_tokenize: out of host memory");
        return HRS_ERR_NOMEM;
    }
    return HRS_OK;
}
