"""Tokenisation in front of the encoder (host side).

sentence-transformers tokenises with the model's WordPiece vocabulary and truncates to 256 tokens
before the BERT forward (the embedding function built at scripts/chromadb_helper.py:58-71).  Two
tokenisers share the BERT "basic" pre-tokenisation (clean, CJK spacing, lower-case, accent strip,
punctuation split):

* WordPieceTokenizer -- greedy longest-match WordPiece over a LOCAL vocab.txt (real-model use);
* HashTokenizer      -- no vocabulary exists offline, so words map to ids by a fixed hash; it keeps
                        the pipeline deterministic and testable with generated weights.

`encode_batch` runs the native implementation of the same rules (hrs_tokenize in libhrs.so, a pool of
host threads: the Python loop below manages ~700 chunks/s per core, the encoder 100 k chunks/s); the
Python code stays the definition, answers the few texts the native path declines (context-dependent
lower-casing such as final sigma) and is what the parity tests compare the native ids with.
"""
import ctypes as C
import os
import unicodedata

import numpy as np

CLS_ID, SEP_ID, PAD_ID, UNK_ID = 101, 102, 0, 100  # the ids bert-base-uncased style vocabularies use


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (
        0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
        or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F
    )


def basic_tokenize(text, lower=True):
    out = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD:
            continue
        cat = unicodedata.category(ch)
        if ch in "\t\n\r" or cat == "Zs":
            out.append(" ")
        elif cat.startswith("C"):
            continue
        elif _is_cjk(cp):
            out.append(f" {ch} ")
        else:
            out.append(ch)
    words = []
    for tok in "".join(out).split():
        if lower:
            tok = tok.lower()
            tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
        cur = []
        for ch in tok:
            if _is_punct(ch):
                if cur:
                    words.append("".join(cur))
                    cur = []
                words.append(ch)
            else:
                cur.append(ch)
        if cur:
            words.append("".join(cur))
    return words


class _Base:
    max_len = 256

    def encode(self, text):
        ids = [self.cls_id]
        for w in basic_tokenize(text, self.lower):
            ids.extend(self.word_ids(w))
            if len(ids) >= self.max_len - 1:
                break
        return ids[: self.max_len - 1] + [self.sep_id]

    def encode_batch_python(self, texts):
        """-> (ids int32 [B, S] padded, lens int32 [B]); S = longest row rounded up to a multiple of 32."""
        rows = [self.encode(t) for t in texts]
        longest = max((len(r) for r in rows), default=2)
        S = min(256, (longest + 31) // 32 * 32)
        ids = np.full((len(rows), S), self.pad_id, np.int32)
        lens = np.zeros(len(rows), np.int32)
        for b, r in enumerate(rows):
            ids[b, : len(r)] = r
            lens[b] = len(r)
        return ids, lens

    # ---- native path ------------------------------------------------------------------------------
    _native = None

    def _vocab_blob(self):
        return None  # hash mode

    def _native_handle(self):
        if self._native is None:
            from . import _native as N

            blob = self._vocab_blob()
            h = C.c_void_p()
            N.check(
                N.lib().hrs_tokenizer_create(blob, len(blob) if blob else 0, getattr(self, "vocab_size", 0), int(self.lower), self.max_len, C.byref(h)),
                "hrs_tokenizer_create",
            )
            self._native = h
        return self._native

    def encode_batch(self, texts, threads=None):
        """Same result as encode_batch_python, computed by hrs_tokenize on `threads` host threads."""
        from . import _native as N

        texts = list(texts)
        B = len(texts)
        if B == 0 or self.max_len > 256:
            return self.encode_batch_python(texts)
        raw = [t.encode("utf-8", "surrogatepass") for t in texts]
        offs = np.zeros(B + 1, np.int64)
        np.cumsum([len(r) for r in raw], out=offs[1:])
        blob = b"".join(raw)
        stride = 256
        ids = np.empty((B, stride), np.int32)
        lens = np.empty(B, np.int32)
        fb = np.empty(B, np.uint8)
        if threads is None:
            threads = min(16, os.cpu_count() or 1)
        N.check(
            N.lib().hrs_tokenize(self._native_handle(), blob, offs.ctypes.data, B, stride, ids.ctypes.data, lens.ctypes.data, fb.ctypes.data, threads),
            "hrs_tokenize",
        )
        for b in np.nonzero(fb)[0]:  # context-dependent code points: the Python rules decide
            r = self.encode(texts[b])
            ids[b, : len(r)] = r
            ids[b, len(r) :] = self.pad_id
            lens[b] = len(r)
        longest = int(lens.max()) if B else 2
        S = min(256, (max(longest, 2) + 31) // 32 * 32)
        return np.ascontiguousarray(ids[:, :S]), lens

    def __del__(self):
        if getattr(self, "_native", None) is not None:
            try:
                from . import _native as N

                N.lib().hrs_tokenizer_release(self._native)
            except Exception:
                pass


class WordPieceTokenizer(_Base):
    def __init__(self, vocab, lower=True, max_len=256):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = [line.rstrip("\n") for line in f]
        self.vocab = {tok: i for i, tok in enumerate(vocab)}
        self.lower, self.max_len = lower, max_len
        self.cls_id, self.sep_id = self.vocab["[CLS]"], self.vocab["[SEP]"]
        self.pad_id, self.unk_id = self.vocab.get("[PAD]", 0), self.vocab["[UNK]"]
        self._vocab_list = list(vocab)

    def _vocab_blob(self):
        return "\n".join(self._vocab_list).encode("utf-8", "surrogatepass")

    def word_ids(self, word):
        if len(word) > 100:
            return [self.unk_id]
        out, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                piece = word[start:end] if start == 0 else "##" + word[start:end]
                if piece in self.vocab:
                    cur = self.vocab[piece]
                    break
                end -= 1
            if cur is None:
                return [self.unk_id]
            out.append(cur)
            start = end
        return out


class HashTokenizer(_Base):
    """word -> 1000 + FNV-1a(word) mod (vocab - 1000): one id per word, no sub-word split."""

    def __init__(self, vocab_size=30522, lower=True, max_len=256):
        self.vocab_size, self.lower, self.max_len = vocab_size, lower, max_len
        self.cls_id, self.sep_id, self.pad_id, self.unk_id = CLS_ID, SEP_ID, PAD_ID, UNK_ID

    def word_ids(self, word):
        h = 0xCBF29CE484222325
        for b in word.encode("utf-8"):
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return [1000 + h % (self.vocab_size - 1000)]
