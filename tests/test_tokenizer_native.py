"""The native tokenizer (hrs_tokenize, host threads) against the Python rules it restates, and both against
transformers' BertTokenizer where a vocabulary is involved."""
import glob
import os
import random

import numpy as np
import pytest

from hybrid_rag_system_amd.tokenizer import HashTokenizer, WordPieceTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))


def _same(tok, texts, threads=4):
    a_ids, a_len = tok.encode_batch(texts, threads=threads)
    b_ids, b_len = tok.encode_batch_python(texts)
    assert np.array_equal(a_len, b_len), np.nonzero(a_len != b_len)[0][:10]
    assert a_ids.shape == b_ids.shape
    bad = np.nonzero((a_ids != b_ids).any(1))[0]
    assert bad.size == 0, (bad[:5], [texts[i] for i in bad[:3]])


def _fixture_chunks():
    out = []
    for f in sorted(glob.glob(os.path.join(HERE, "golden", "context", "infrastructure", "*.md"))):
        t = open(f, encoding="utf-8").read()
        out += [t[i : i + 1000] for i in range(0, len(t), 800)]
    return out


def test_native_equals_python_on_documents():
    chunks = _fixture_chunks() + ["", " ", "\n\t", "a", "x" * 5000, "Hello, World! (test) [CLS] ##ing"]
    _same(HashTokenizer(), chunks)
    _same(HashTokenizer(lower=False), chunks)


def test_native_equals_python_every_code_point():
    """Every BMP code point and every 7th beyond, each in a little context: classification, lower-casing,
    decomposition and mark stripping agree (or the native path declines and Python answers)."""
    cps = list(range(1, 0x10000)) + list(range(0x10000, 0x110000, 7))
    texts = []
    for cp in cps:
        ch = chr(cp)
        texts.append(f"ab{ch}cd {ch}{ch} X{ch}. {ch}")
    tok = HashTokenizer()
    for s in range(0, len(texts), 50_000):
        _same(tok, texts[s : s + 50_000], threads=8)


def test_native_equals_python_random_unicode():
    rng = random.Random(7)
    pools = [
        "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 .,;:!?-_/()[]{}'\"\n\t",
        "àáâãäåæçèéêëìíîïñòóôõöøùúûüýÿÀÁÂÃÄÅÆÇÈÉÊËÌÍÎÏÑÒÓÔÕÖØÙÚÛÜÝŸßœŒšŠžŽ€£¥§©®°±µ¶·¿¡«»",
        "αβγδεζηθικλμνξοπρστυφχψωΑΒΓΔΕΖΗΘΙΚΛΜΝΞΟΠΡΣΤΥΦΧΨΩςάέήίόύώ",
        "абвгдежзийклмнопрстуфхцчшщъыьэюяАБВГДЕЖЗИЙКЛМНОПРСТУФХЦЧШЩЪЫЬЭЮЯёЁ",
        "中文字符日本語のテキスト한국어텍스트漢字仮名カタカナ",
        "̣́̀​‍   　﻿�İıİıẞ",
        "😀🎉🚀👍🏽🇺🇳𝔘𝔫𝔦𝔠𝔬𝔡𝔢",
    ]
    texts = []
    for _ in range(3000):
        n = rng.randint(0, 400)
        pool = "".join(rng.sample(pools, rng.randint(1, 3)))
        texts.append("".join(rng.choice(pool) for _ in range(n)))
    _same(HashTokenizer(), texts)
    _same(HashTokenizer(max_len=64), texts[:500])


def _vocab():
    words = set()
    for t in _fixture_chunks():
        for w in t.lower().replace("\n", " ").split(" "):
            w = "".join(c for c in w if c.isalnum())
            if w:
                words.add(w)
                for cut in (2, 3, 4):
                    if len(w) > cut + 1:
                        words.add(w[:cut])
                        words.add("##" + w[cut:])
    specials = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789.,;:!?-_/()[]{}'\"#*`=+<>|\\@$%^&~")
    return specials + chars + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + sorted(words)


def test_native_wordpiece_equals_python_and_transformers(tmp_path):
    vocab = _vocab()
    tok = WordPieceTokenizer(vocab)
    chunks = _fixture_chunks() + ["unaffable résumé naïve", "x" * 150 + " tail", "中文 mixed 文本", ""]
    _same(tok, chunks)
    transformers = pytest.importorskip("transformers")
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    ref = transformers.BertTokenizer(str(vf), do_lower_case=True)
    ids, lens = tok.encode_batch(chunks[:40])
    for b, t in enumerate(chunks[:40]):
        want = ref.encode(t, add_special_tokens=True, truncation=True, max_length=256)
        assert ids[b, : lens[b]].tolist() == want, t[:60]
