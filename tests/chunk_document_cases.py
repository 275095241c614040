"""Inputs of the chunk_document parity test, shared by the golden generator and the test: the fixture markdown files,
texts without headings, many small sections, edge cases and seeded random mixtures."""
import os
import random

CTX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "context", "infrastructure")


def build_cases():
    cases = {}
    for fn in sorted(os.listdir(CTX)):
        if fn.endswith(".md"):
            text = open(os.path.join(CTX, fn), encoding="utf-8").read()
            cases["file:" + fn] = (text, {})
            cases["file:" + fn + ":800/100"] = (text, {"chunk_size": 800, "overlap": 100})
            cases["file:" + fn + ":500/0"] = (text, {"chunk_size": 500, "overlap": 0})
    cases["empty"] = ("", {})
    cases["blank"] = ("   \n\n  ", {})
    cases["short"] = ("# Title\nhello", {})
    cases["no headings, long"] = ("word " * 3000, {})
    cases["one huge section"] = ("# A\n" + "x" * 7000 + "\n## B\nshort", {})
    cases["many small sections"] = ("".join(f"## S{i}\n" + "line of text\n" * (i % 7 + 1) for i in range(300)), {})
    cases["heading at the end"] = ("text " * 500 + "\n# Last", {})
    cases["level three is not a cut"] = ("### deep\n" + "abc " * 700 + "\n### deeper\n" + "def " * 700, {})
    rng = random.Random(7)
    for k in range(12):
        parts = []
        for _ in range(rng.randint(1, 40)):
            if rng.random() < 0.4:
                parts.append("#" * rng.randint(1, 3) + " " + "h" * rng.randint(1, 30) + "\n")
            parts.append("".join(rng.choice("ab \n.,") for _ in range(rng.randint(0, 1500))))
        cs = rng.choice([300, 1000, 2000])
        cases[f"random {k}"] = ("".join(parts), {"chunk_size": cs, "overlap": rng.choice([0, 50, cs // 4])})
    return cases
