#!/usr/bin/env python3
"""Generate tests/golden/text_goldens.json by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

The reference helper is imported from /root/reference/scripts; its
``get_collection`` is replaced by a stub object that answers ``query`` with
Chroma-shaped lists, because no ChromaDB server or package exists here.  The
output holds inputs and expected outputs only (no reference source text).
"""
import contextlib
import hashlib
import io
import json
import os
import sys

REF = os.environ.get("HRS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "scripts"))

import chromadb_helper as ref  # noqa: E402
import mcp_chromadb_bridge as refbridge  # noqa: E402

CORPUS = ["README.md", "automation-guide.md", "bash-linux-reference.md", "windows-powershell-reference.md"]

QUERIES = [
    "How do I restart the nginx container on Unraid?",
    "docker compose up fails with permission denied",
    "proxmox lxc container backup schedule",
    "What is the IP address of the firewall?",
    "powershell Get-ADUser filter examples",
    "bash loop over files in a directory",
    "the and of to in",
    "",
    "a",
    "Is it?",
    "UPPER lower MiXeD Case Words",
    "hyphen-ated words and under_scores and dots.in.names",
    "numbers 123 and v2 and ipv4 10.0.0.1",
    "unicode café naïve résumé",
    "repeat repeat repeat docker docker",
    "how to check disk usage on linux",
    "rsync backup to remote server with ssh key",
    "Windows service restart from command line",
    "vlan configuration unifi switch",
    "cron job every five minutes",
    "systemctl status failed unit",
    "grep recursive case insensitive",
    "ansible playbook run against inventory",
    "ssh tunnel port forwarding",
    "zfs pool scrub status",
    "certificate renewal letsencrypt",
    "What are the steps for troubleshooting DNS?",
    "git pull origin main conflict",
    "mount nfs share at boot",
    "who am i",
    "list all running containers",
    "powershell remoting enable",
    "active directory group membership",
    "array parity check unraid",
    "qemu guest agent install",
    "tail -f /var/log/syslog",
    "find files larger than 1GB",
    "chmod 755 script.sh",
    "environment variables in docker-compose.yml",
    "CPU temperature monitoring",
    "automation of backups; retention: 30 days, weekly",
    "don't can't won't",
    "e-mail notifications (smtp) setup!",
    "tabs\tand\nnewlines",
    "   leading and trailing spaces   ",
    "x y z ab cd",
    "gpu passthrough vm",
    "wireguard vpn peer config",
    "update all packages apt",
    "scheduled task windows powershell script",
]


class Stub:
    """Chroma-shaped collection returning canned rows."""

    name = "stub"
    metadata = {"description": "stub"}

    def __init__(self, docs, metas, ids, dists):
        self.rows = (docs, metas, ids, dists)
        self.calls = []

    def query(self, query_texts, n_results, include, where=None):
        self.calls.append(dict(n_results=n_results, include=list(include), where=where))
        docs, metas, ids, dists = self.rows
        k = min(n_results, len(docs))
        return {
            "ids": [ids[:k]],
            "documents": [docs[:k]],
            "metadatas": [metas[:k]],
            "distances": [dists[:k]],
        }


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest()


def main():
    out = {}
    texts = {n: open(os.path.join(HERE, "context", "infrastructure", n), encoding="utf-8", errors="ignore").read() for n in CORPUS}

    # G1: chunker over the four corpus files (+ a few synthetic shapes)
    g1 = {}
    for n, t in texts.items():
        g1[n] = [[c["chunk_index"], c["char_start"], c["char_end"], c["total_chunks"], sha(c["content"])] for c in ref.chunk_text(t)]
    out["chunk_corpus"] = g1
    synth = {
        "short": "tiny text",
        "empty": "",
        "nospace": "x" * 2500,
        "sentences": "Sentence number one is here. " * 120,
        "commas": "alpha, beta, gamma; delta: epsilon " * 90,
        "exact1000": "y" * 1000,
        "w1001": ("word " * 200) + "z",
        "nul": "a\x00b  c\n\n\nd " * 300,
    }
    out["chunk_synth"] = {
        k: dict(text=v, chunks=[[c["chunk_index"], c["char_start"], c["char_end"], c["total_chunks"], sha(c["content"])] for c in ref.chunk_text(v)])
        for k, v in synth.items()
    }
    out["chunk_params"] = [
        dict(size=s, overlap=o, chunks=[[c["char_start"], c["char_end"]] for c in ref.chunk_text(texts["README.md"], s, o)])
        for s, o in ((500, 100), (300, 0), (2000, 400))
    ]
    out["clean"] = [[s, ref.clean_text(s)] for s in ["  a \n\n b\t\tc  ", "x\x00y", "", "\n", "no change"]]

    # G2: doc ids
    out["doc_id"] = [
        dict(content_sha=sha(texts[n]), name=n, source=f"/data/ctx/{n}", id=ref.generate_doc_id(texts[n], f"/data/ctx/{n}")) for n in CORPUS
    ] + [dict(content="hello", source="", id=ref.generate_doc_id("hello")), dict(content="hello", source="s.md", id=ref.generate_doc_id("hello", "s.md"))]

    # metadata extraction (a11)
    md_cases = [(texts[n], f"/data/ctx/{n}") for n in CORPUS] + [
        ("# Title here\nunraid array docker", None),
        ("no heading powershell", "docs/howto/x.md"),
        ("nothing", "a/troubleshoot/b.txt"),
        ("pve qemu\n# Late  heading  \n", "x/api/config.yaml"),
        ("plain", "x/config/y.json"),
        ("# " + "T" * 300, "general.md"),
    ]
    out["metadata"] = [dict(content_sha=sha(c), content=(c if len(c) < 400 else None), filepath=p, md=ref.extract_metadata_from_content(c, p)) for c, p in md_cases]
    for e, (c, p) in zip(out["metadata"], md_cases):
        if e["content"] is None:
            e["corpus_name"] = os.path.basename(p)

    # G3: keywords (sorted; the reference's list order depends on the hash seed)
    out["keywords"] = [[q, sorted(ref.extract_keywords(q))] for q in QUERIES]

    # G4: keyword score
    docs = [c["content"] for n in CORPUS for c in ref.chunk_text(texts[n])]
    kw_cases = [
        ("Dockerized services run here", ["docker", "nginx"]),
        ("anything", []),
        ("", ["x"]),
        ("UPPER CASE DOC with Nginx", ["nginx", "upper", "missing"]),
    ]
    for q in QUERIES[:20]:
        kws = sorted(ref.extract_keywords(q))
        for d in (docs[0], docs[17], docs[40], docs[62]):
            kw_cases.append((d, kws))
    out["keyword_score"] = [dict(doc_sha=sha(d), doc=(d if len(d) < 200 else None), kws=k, score=ref.calculate_keyword_score(d, k)) for d, k in kw_cases]
    out["docs_sha"] = [sha(d) for d in docs]

    # G5: hybrid_query through the stub
    g5 = []
    cases = [
        ("restart docker container", 5, list(range(0, 10)), [0.31, 0.35, 0.35, 0.62, 0.88, 0.99, 1.0, 1.01, 1.4, 1.93]),
        ("powershell active directory users", 3, [50, 51, 52, 53, 54, 55], [0.2, 0.2, 0.2, 0.2, 0.2, 0.2]),
        ("the of and", 5, list(range(20, 30)), [0.5 + 0.01 * i for i in range(10)]),
        ("backup schedule cron", 5, [3, 9, 27], [1.2, 1.3, 1.5]),
        ("anything at all", 5, [], []),
        ("unraid array parity", 1, [1, 2], [0.7500000238418579, 0.75]),
        ("linux bash grep find", 4, list(range(30, 38)), [1.0, 1.0000001192092896, 0.9999999403953552, 2.0, 0.0, 1e-7, 0.5, 0.25]),
    ]
    for q, n, rows, dists in cases:
        d = [docs[r] for r in rows]
        metas = [{"source": f"doc{r}.md", "row": r} for r in rows]
        ids = [f"id-{r:04d}" for r in rows]
        stub = Stub(d, metas, ids, dists)
        ref.get_collection = lambda name=None, _s=stub: _s
        res = ref.hybrid_query(q, n, "whatever")
        g5.append(
            dict(
                query=q,
                n=n,
                rows=rows,
                distances=dists,
                stub_calls=stub.calls,
                result=[
                    dict(
                        id=r["id"],
                        metadata=r["metadata"],
                        semantic_score=r["semantic_score"],
                        semantic_is_int=isinstance(r["semantic_score"], int),
                        keyword_score=r["keyword_score"],
                        combined_score=r["combined_score"],
                        distance=r["distance"],
                        doc_sha=sha(r["document"]),
                    )
                    for r in res
                ],
                bridge_json=json.dumps(refbridge.format_hybrid_results(res, 800)),
                bridge_json_100=json.dumps(refbridge.format_hybrid_results(res, 100)),
                cli_json=json.dumps(res, indent=2),
            )
        )
    out["hybrid"] = g5

    # G6: plain query + bridge formatter + error path
    g6 = []
    for q, n, rows, dists in cases[:4] + [cases[6]]:
        d = [docs[r] for r in rows]
        metas = [{"source": f"doc{r}.md", "row": r} for r in rows]
        ids = [f"id-{r:04d}" for r in rows]
        stub = Stub(d, metas, ids, dists)
        ref.get_collection = lambda name=None, _s=stub: _s
        res = ref.query(q, n, "c", {"system": "docker"} if n == 3 else None)
        g6.append(dict(query=q, n=n, rows=rows, distances=dists, stub_calls=stub.calls, bridge_json=json.dumps(refbridge.format_query_results(res, 800)), cli_json=json.dumps(res, indent=2)))
    out["plain"] = g6

    class Boom:
        def query(self, **kw):
            raise RuntimeError("collection exploded")

    ref.get_collection = lambda name=None: Boom()
    err = ref.query("x")
    out["error"] = dict(result=err, bridge_json=json.dumps(refbridge.format_query_results(err)))

    # G7: CLI text output of the hybrid branch (main() printing), via the stub
    q, n, rows, dists = cases[0]
    stub = Stub([docs[r] for r in rows], [{"source": f"doc{r}.md", "system": "docker"} if r % 2 else {} for r in rows], [f"id-{r}" for r in rows], dists)
    ref.get_collection = lambda name=None, _s=stub: _s
    buf = io.StringIO()
    argv = sys.argv
    sys.argv = ["chromadb_helper.py", "query", q, "--hybrid", "-n", str(n)]
    with contextlib.redirect_stdout(buf):
        ref.main()
    sys.argv = argv
    out["cli_text_hybrid"] = dict(query=q, n=n, rows=rows, distances=dists, stdout=buf.getvalue())

    with open(os.path.join(HERE, "text_goldens.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "text_goldens.json"), "chunks:", [len(v) for v in g1.values()])


if __name__ == "__main__":
    main()
