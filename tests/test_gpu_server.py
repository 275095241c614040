"""The resident engine on the GPU: `python -m hybrid_rag_system_amd.server` as a child process, the CLI and
the bridge as one-shot clients of it (HRS_SERVER), their output equal to the in-process helper's."""
import contextlib
import io
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORPUS_DIR = os.path.join(ROOT, "tests", "golden", "context", "infrastructure")
QUERY = "How do I restart the nginx container on Unraid?"


def _cli(script, args, env):
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + args, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout, time.time() - t0


def test_cli_and_bridge_through_a_resident_engine(tmp_path, monkeypatch):
    sock = str(tmp_path / "hrs.sock")
    env = dict(os.environ, HRS_DATA_DIR=str(tmp_path / "served"), PYTHONPATH=ROOT)
    env.pop("HRS_SERVER", None)
    srv = subprocess.Popen([sys.executable, "-u", "-m", "hybrid_rag_system_amd.server", "--socket", sock], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    try:
        line = srv.stdout.readline()
        assert line, srv.stderr.read()
        hello = json.loads(line)
        assert hello["ready"] and hello["socket"] == sock
        cenv = dict(env, HRS_SERVER=sock)
        out, _ = _cli("chromadb_helper.py", ["index", CORPUS_DIR, "-c", "cfg1", "-p", "*.md"], cenv)
        res = json.loads(out)
        assert res["status"] == "success" and res["indexed"] == 4 and sorted(f["chunks"] for f in res["files"]) == [6, 14, 15, 28]
        hybrid_json, t_hybrid = _cli("chromadb_helper.py", ["query", QUERY, "--hybrid", "--json", "-n", "5", "-c", "cfg1"], cenv)
        plain_json, _ = _cli("chromadb_helper.py", ["query", QUERY, "--json", "-n", "3", "-c", "cfg1", "--system", "unraid"], cenv)
        bridge_json, t_bridge = _cli("mcp_chromadb_bridge.py", ["query", QUERY, "--hybrid", "--n-results", "3", "--collection", "cfg1"], cenv)
        stats_json, _ = _cli("chromadb_helper.py", ["stats", "-c", "cfg1"], cenv)
        assert json.loads(stats_json)["count"] == 63
        # a client never loads torch or the library
        probe = "import sys; sys.path.insert(0, %r); from hybrid_rag_system_amd import helper; r = helper.hybrid_query(%r, 5, 'cfg1'); assert len(r) == 5; assert 'torch' not in sys.modules and 'numpy' not in sys.modules" % (ROOT, QUERY)
        r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=cenv, timeout=60)
        assert r.returncode == 0, r.stderr
        print(f"one-shot CLI through the resident engine: hybrid query {t_hybrid*1e3:.0f} ms, bridge {t_bridge*1e3:.0f} ms")
    finally:
        srv.terminate()
        try:
            srv.wait(20)
        except subprocess.TimeoutExpired:
            srv.kill()
    assert not os.path.exists(sock)

    # the same collection, indexed and queried in this process (no server): equal output, byte for byte
    monkeypatch.setenv("HRS_DATA_DIR", str(tmp_path / "local"))
    monkeypatch.delenv("HRS_SERVER", raising=False)
    import importlib

    from hybrid_rag_system_amd import bridge, helper

    importlib.reload(helper)
    importlib.reload(bridge)
    helper.index_directory(CORPUS_DIR, ["*.md"], "cfg1")

    def run(mod, argv):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            mod.main(argv)
        return buf.getvalue()

    def strip_times(text):  # indexed_at / filepath differ between the two runs by construction
        obj = json.loads(text)
        rows = obj if isinstance(obj, list) else (obj.get("results") or [m for ms in obj.get("metadatas", []) for m in ms])
        for r in rows:
            meta = r.get("metadata", r) if isinstance(r, dict) else {}
            meta.pop("indexed_at", None)
        return obj

    assert strip_times(run(helper, ["query", QUERY, "--hybrid", "--json", "-n", "5", "-c", "cfg1"])) == strip_times(hybrid_json)
    assert strip_times(run(helper, ["query", QUERY, "--json", "-n", "3", "-c", "cfg1", "--system", "unraid"])) == strip_times(plain_json)
    assert strip_times(run(bridge, ["query", QUERY, "--hybrid", "--n-results", "3", "--collection", "cfg1"])) == strip_times(bridge_json)
    # the served directory persists like the local one: a fresh in-process client reads what the server wrote
    from hybrid_rag_system_amd.store import Client

    assert Client(str(tmp_path / "served")).get_or_create_collection("cfg1").count() == 63
