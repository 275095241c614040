"""The N > 1 exchange step on CPU: world_size 2 over gloo.  The per-shard scorer here is the oracle
(test infrastructure); what is exercised is the shipped sharding / packing / all-gather code."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hybrid_rag_system_amd.dist import allgather_candidates, embed_data_parallel, gather_rows, shard_bounds
from oracle import search_oracle as so


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    rng = np.random.default_rng(99)
    C = rng.standard_normal((3001, 64)).astype(np.float32)
    Q = rng.standard_normal((5, 64)).astype(np.float32)
    return Q, C


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Q, C = _inputs()
    lo, hi = shard_bounds(C.shape[0], world, rank)
    keys = so.topk_keys(Q, C[lo:hi], 8, "l2", row_offset=lo)
    counts = (keys & np.uint64(7)).astype(np.int32)  # any payload tied to the key
    gk, gc = allgather_candidates(torch.from_numpy(keys.view(np.int64)), torch.from_numpy(counts))
    merged = so.merge(gk.numpy().view(np.uint64))
    # data-parallel indexing: each rank embeds the rows it owns; a stand-in encoder (the real one needs the GPU)
    texts = [f"chunk {i}" for i in range(11)]
    fake = lambda ts: torch.tensor([[float(t.split()[1]), len(t)] for t in ts], dtype=torch.float32).reshape(-1, 2)
    elo, ehi, mine = embed_data_parallel(fake, texts)
    full = gather_rows(mine, len(texts))
    ret[rank] = (merged, gk.numpy().view(np.uint64), gc.numpy(), (elo, ehi), full.numpy())
    dist.destroy_process_group()


def test_two_rank_allgather_and_merge():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    Q, C = _inputs()
    want = so.topk_keys(Q, C, 8, "l2")
    for r in range(world):
        merged, gk, gc, bounds, full = ret[r]
        assert bounds == shard_bounds(11, world, r)
        assert np.array_equal(full, np.array([[i, len(f"chunk {i}")] for i in range(11)], np.float32))
        assert np.array_equal(merged, want)
        assert gk.shape == (2, 5, 8) and np.array_equal(gc, (gk & np.uint64(7)).astype(np.int32))
    assert np.array_equal(ret[0][1], ret[1][1])


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 10_000_000):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
