"""Randomised differential check of hrs_score_topk (both modes, both shadows) against the CPU oracle."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd.engine import Engine
from oracle import search_oracle as so


def run(eng, seed, cases, verbose=True, large=False):
  rng = np.random.default_rng(seed)
  bad = 0
  for it in range(cases):
      n = int(rng.choice([4096, 5000, 9999, 33000, 70001, 131072]))
      d = int(rng.choice([4, 8, 60, 64, 100, 128, 132, 256, 384, 388, 512, 516, 768, 1024])) 
      b = int(rng.choice([1, 2, 3, 5, 17, 64, 100, 257, 300, 513, 700]))  # (from 513: the 16 x 16 x 64 main pass)
      if large:  # slices of many tiles: the sampled passes really sample (stride > 1)
          n = int(rng.choice([400_000, 1_000_003, 2_000_000]))
          d = int(rng.choice([64, 128, 384, 768, 1024]))
          b = int(rng.choice([1, 4, 5, 33, 64, 130, 300, 600, 600, 1024]))
          while n * d * b > 1.2e11: b = max(1, b // 2)
      k = int(rng.choice([1, 3, 10, 20, 64, 128]))
      if n < k * 64: k = 10
      space = str(rng.choice(["l2", "cosine", "ip"]))
      kind = rng.integers(0, 7)
      C = rng.standard_normal((n, d)).astype(np.float32)
      if kind == 0: C /= np.linalg.norm(C, axis=1, keepdims=True)
      elif kind == 1: C *= rng.uniform(0.01, 10, (n, 1)).astype(np.float32)
      elif kind == 2: C = np.round(C * 2).astype(np.float32) / 2      # many exact ties
      elif kind == 3: C[rng.integers(0, n, n // 10)] = C[0]                # duplicates
      elif kind == 4:                                                  # tight clusters: near-ties by the thousand
          cen = rng.standard_normal((int(rng.choice([2, 8, 50])), d)).astype(np.float32)
          C = (cen[rng.integers(0, len(cen), n)] + np.float32(rng.choice([1e-3, 0.05, 0.3])) * C).astype(np.float32)
      elif kind == 5:                                                  # zero rows, and a zero query below
          C[rng.integers(0, n, n // 20)] = 0
      else:                                                            # a few dominant components
          C[:, rng.integers(0, d, max(1, d // 16))] *= np.float32(100)
      Q = rng.standard_normal((b, d)).astype(np.float32)
      if rng.random() < 0.5: Q[0] = C[rng.integers(0, n)]
      if kind == 5: Q[-1] = 0
      if kind == 4: Q[: max(1, b // 2)] = C[rng.integers(0, n, max(1, b // 2))]
      use_mask = rng.random() < 0.3
      mask = (rng.random(n) < rng.choice([0.9, 0.3, 0.01])).astype(np.uint8) if use_mask else None
      want = so.topk_keys(Q, C, k, space, mask=mask)
      dC, dQ = torch.from_numpy(C).cuda(), torch.from_numpy(Q).cuda()
      dm = torch.from_numpy(mask).cuda() if use_mask else None
      grow = rng.random() < 0.3  # the corpus arrives in pieces (hrs_corpus_append) instead of in one registration
      cuts = sorted(set(int(x) for x in rng.integers(1, n, int(rng.integers(1, 4))))) + [n] if grow else [n]
      for shadow in ("i8", "bf16"):
          c = eng.corpus(dC, shadow=shadow, n=cuts[0])
          for cut in cuts[1:]:
              c.append(dC, cut)
          for mode in ("mfma", "exact", "auto"):
              if mode == "exact" and shadow == "bf16": continue
              dist, rows, keys = eng.score_topk(c, dQ, k, space=space, mask=dm, mode=mode, want_keys=True)
              torch.cuda.synchronize()
              wd, wr = so.unpack(want)  # the unpacked outputs come from the kernels that write the final keys
              if not (np.array_equal(rows.cpu().numpy(), wr) and np.array_equal(dist.cpu().numpy().view(np.uint32), wd.view(np.uint32))):
                  bad += 1
                  print(f"MISMATCH (unpacked) it={it} n={n} d={d} b={b} k={k} {space} kind={kind} mask={use_mask} {shadow} {mode}", flush=True)
              if not np.array_equal(keys.cpu().numpy().view(np.uint64), want):
                  bad += 1
                  print(f"MISMATCH it={it} n={n} d={d} b={b} k={k} {space} kind={kind} mask={use_mask} {shadow} {mode} {eng.score_stats()}", flush=True)
          c.close()
      if verbose: print(f"case {it}: n={n} d={d} b={b} k={k} {space} kind={kind} mask={use_mask} grow={cuts if grow else 0} ok", flush=True)
  return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    print("mismatches:", run(Engine(0), seed, cases, large=len(sys.argv) > 3 and sys.argv[3] == "large"))
