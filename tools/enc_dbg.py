"""Debug: compare the dumped rows of the last block (HRS_ENC_DUMP, debug library) with the oracle's."""
import os, sys
sys.path.insert(0, '.')
os.environ["HRS_ENC_DUMP"] = "/tmp/enc_dump.bin"
import numpy as np, torch, torch.nn.functional as F
from hybrid_rag_system_amd import weights as W
from hybrid_rag_system_amd.engine import Engine
from oracle import encoder_oracle as EO
B, S = int(sys.argv[1]), int(sys.argv[2])
cfg = W.EncoderConfig(layers=1)
w = W.generate(cfg, 2026)
eng = Engine(0); enc = eng.encoder(w, cfg)
rng = np.random.default_rng(1)
ids = rng.integers(1000, 30000, (B, S)).astype(np.int32); lens = np.full(B, S, np.int32)
got = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
emb, hidden = EO.forward(w, ids, lens, cfg, return_hidden=True)
raw = open("/tmp/enc_dump.bin", "rb").read()
M = int(np.frombuffer(raw[:4], np.int32)[0]); H = 384
o = 4
def bf(buf): return torch.frombuffer(bytearray(buf), dtype=torch.bfloat16).float().numpy()
z = bf(raw[o:o + M * H * 2]).reshape(M, H); o += M * H * 2
y = bf(raw[o:o + M * H * 2]).reshape(M, H); o += M * H * 2
st = np.frombuffer(raw[o:o + 3 * M * 8], np.float32).reshape(3, M, 2)
T = B * S
t = {k: torch.as_tensor(v) for k, v in w.items()}
# oracle pieces
x0 = hidden[0].reshape(T, H)
# recompute oracle y (LN1 output) and z2 = y + ffn
q = (hidden[0] @ t["l0.q_w"].T + t["l0.q_b"]).view(B, S, 12, 32).transpose(1, 2)
k = (hidden[0] @ t["l0.k_w"].T + t["l0.k_b"]).view(B, S, 12, 32).transpose(1, 2)
v = (hidden[0] @ t["l0.v_w"].T + t["l0.v_b"]).view(B, S, 12, 32).transpose(1, 2)
a = (torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, -1) @ v).transpose(1, 2).reshape(B, S, H)
y_o = F.layer_norm(hidden[0] + a @ t["l0.o_w"].T + t["l0.o_b"], (H,), t["l0.ln1_g"], t["l0.ln1_b"], cfg.eps)
z_o = y_o + F.gelu(y_o @ t["l0.i_w"].T + t["l0.i_b"]) @ t["l0.d_w"].T + t["l0.d_b"]
y_o, z_o = y_o.reshape(T, H).numpy(), z_o.reshape(T, H).numpy()
print("y  : max err", np.abs(y[:T] - y_o).max(), "mean", np.abs(y[:T] - y_o).mean(), "|y| mean", np.abs(y_o).mean())
print("z2 : max err", np.abs(z[:T] - z_o).max(), "mean", np.abs(z[:T] - z_o).mean(), "|z| mean", np.abs(z_o).mean())
s1 = st[:, :T, 0].sum(0); s2 = st[:, :T, 1].sum(0)
print("sum: max err vs dumped rows", np.abs(s1 - z[:T].sum(1)).max(), " sumsq:", np.abs(s2 - (z[:T] ** 2).sum(1)).max())
print("per-fg sums err", [float(np.abs(st[g, :T, 0] - z[:T, 128 * g:128 * g + 128].sum(1)).max()) for g in range(3)])
blk = np.abs(z[:T] - z_o).reshape(T, 12, 32).max(axis=(0, 2)); print("z2 err per 32-feature block", np.round(blk, 4))
# pool in Python from the dumped rows (exact LN of the stored values) and compare with the kernel's output and the oracle
zz = torch.as_tensor(z[:T]).reshape(B, S, H)
pooled = F.layer_norm(zz, (H,), t["l0.ln2_g"], t["l0.ln2_b"], cfg.eps).mean(1)
e2 = F.normalize(pooled, dim=1).numpy()
print("pool kernel vs python pool of dumped rows:", np.abs(got - e2).max())
print("python pool of dumped rows vs oracle     :", np.abs(e2 - emb.numpy()).max())
print("kernel output vs oracle                  :", np.abs(got - emb.numpy()).max())
# what a bf16 pipeline should give: round the oracle's own z2 / LN output to bf16 and pool
zo = torch.as_tensor(z_o).reshape(B, S, H)
e3 = F.normalize(F.layer_norm(zo.bfloat16().float(), (H,), t["l0.ln2_g"], t["l0.ln2_b"], cfg.eps).mean(1), dim=1).numpy()
print("oracle z2 rounded to bf16, pooled vs oracle:", np.abs(e3 - emb.numpy()).max())
err = (z[:T] - z_o)
print("z2 error: mean over all", err.mean(), " per-token mean abs max", np.abs(err.mean(1)).max(), " per-feature mean abs max", np.abs(err.mean(0)).max())
yerr = (y[:T] - y_o)
print("y  error: mean over all", yerr.mean(), " per-feature mean abs max", np.abs(yerr.mean(0)).max())
pf = err.mean(0)
top = np.argsort(-np.abs(pf))[:24]
print("features with the largest mean error:", [(int(f), round(float(pf[f]), 4)) for f in top])
b2 = w["l0.d_b"]
print("d_b at those:", [round(float(b2[f]), 4) for f in top])
print("corr(mean err, d_b) =", np.corrcoef(pf, b2)[0, 1])
print("mean |mean err| by f%32:", np.round(np.abs(pf).reshape(12, 32).mean(0), 4))
sel = np.arange(H)[(np.arange(H) % 32) >= 24]
for sh in range(-40, 41):
    src = np.clip(np.arange(H) + sh, 0, H - 1)
    c = np.corrcoef(pf[sel], (b2[src] - b2)[sel])[0, 1]
    if abs(c) > 0.5:
        print("shift", sh, "corr", round(float(c), 3))
