"""Diagnostic: per-phase cycle totals of tagru_fwd (needs a -DMTAM_GRU_STAMPS build of the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mtamrecommender_amd import hip_ops as ops
B, L, D = 128, 50, 128
rng = np.random.default_rng(0)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
R = B * L
xproj = dev(rng.standard_normal((R, 3 * D)).astype(np.float32) * 0.1)
x = dev(rng.standard_normal((R, D)).astype(np.float32) * 0.1)
tl = dev(np.floor(rng.exponential(24, R)).astype(np.float32))
sl = dev(np.full(B, L, np.int32))
whg = dev(rng.standard_normal((D, 2 * D)).astype(np.float32) * 0.05)
whc = dev(rng.standard_normal((D, D)).astype(np.float32) * 0.05)
tvec = dev(rng.standard_normal((8, D)).astype(np.float32) * 0.05)
hs = torch.zeros((R, D), device="cuda"); short = torch.zeros((B, D), device="cuda"); save = torch.zeros((R, 5 * D), device="cuda")
for _ in range(3):
    ops.tagru_fwd(xproj, x, tl, sl, whg, whc, tvec, B, L, hs, short, save)
torch.cuda.synchronize()
h = hs.cpu().numpy().reshape(B, L, D)
steps = L - 1
names = ["loop top->", "gate FMA", "gate finalize (->barrier A arrive)", "barrier A wait", "cand FMA", "cand finalize", "barrier B wait", "-"]
for b in (0, 64, 127):
    g, c = h[b, steps, 0:8] / steps, h[b, steps, 8:16] / steps
    print("block", b)
    for i in range(7):
        print("  %-36s gate wave %8.0f   cand wave %8.0f cycles/step" % (names[i], g[i], c[i]))
    print("  total per step: gate %.0f cand %.0f" % (g[1:7].sum() + g[0], c[1:7].sum() + c[0]))
