"""Developer timing of the fp32 logits-free scoring pair at large V (GPU box), with the lab switches of
x3::bwd_pc_kernel (MTAM_SCORE32_LAB bits: 1 no dE stores, 2 no G[v][b] element writes, 4 no E^T gathers, 8 no exp).
usage: score32_time.py V [B]"""
import os
import sys
import torch
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.build()
from mtamrecommender_amd import hip_ops as ops

V = int(sys.argv[1]) if len(sys.argv) > 1 else 10000003
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
D = 128
torch.manual_seed(0)
E = torch.empty((V, D), device="cuda").uniform_(-0.2165, 0.2165)
P = torch.randn((B, D), device="cuda")
tgt = torch.randint(0, V, (B,), device="cuda", dtype=torch.int32)
lse, ce = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
partial = torch.zeros(ops.score32_partials(B, V), device="cuda")
sq = torch.zeros(ops.score32_sq_partials(V), device="cuda")
d_pred = torch.zeros((B, D), device="cuda")
dE = torch.empty((V, D), device="cuda")


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("V = %d, B = %d" % (V, B))
print("lse pass                 %.3f ms" % timeit(lambda: ops.score32_lse(E, P, tgt, B, V, partial, lse, ce)))
for lab, what in ((0, "backward, product build"), (1, "  no dE stores"), (2, "  no G[v][b] element writes"),
                  (4, "  no E^T gathers"), (8, "  no exp"), (7, "  none of the three"), (15, "  none of the four")):
    os.environ["MTAM_SCORE32_LAB"] = str(lab)
    print("%-28s %.3f ms" % (what, timeit(lambda: ops.score32_bwd(E, P, lse, tgt, B, V, 1.0 / B, d_pred, dE, sq))))
os.environ["MTAM_SCORE32_LAB"] = "0"
