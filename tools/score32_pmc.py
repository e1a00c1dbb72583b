"""Three launches of the fp32 logits-free scoring pair at V rows (run under rocprofv3 --pmc ...; GPU box).
usage: score32_pmc.py V [B]"""
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
for _ in range(3):
    ops.score32_lse(E, P, tgt, B, V, partial, lse, ce)
    ops.score32_bwd(E, P, lse, tgt, B, V, 1.0 / B, d_pred, dE, sq)
    torch.cuda.synchronize()
print("done")
