"""Developer timing: the clip launch (sqnorm + ticket + last-workgroup reduction) against a ticket-less partial pass
over the same dense gradient, ml-1m sizes; run under rocprofv3 --kernel-trace --stats."""
import sys
import torch
sys.path.insert(0, ".")
from mtamrecommender_amd import hip_ops as ops

n_dense = 61 * ops.adam_block()
n_part = ops.sqnorm_blocks(n_dense) + 464 + 4832
g = torch.randn(n_dense, device="cuda")
part = torch.rand(n_part + 8, device="cuda")
scale = torch.zeros(2, device="cuda")
lr = torch.full((1,), 1e-3, device="cuda")
state = torch.tensor([0.0, 0.9, 0.999, 1e-8, 0.9, 0.999, 0.0, 0.0], device="cuda")
ticket = torch.zeros(4, dtype=torch.int32, device="cuda")
l2 = torch.rand(4832, device="cuda")
ce = torch.rand(128, device="cuda")
loss = torch.zeros(4, device="cuda")
for _ in range(200):
    ops.sqnorm_clip_scale(g, n_dense, part, 0, n_part, 5.0, scale, lr, state, ticket, l2, 4832, ce, 128, 1e-5, 1 / 128., loss)
    ops.sqnorm_partial(g, n_dense, part)
torch.cuda.synchronize()
print("blocks", ops.sqnorm_blocks(n_dense), "partials", n_part)
