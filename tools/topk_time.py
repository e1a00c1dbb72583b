import sys, time
sys.path.insert(0, "/root/repo")
import torch
from mtamrecommender_amd import hip_ops as ops
for V in (3709, 1000003):
    ld = (V + 3) // 4 * 4
    s = torch.randn(128, ld, device="cuda")
    idx = torch.zeros(128, 50, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ops.topk(s, ld, 128, V, 50, idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ops.topk(s, ld, 128, V, 50, idx)
    torch.cuda.synchronize()
    print("topk 128 x %d: %.1f us" % (V, (time.perf_counter() - t0) / 10 * 1e6))
