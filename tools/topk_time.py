"""Developer timing of mtam_topk: single pass vs the two-level form (workspace) for long rows."""
import sys, time
sys.path.insert(0, ".")
import torch
from mtamrecommender_amd import hip_ops as ops
for V in (3709, 1000003, 10000003, 50000003):
    ld = (V + 3) // 4 * 4
    s = torch.randn(128, ld, device="cuda")
    idx = torch.zeros(128, 50, dtype=torch.int32, device="cuda")
    nbytes = ops.topk_workspace_bytes(128, V, 50)
    for ws in ([None] + ([torch.empty(nbytes // 4, device="cuda")] if nbytes else [])):
        n = 3 if V > 5000000 else 10
        ops.topk(s, ld, 128, V, 50, idx, workspace=ws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ops.topk(s, ld, 128, V, 50, idx, workspace=ws)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("topk 128 x %d %s: %.1f us (%.0f GB/s of one read of the rows)"
              % (V, "two-level" if ws is not None else "one pass ", dt * 1e6, 128 * V * 4 / dt / 1e9), flush=True)
    del s
