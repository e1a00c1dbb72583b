"""Developer timing of the catalog-scoring kernels at large V (GPU box): bf16 logits-free pair
(csrc/score16.hip) beside the fp32 GEMM + softmax sequence it replaces.  usage: score16_time.py V [B]"""
import sys
import torch
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.build()
from mtamrecommender_amd import hip_ops as ops

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1000003
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
FP32 = "--no-f32" not in sys.argv
D = 128
torch.manual_seed(0)
E = torch.empty((V, D), device="cuda").uniform_(-0.2165, 0.2165)
P = torch.randn((B, D), device="cuda")
tgt = torch.randint(0, V, (B,), device="cuda", dtype=torch.int32)
E16 = torch.empty((V, D), dtype=torch.bfloat16, device="cuda")
P16 = torch.zeros((ops.score16_batch_pad(B), D), dtype=torch.bfloat16, device="cuda")
lse, ce = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
partial = torch.zeros(ops.score16_partials(B, V), device="cuda")
sq = torch.zeros(ops.score16_sq_partials(V), device="cuda")
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


rows = []
rows.append(("f32_to_bf16 table", timeit(lambda: ops.f32_to_bf16(E.view(-1), E16.view(-1))), V * D * 6))
ops.f32_to_bf16(P.view(-1), P16.view(-1))
rows.append(("score16_lse", timeit(lambda: ops.score16_lse(E16, P16, tgt, B, V, partial, lse, ce)), V * D * 2))
rows.append(("score16_bwd", timeit(lambda: ops.score16_bwd(E16, P16, lse, tgt, B, V, 1.0 / B, d_pred, dE, sq)),
             V * D * 6))
if FP32:
    ld = (V + 3) // 4 * 4
    logits = torch.empty((B, ld), device="cuda")
    rows.append(("score16_logits", timeit(lambda: ops.score16_logits(E16, P16, B, V, logits, ld)),
                 V * D * 2 + B * V * 4))
    cep = torch.zeros(ops.softmax_ce_partials(B, V) + 4, device="cuda")
    rows.append(("f32 logits GEMM", timeit(lambda: ops.gemm(P, E, logits, trans_b=True)), V * D * 4 + B * V * 4))
    rows.append(("f32 softmax_ce + grad", timeit(lambda: ops.softmax_ce(logits, ld, tgt, B, V, 1.0 / B, lse, ce, logits,
                                                                        cep)), 3 * B * V * 4))
    rows.append(("f32 dE GEMM (+sq)", timeit(lambda: ops.gemm(logits, P, dE, trans_a=True, epilogue=ops.EPI_STORE_SQ,
                                                             aux_out=torch.zeros(ops.gemm_sq_partials(V, D),
                                                                                 device="cuda"), M=V)),
                 B * V * 4 + V * D * 4))
    split_v = max(1, min(64, (V + 127) // 128), min(1024, V // 2048))
    rows.append(("f32 d_pred GEMM", timeit(lambda: ops.gemm(logits, E, d_pred, epilogue=ops.EPI_ATOMIC, split_k=split_v,
                                                           K=V)), B * V * 4 + V * D * 4))
print("V = %d, B = %d" % (V, B))
for name, ms, nbytes in rows:
    print("%-24s %9.3f ms   %7.0f GB/s of its algorithmic bytes (%.2f GB)" % (name, ms, nbytes / ms / 1e6, nbytes / 1e9))
