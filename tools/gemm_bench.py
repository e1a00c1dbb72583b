"""Developer micro-benchmark of mtam_gemm_f32 on the shapes of the training step (graph-timed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mtamrecommender_amd import hip_ops as ops  # noqa: E402


def timeit(fn, reps=20, replays=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * replays)


def run(name, M, N, K, ta=False, tb=False, epi=0, split=1):
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    C = torch.zeros((M, N), device="cuda")
    kw = {}
    if epi in (1, 2):
        kw["bias"] = torch.randn(N, device="cuda")
    if epi in (3, 5):
        kw["aux_in"] = torch.randn((M, N), device="cuda")
        kw["aux_out"] = torch.zeros((M, N), device="cuda")
    us = timeit(lambda: ops.gemm(A, B, C, trans_a=ta, trans_b=tb, epilogue=epi, split_k=split, **kw))
    print("%-22s M=%5d N=%5d K=%5d ta=%d tb=%d epi=%d split=%2d : %8.2f us %7.1f TFLOP/s"
          % (name, M, N, K, ta, tb, epi, split, us, 2.0 * M * N * K / us / 1e6), flush=True)


if __name__ == "__main__":
    run("dense4emb relu_add", 6400, 128, 256, epi=3)
    run("xproj bias", 6400, 384, 128, epi=1)
    run("kv bias_relu", 6400, 256, 128, epi=2)
    run("logits NT", 128, 3709, 128, tb=True)
    run("d_ic NT", 6400, 256, 128, tb=True)
    run("dx accum NT", 6400, 128, 256, tb=True, epi=4)
    run("dx accum_mask NT", 6400, 128, 384, tb=True, epi=5)
    run("dE TN", 3709, 128, 128, ta=True)
    run("dpred atomic", 128, 128, 3709, epi=6, split=29)
    run("dWx TN atomic", 128, 384, 6400, ta=True, epi=6, split=16)
    run("logits V=1M NT", 128, 1000003, 128, tb=True)
    run("dE V=1M TN", 1000003, 128, 128, ta=True)
    run("dpred V=1M atomic", 128, 128, 1000003, epi=6, split=488)
    run("big square", 4096, 4096, 4096)
    run("big square NT", 4096, 4096, 4096, tb=True)
    run("tall K=32", 6400, 128, 32)
    run("tall K=64", 6400, 128, 64)
    run("tall K=128", 6400, 128, 128)
    run("tall K=512", 6400, 128, 512)
    run("M=12800 K=256", 12800, 128, 256)
    run("M=25600 K=256", 25600, 128, 256)
