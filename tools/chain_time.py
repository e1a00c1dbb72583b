"""The two stripe kernels with and without the K/V work (what riding with the GRU launches saves), and the two
products as launches of their own.  python3 tools/chain_time.py on a GPU box."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mtamrecommender_amd import hip_ops as ops
from bench import time_kernel
D, B, L = 128, 128, 50
R = B * L
f = lambda *s: torch.randn(s, device="cuda") * 0.1
V, C, U = 3709, 304, 4835
T = dict(item=f(V, D), cat=f(C, D), pos=f(L + 3, D), user=f(U, D))
g = torch.Generator().manual_seed(1)
ids = dict(item=torch.randint(0, V, (B, L), generator=g).int().cuda(), cat=torch.randint(0, C, (B, L), generator=g).int().cuda(),
           pos=torch.arange(L).repeat(B, 1).int().cuda(), user=torch.randint(0, U, (B,), generator=g).int().cuda())
W4, Wkv, bkv, Wx, bx = f(2 * D, D), f(D, 256), f(256), f(D, 384), f(384)
def images(n_kv):
    buf = torch.zeros(ops.seq_chain_images_elems(n_kv, 384), dtype=torch.bfloat16, device="cuda")
    bufr = torch.zeros_like(buf)
    for which, W in ((0, W4), (1, Wkv if n_kv else None), (2, Wx)):
        if W is not None:
            o = ops.seq_chain_image_offset(which, 384)
            ops.split_weight_images(W, buf[o:]); ops.split_weight_rows(W, bufr[o:])
    return buf, bufr
ic, user, zr, x, kv, xproj = f(R, 2 * D), f(B, D), f(R, D), f(R, D), f(R, 256), f(R, 384)
l2 = torch.zeros(ops.seq_chain_gather_partials(B, L), device="cuda")
for n_kv in (256, 0):
    img, imgr = images(n_kv)
    fwd = lambda: ops.seq_chain_gather_fwd(T["item"], T["cat"], T["pos"], T["user"], ids["item"], ids["cat"], ids["pos"], ids["user"],
                                           B, L, 1, W4, Wkv if n_kv else None, bkv if n_kv else None, Wx, bx, ic, user, l2, zr, x,
                                           kv if n_kv else None, xproj, w_images=img)
    d_xproj, d_kv, d_xt, d_x, d_z, d_ic = f(R, 384), f(R, 256), f(R, D), f(R, D), f(R, D), f(R, 2 * D)
    bwd = lambda: ops.seq_chain_bwd(d_xproj, d_kv if n_kv else None, d_xt, zr, R, d_x, d_z, d_ic, imgr)
    print("n_kv", n_kv, "fwd %.2f us" % (time_kernel(fwd, torch) * 1e6), "bwd %.2f us" % (time_kernel(bwd, torch) * 1e6))
# the stand-alone K/V projection and d_kv Wkv^T as GEMMs (what the co-scheduled roles would have to do)
t1 = time_kernel(lambda: ops.gemm(x, Wkv, kv, epilogue=ops.EPI_BIAS_RELU, bias=bkv), torch)
t2 = time_kernel(lambda: ops.gemm(d_kv, Wkv, d_x, trans_b=True, epilogue=ops.EPI_ACCUM), torch)
print("kv gemm %.2f us, d_kv Wkv^T gemm %.2f us" % (t1 * 1e6, t2 * 1e6))
