"""The two stripe kernels at 25 .. 400 workgroups (B = 16 .. 256 sequences of 50): does a workgroup's time depend on how
many others read the same weight images at the same time (a shared L2 limit) or not (a per-CU limit)?
python3 tools/chain_scale.py on a GPU box."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from mtamrecommender_amd import hip_ops as ops
from bench import time_kernel
D, L = 128, 50
f = lambda *s: torch.randn(s, device="cuda") * 0.1
V, C, U = 3709, 304, 4835
T = dict(item=f(V, D), cat=f(C, D), pos=f(L + 3, D), user=f(U, D))
W4, Wx, bx = f(2 * D, D), f(D, 384), f(384)
img = torch.zeros(ops.seq_chain_images_elems(0, 384), dtype=torch.bfloat16, device="cuda")
imgr = torch.zeros_like(img)
for which, W in ((0, W4), (2, Wx)):
    o = ops.seq_chain_image_offset(which, 384)
    ops.split_weight_images(W, img[o:]); ops.split_weight_rows(W, imgr[o:])
g = torch.Generator().manual_seed(1)
for B in (16, 32, 64, 128, 160, 256):
    R = B * L
    ids = dict(item=torch.randint(0, V, (B, L), generator=g).int().cuda(), cat=torch.randint(0, C, (B, L), generator=g).int().cuda(),
               pos=torch.arange(L).repeat(B, 1).int().cuda(), user=torch.randint(0, U, (B,), generator=g).int().cuda())
    ic, user, zr, x, xproj = f(R, 2 * D), f(B, D), f(R, D), f(R, D), f(R, 384)
    l2 = torch.zeros(ops.seq_chain_gather_partials(B, L), device="cuda")
    fwd = lambda: ops.seq_chain_gather_fwd(T["item"], T["cat"], T["pos"], T["user"], ids["item"], ids["cat"], ids["pos"], ids["user"],
                                           B, L, 1, W4, None, None, Wx, bx, ic, user, l2, zr, x, None, xproj, w_images=img)
    d_xproj, d_xt, d_x, d_z, d_ic = f(R, 384), f(R, D), f(R, D), f(R, D), f(R, 2 * D)
    bwd = lambda: ops.seq_chain_bwd(d_xproj, None, d_xt, zr, R, d_x, d_z, d_ic, imgr)
    print("B %3d  workgroups %3d  fwd %.2f us  bwd %.2f us" % (B, (R + 31) // 32, time_kernel(fwd, torch) * 1e6,
                                                             time_kernel(bwd, torch) * 1e6))
