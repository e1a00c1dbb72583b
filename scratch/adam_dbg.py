import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mtamrecommender_amd import hip_ops as ops
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(11)
blk = ops.adam_block()
n = 3 * blk + 128 * 40 + 8
mats = [(256, 16, 128), (256 + 16 * 128 + 4, 128, 36), (2 * blk - 8 * 128, 8, 128)]
g, p0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
m0, v0 = (rng.standard_normal(n) * 0.1).astype(np.float32), rng.uniform(0, 0.1, n).astype(np.float32)
scale = dev(np.array([0.7, 1.0], np.float32)); hyper = dev(np.array([1e-3, 0.9, 0.999, 1e-8], np.float32))
p, m, v = dev(p0), dev(m0), dev(v0)
ops.adam(p, m, v, dev(g), n, scale, hyper, 2 * blk)
p2, m2, v2 = dev(p0), dev(m0), dev(v0)
imgs = [torch.full((3 * K * N,), 7.0, dtype=torch.bfloat16, device="cuda") for _, K, N in mats]
descs = ops.weight_image_descs([(b, K, N, im) for (b, K, N), im in zip(mats, imgs)])
ops.adam_images(p2, m2, v2, dev(g), n, scale, hyper, 2 * blk, descs)
torch.cuda.synchronize()
for name, a, b in (("p", p, p2), ("m", m, m2), ("v", v, v2)):
    d = (a - b).abs()
    idx = torch.nonzero(d > 0).flatten()
    print(name, "max diff", float(d.max()), "count", idx.numel(), "first", idx[:8].tolist(), "rel", float((d / (a.abs() + 1e-30)).max()))
    unchanged = torch.nonzero(b == dev({"p": p0, "m": m0, "v": v0}[name])).flatten()
    print("  unchanged elements", unchanged.numel(), unchanged[:8].tolist())
