"""Embedding gather / scatter-add roofline sweep (developer tool, GPU box only).

    python3 tools/emb_roofline.py sweep            # graph-timed us/launch over batch x catalog sizes
    python3 tools/emb_roofline.py pmc B V [N] [dist]   # N eager launches of each kernel (run under rocprofv3 --pmc)

Algorithmic bytes follow SURVEY.md 8(d): gather (3L+1)(2*D*4+4) per sequence,
scatter-add (3L+1)(3*D*4+4) per sequence, L = 50, D = 128, fp32.
Ids follow bench.py's synthetic generator shape (Zipf items folded by modulo,
fixed item->category map, positions 0..len-1, len ~ U{2..L}).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mtamrecommender_amd import hip_ops as ops  # noqa: E402

L, D = 50, 128


def make_case(B, V, dist="zipf", seed=1234, n_cat=304, n_user=4835, tables=None):
    """ids from mtamrecommender_amd.data.synthetic.make_id_batch -- the generator bench.py's roofline_at_scale
    legs use, same seeds, so the --pmc passes taken here describe those legs."""
    from mtamrecommender_amd.data.synthetic import make_id_batch
    ids = make_id_batch(B, L, V, n_cat, n_user, dist, seed)
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)
    case = dict(B=B, V=V, item_ids=t(ids["item_list"]), cat_ids=t(ids["category_list"]),
                pos_ids=t(ids["position_list"]), user_ids=t(ids["user_id"]), seq_len=t(ids["seq_length"]))
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev).uniform_(-0.2, 0.2)
    R = B * L
    if tables is None:
        case.update(item=f(V, D), cat=f(n_cat, D), pos=f(L + 3, D), user=f(n_user, D),
                    g_item=torch.zeros(V, D, device=dev), g_cat=torch.zeros(n_cat, D, device=dev),
                    g_pos=torch.zeros(L + 3, D, device=dev), g_user=torch.zeros(n_user, D, device=dev))
    else:       # another id / activation set over the same tables
        case.update({k: tables[k] for k in ("item", "cat", "pos", "user", "g_item", "g_cat", "g_pos", "g_user")})
    case.update(ic=f(R, 2 * D), pos_out=f(R, D), user_out=f(B, D), d_ic=f(R, 2 * D), d_pos=f(R, D),
                l2=torch.zeros(ops.emb_gather_partials(B, L), device=dev),
                sq=torch.zeros(ops.emb_scatter_partials(B, L), device=dev))
    case["live_rows"] = ids["live_rows"]
    return case


def gather(c):
    ops.emb_gather_fwd(c["item"], c["cat"], c["pos"], c["user"], c["item_ids"], c["cat_ids"], c["pos_ids"],
                       c["user_ids"], c["B"], L, 1, c["ic"], c["pos_out"], c["user_out"], c["l2"])


def scatter(c):
    ops.emb_scatter_add_bwd(c["d_ic"], c["d_pos"], c["ic"], c["pos_out"], c["user_out"], c["item_ids"],
                            c["cat_ids"], c["pos_ids"], c["user_ids"], c["seq_len"], c["B"], L, 5e-5, 1,
                            c["g_item"], c["g_cat"], c["g_pos"], c["g_user"], c["sq"])


def graph_time(fns, reps, replays=10):
    """fns: callables doing the same work on different buffer sets; the graph walks them in turn."""
    if callable(fns):
        fns = [fns]
    reps = max(reps, len(fns)) // len(fns) * len(fns)
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fns[i % len(fns)]()
    g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / (reps * replays)


def sweep():
    out = []
    for V in (3709, 1000003, 10000003):
        for B in (128, 512, 2048, 8192):
            for dist in ("zipf", "uniform"):
                # enough distinct (ids, activation buffers) sets that a launch never finds its rows in the
                # 256 MiB Infinity Cache: >= 320 MB of activations in rotation (tables shared)
                per_set = B * L * 3 * D * 4 * 2
                nset = max(1, min(32, (320 << 20) // per_set + 1))
                c = make_case(B, V, dist)
                cases = [c] + [dict(c, **{k: v for k, v in make_case(B, V, dist, seed=1234 + i, tables=c).items()})
                               for i in range(1, nset)]
                reps = 50 if B <= 512 else 10
                tg = graph_time([(lambda cc: (lambda: gather(cc)))(cc) for cc in cases], reps)
                ts = graph_time([(lambda cc: (lambda: scatter(cc)))(cc) for cc in cases], reps)
                gb = (3 * L + 1) * (2 * D * 4 + 4) * B
                sb = (3 * L + 1) * (3 * D * 4 + 4) * B
                row = dict(V=V, B=B, dist=dist, buffer_sets=nset, gather_us=tg * 1e6, gather_GBs=gb / tg / 1e9,
                           gather_frac_8TBs=gb / tg / 8e12, scatter_us=ts * 1e6, scatter_GBs_raw=sb / ts / 1e9,
                           scatter_frac_8TBs_raw=sb / ts / 8e12,
                           scatter_live_GBs=c["live_rows"] * (3 * D * 4 + 4) / ts / 1e9,
                           atomic_added_GBs=c["live_rows"] * D * 4 / ts / 1e9)
                out.append(row)
                print(json.dumps(row), flush=True)
                del c, cases
                torch.cuda.empty_cache()
    return out


def pmc(B, V, n, dist="zipf"):
    """n eager launches of each kernel, every launch on its own id set and buffers (as in the timed legs)."""
    c = make_case(B, V, dist)
    cases = [c] + [make_case(B, V, dist, seed=1234 + i, tables=c) for i in range(1, n)]
    torch.cuda.synchronize()
    for cc in cases:
        gather(cc)
        torch.cuda.synchronize()
    for cc in cases:
        scatter(cc)
        torch.cuda.synchronize()
    print(json.dumps(dict(B=B, V=V, dist=dist, launches=n, live_rows=c["live_rows"],
                          gather_alg_bytes=(3 * L + 1) * (2 * D * 4 + 4) * B,
                          scatter_alg_bytes=(3 * L + 1) * (3 * D * 4 + 4) * B)))


if __name__ == "__main__":
    if sys.argv[1] == "sweep":
        sweep()
    else:
        pmc(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 6,
            sys.argv[5] if len(sys.argv) > 5 else "zipf")
