#!/usr/bin/env python
"""Headline benchmark: MTAM training step, ml-1m shapes, synthetic data.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full training step (embedding gather -> time-aware GRU ->
time-aware attention -> full-catalog softmax -> backward -> global-norm clip ->
Adam on every variable) on a batch of 128 synthetic ml-1m-shaped sequences per
GPU (seq_len 50, emb 128, fp32).  Inputs are resident in HBM before the timed
region; the region is bracketed by barrier + synchronize, the slowest rank's
time counts, and rank 0 prints one JSON line.  value = sequences/s over all
ranks.  Extra objects on the same line:
  roofline      embedding gather kernel: algorithmic bytes / HIP-event time vs HBM peak
  cpu_baseline  the CPU restatement of the TF1.14 graph (oracle) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3
D = 128
# headline workload (BASELINE.json configs[1]); the flags below select the other configurations
L, NB, H, B_PER_GPU = 50, 1, 1, 128


def gather_bytes_per_seq(L, D, e=4):
    return (3 * L + 1) * (2 * D * e + 4)          # read row + write row + index


def scatter_bytes_per_seq(L, D):
    return (3 * L + 1) * (3 * D * 4 + 4)          # read grad row + RMW table-grad row + index


def time_kernel(fns, torch, reps=50, replays=20):
    """Average duration of one launch: back-to-back launches captured into a hipGraph (so the host launch path
    is out of the picture), replayed `replays` times between two HIP events on the stream the graph runs on.
    Includes the ~1 us dependent-kernel boundary per launch.  `fns`: one callable, or a list of callables that
    do the SAME work on DIFFERENT buffers -- the graph then walks the list (`reps` launches in all), so that with
    more than 256 MiB of distinct buffers no launch finds its inputs or last outputs in the Infinity Cache and the
    figure is an HBM one (MI355X_MICROARCH.md, Infinity Cache)."""
    if callable(fns):
        fns = [fns]
    reps = max(reps, len(fns)) // len(fns) * len(fns)
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    kw = {"capture_error_mode": "thread_local"} if torch.distributed.is_initialized() else {}
    with torch.cuda.graph(g, **kw):
        for i in range(reps):
            fns[i % len(fns)]()
    g.replay()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(replays):
        g.replay()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) * 1e-3 / (reps * replays)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, gfx950 read correction applied: tools/summarize_prof.py pmc).  PMC
    counters cannot be collected inside this process, so the figure comes from the newest
    profiles/r*_pmc_emb_*.json (kernel names there carry their C++ decoration, e.g.
    "void emb_gather_kernel<false>": matched by substring); None when no such file travels with the tree."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_emb_*.json")))
    if not files:
        return None
    try:
        table = json.load(open(files[-1]))
        for name, row in table.items():
            if kernel in name:
                return float(row["traffic_bytes"])
    except (KeyError, ValueError, OSError):
        pass
    return None


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def host_inclusive_rate(model, emb, records, steps, torch):
    """Sequences/s through the drop-in host loop: records -> libmtam_host.so packer (worker thread, one batch
    ahead) -> pinned arena -> H2D copy -> model.train() -> loss read back one step late.  Everything the
    reference's `for batch in DataInput: model.train(sess, batch, lr)` loop does per step
    (train_process.py:326-347).  Never the headline `value`."""
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet
    rs = RecordSet.from_records(records)
    packer = BatchPacker(model.path, emb)
    model.async_loss = True
    done, t0 = 0, None
    while done < steps + 20:
        for _, batch in NativeDataInput(rs, B_PER_GPU, packer, consumer="bench"):
            if len(batch) != B_PER_GPU:
                continue
            if done == 20:                      # 20 untimed steps: graph already captured, pools allocated
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            model.train(model.sess, batch, 1e-3)
            done += 1
            if done >= steps + 20:
                break
    model.last_loss()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.async_loss = False
    return {"value": B_PER_GPU * steps / elapsed, "unit": "sequences/s", "ms_per_step": elapsed / steps * 1e3,
            "steps": steps, "route": "RecordSet -> native packer thread -> pinned arena -> H2D -> model.train(), "
                                     "loss returned one step late (async_loss)"}


def cpu_baseline(records_batches, FLAGS, arrays, budget_s=15.0, model_name="MTAM"):
    """Oracle (torch-CPU fp32, unfused, autograd) on the host cores: sequences/s."""
    import torch
    import oracle.mtam_oracle as O
    from mtamrecommender_amd.Embedding.feed import pad_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads (os.cpu_count() = %s)" % (cores, os.cpu_count()))
    arrays = {k: v.copy() for k, v in arrays.items()}
    state = O.AdamState(arrays)
    feeds = [pad_batch(b, L) for b in records_batches]
    times = []
    t_begin = time.perf_counter()
    i = 0
    while True:
        t0 = time.perf_counter()
        O.train_step(model_name, arrays, state, feeds[i % len(feeds)], 1e-3, H, NB, FLAGS.regulation_rate,
                     FLAGS.max_gradient_norm, True)
        times.append(time.perf_counter() - t0)
        i += 1
        if i >= 5 + 30 or (time.perf_counter() - t_begin > budget_s and i >= 5 + 3):
            break
    timed = times[5:]
    med = float(np.median(timed))
    return {"value": B_PER_GPU / med, "unit": "sequences/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d training steps of batch %d after 5 warm-up steps (median step %.1f ms); "
                      "CPU restatement of the TF1.14 graph (oracle/mtam_oracle.py, torch-CPU fp32)"
                      % (len(timed), B_PER_GPU, med * 1e3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--id-dist", default="zipf", choices=["zipf", "uniform"])
    # non-headline configurations (BASELINE.json configs[2..]); defaults = the headline ml-1m workload
    ap.add_argument("--model", default="MTAM", choices=["MTAM", "PISTRec"])
    ap.add_argument("--items", type=int, default=0, help="catalog size (0: ml-1m's 3706)")
    ap.add_argument("--seq-len", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--heads", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="sequences per GPU")
    ap.add_argument("--score-dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: logits-free bf16-MFMA catalog scoring (BASELINE.json configs[4]); default fp32")
    args = ap.parse_args()
    global L, NB, H, B_PER_GPU
    L, NB, H, B_PER_GPU = args.seq_len, args.blocks, args.heads, args.batch

    # Only the result line may reach stdout: RCCL prints a version banner to stdout when it creates its
    # first communicator, and torch / ROCm libraries may print too.  Everything else goes to stderr.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (see the module docstring)")
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    # MTAM_BENCH_FORCE_DP=1: run the multi-GPU code path (RCCL group, gradient exchange, barriers, max over
    # ranks) with ONE rank -- a rehearsal of the N > 1 launch on a one-GPU box
    force_dp = world == 1 and os.environ.get("MTAM_BENCH_FORCE_DP", "0") == "1"
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    use_dist = world > 1 or force_dp
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device(device))

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()
    from mtamrecommender_amd import data_parallel, hip_ops as ops
    from mtamrecommender_amd.config.model_parameter import model_parameter
    from mtamrecommender_amd.data.synthetic import ML1M, SyntheticCatalog, make_records
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    from mtamrecommender_amd.Model.base_model import Session
    from mtamrecommender_amd.Model.MTAMRec_model import MTAM
    from mtamrecommender_amd.Model.PISTRec_model import Time_Aware_self_Attention_model

    FLAGS = model_parameter().get_parameter("MTAMb1_movielen").FLAGS
    FLAGS.num_blocks, FLAGS.num_heads, FLAGS.length_of_user_history = NB, H, L
    FLAGS.checkpoint_path_dir = "/tmp/mtam_bench_ckpt"
    FLAGS.score_dtype = args.score_dtype
    shape = dict(ML1M)
    if args.items:
        shape.update(item_count=args.items, category_count=max(301, min(1000, args.items // 1000)))
    cat = SyntheticCatalog(seed=1234, **shape)
    emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, L,
                                                  seed=1234)
    model = (MTAM if args.model == "MTAM" else Time_Aware_self_Attention_model)(FLAGS, emb, Session(device))
    p = model.path
    if use_dist:
        data_parallel.attach(p, world, force=force_dp)
        data_parallel.broadcast_parameters(p)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    arrays0 = model.get_variables() if want_cpu else None

    # synthetic records: 32 distinct batches per rank, staged in HBM before the timed region
    n_batches = 32
    records = make_records(cat, n_batches * B_PER_GPU, L, seed=1234 + 977 * rank, id_dist=args.id_dist)
    batches = [records[i * B_PER_GPU:(i + 1) * B_PER_GPU] for i in range(n_batches)]
    feeds = [emb.make_feed_dic_new(b) for b in batches]
    for f in feeds:
        emb.validate_ids(f)
    lr = 1e-3
    staged = [p.stage(f, lr) for f in feeds]
    total = args.warmup + args.steps
    bt = p.batch(B_PER_GPU)

    def step(i):
        bt.arena.copy_(staged[i % n_batches], non_blocking=True)      # device -> device, 128 KB
        model.step_train(bt)

    log("rank %d: model built, %d batches staged" % (rank, n_batches))
    loss_first = None
    for i in range(args.warmup):
        step(i)
        if i == 0:
            loss_first = float(bt.loss[0].item())
    log("rank %d: warm-up done" % rank)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        elapsed = data_parallel.max_over_ranks(elapsed, device)

    log("rank %d: timed region %.3f s for %d steps" % (rank, elapsed, args.steps))
    loss_last = float(bt.loss[0].item())
    if not np.isfinite(loss_last):
        raise SystemExit("non-finite training loss")

    # ---- per-kernel roofline legs (rank 0): back-to-back launches between HIP events
    result = None
    if rank == 0:
        fd, T = bt.feed, p.tables
        # ---- the two embedding kernels through their C entry points, each launch on its own id set and its own
        # output (gather) / input (scatter) buffers: 32 sets x 9.9 MB and 16 x 19.7 MB, more than the 256 MiB
        # Infinity Cache, so written rows go to HBM and gradient rows come from it.  (The ml-1m tables themselves,
        # 2.4 MB, are cache-resident in the real step too.)  "hot" = one buffer set re-used, the round-1 figure.
        R = B_PER_GPU * L
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
        n_rot = max(2, min(32, (288 << 20) // (R * 3 * D * 4) + 1))
        arenas = [st.clone() for st in staged[:n_rot]]
        ids = [{k: bt._view(a, k) for k in ("item_list", "category_list", "position_list", "user_id", "seq_length")}
               for a in arenas]
        outs = [(f32(R, 2 * D), f32(R, D), f32(B_PER_GPU, D)) for _ in range(n_rot)]
        l2p = torch.zeros(ops.emb_gather_partials(B_PER_GPU, L), device=device)

        def gather_fn(i):
            f, (ic, pos, user) = ids[i % len(ids)], outs[i]
            return lambda: ops.emb_gather_fwd(T["item"], T["category"], T["position"], T["user"], f["item_list"],
                                              f["category_list"], f["position_list"], f["user_id"], B_PER_GPU, L, 1,
                                              ic, pos, user, l2p)
        t_gather = time_kernel([gather_fn(i) for i in range(n_rot)], torch, reps=2 * n_rot)
        t_gather_hot = time_kernel(gather_fn(0), torch)
        part = torch.zeros(ops.emb_scatter_partials(B_PER_GPU, L), device=device)
        n_rot_s = max(2, n_rot // 2)
        grads_in = [(f32(R, 2 * D).normal_(0, 1e-3), f32(R, D).normal_(0, 1e-3)) for _ in range(n_rot_s)]

        def scatter_fn(i):
            f, (ic, pos, user), (d_ic, d_x) = ids[i % len(ids)], outs[i], grads_in[i]
            return lambda: ops.emb_scatter_add_bwd(d_ic, d_x, ic, pos, user, f["item_list"], f["category_list"],
                                                   f["position_list"], f["user_id"], f["seq_length"], B_PER_GPU, L,
                                                   p.reg, 1, p.g_tab["item"], p.g_tab["category"], p.g_tab["position"],
                                                   p.g_tab["user"], part)
        t_scatter = time_kernel([scatter_fn(i) for i in range(n_rot_s)], torch, reps=2 * n_rot_s)
        t_scatter_hot = time_kernel(scatter_fn(0), torch)
        del outs, grads_in
        # ---- the form the training step actually runs: the lookups folded into the forward's first GEMM kernel
        # (mtam_seq_chain_gather_fwd: gather + dense4emb + K/V projection + GRU input projection).  MFMA-bound:
        # 2 R (2D D + D n_kv + D n_x) flops on v_mfma_f32_32x32x2_f32; its HBM side is the fused-variant bound of
        # SURVEY.md 8(d), (3L+1)(D e + 4) + L D e per sequence of table rows + ids, plus what it must write.
        t_fused, fused = None, None
        if args.model == "MTAM" and args.score_dtype == "f32" and getattr(bt, "fused_gather", False):
            kvw, kvb = (p.seg("kv/w"), p.seg("kv/b")) if p.cfg["keys"] == "x" and p.cfg["attention"] else (None, None)

            def fused_fn(i):
                f = ids[i % len(ids)]
                return lambda: ops.seq_chain_gather_fwd(
                    T["item"], T["category"], T["position"], T["user"], f["item_list"], f["category_list"],
                    f["position_list"], f["user_id"], B_PER_GPU, L, 1, p.seg("dense4emb/w"), kvw, kvb,
                    p.seg("gru/wx"), p.seg("gru/bx"), bt.ic, bt.user, bt.l2_partial, bt.zr, bt.x,
                    bt.kv if kvw is not None else None, bt.xproj)
            t_fused = time_kernel([fused_fn(i) for i in range(len(ids))], torch, reps=2 * len(ids))
            n_kv = kvw.shape[1] if kvw is not None else 0
            flops = 2.0 * R * (2 * D * D + D * n_kv + D * p.seg("gru/wx").shape[1])
            lookup_bytes = ((3 * L + 1) * (D * 4 + 4) + L * D * 4) * B_PER_GPU
            fused = {"kernel": "seq_chain_fwd_kernel<true>", "bound": "mfma", "achieved": flops / t_fused / 1e12,
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / t_fused / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     "traffic": None, "flops_per_launch": flops, "us_per_launch": t_fused * 1e6,
                     "lookup_bytes_per_launch_fused_bound": lookup_bytes,
                     "note": "the training step's embedding lookups run inside this kernel; the stand-alone gather "
                             "kernel of `roofline` is what mtam_emb_gather_fwd callers (PISTRec, bf16 mode) launch"}
        gb = gather_bytes_per_seq(L, D) * B_PER_GPU
        sb = scatter_bytes_per_seq(L, D) * B_PER_GPU
        log("gather %.2f us (one buffer set re-used: %.2f), scatter-add %.2f us (%.2f) per launch"
            % (t_gather * 1e6, t_gather_hot * 1e6, t_scatter * 1e6, t_scatter_hot * 1e6))
        recall = model.recall_at(model.sess, batches[0], 20)
        result = {
            "metric": "training sequences/sec", "value": B_PER_GPU * world * args.steps / elapsed,
            "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.score_dtype == "f32" else "bf16 scoring operands, f32 accumulate and elsewhere",
            "data": "synthetic",
            "config": {"workload": "%s training step, %s synthetic (%d items, %d categories, %d users), seq_len=%d "
                                   "emb=128 num_blocks=%d num_heads=%d, batch=%d per GPU"
                                   % ("MTAMRec" if args.model == "MTAM" else "PISTRec (Time_Aware_self_Attention_model)",
                                      "ml-1m-shaped" if not args.items else "large-catalog", cat.item_count,
                                      cat.category_count, cat.user_count, L, NB, H, B_PER_GPU),
                       "global_batch": B_PER_GPU * world, "seq_len": L, "parallelism": "dp%d" % world,
                       "id_dist": args.id_dist, "optimizer": "adam", "hipgraph": bool(model.use_graph),
                       "dp_graph": getattr(model, "_dp_mode", None) if use_dist else None},
            "recall_at_20": recall, "loss_first": loss_first, "loss_last": loss_last,
            "roofline": {"kernel": "emb_gather_kernel", "bound": "hbm", "achieved": gb / t_gather / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / t_gather / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("emb_gather_kernel") if (L, B_PER_GPU) == (50, 128) else None,
                         "bytes_per_launch": gb, "us_per_launch": t_gather * 1e6,
                         "buffer_sets": n_rot, "us_per_launch_cache_hot": t_gather_hot * 1e6,
                         "note": "latency-bound at 128 sequences per launch (a 1.7 us empty kernel in the same "
                                 "harness); HBM-bound sizes: profiles/r02_emb_sweep_*.jsonl"},
            "roofline_scatter_add": {"kernel": "emb_scatter_kernel", "bound": "hbm", "achieved": sb / t_scatter / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": sb / t_scatter / 1e9 / HBM_PEAK_GBS,
                                     "traffic": pmc_traffic("emb_scatter_kernel")
                                     if (L, B_PER_GPU) == (50, 128) else None,
                                     "bytes_per_launch": sb, "us_per_launch": t_scatter * 1e6,
                                     "buffer_sets": n_rot_s, "us_per_launch_cache_hot": t_scatter_hot * 1e6},
        }
        if fused is not None:
            result["roofline_fused_forward"] = fused
            log("fused lookups + projections: %.2f us per launch (%.1f TFLOP/s)" % (t_fused * 1e6, fused["achieved"]))
        if args.score_dtype == "bf16":
            # the two catalog passes of the bf16 scoring (csrc/score16.hip): algorithmic bytes per catalog row =
            # 256 (bf16 row, lse pass) and 256 + 512 (bf16 row read, fp32 gradient row written; backward)
            V, reps = p.item_rows, (50 if p.item_rows < 2000000 else 3)
            tgt = fd["target_item_id"]
            t_lse = time_kernel(lambda: ops.score16_lse(p.item16, bt.pred16, tgt, B_PER_GPU, V, bt.s16_partial, bt.lse,
                                                        bt.ce), torch, reps=reps, replays=5)
            sq = bt.norm_partial[p.nb_dense:]
            t_bwd = time_kernel(lambda: ops.score16_bwd(p.item16, bt.pred16, bt.lse, tgt, B_PER_GPU, V,
                                                        1.0 / (B_PER_GPU * world), bt.d_pred, p.g_tab["item"], sq),
                                torch, reps=reps, replays=5)
            for key, kernel, t, nbytes in (("roofline_score16_lse", "score16_lse_kernel", t_lse, V * 256.0),
                                           ("roofline_score16_bwd", "score16_bwd_kernel", t_bwd, V * 768.0)):
                result[key] = {"kernel": kernel, "bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "bytes_per_launch": nbytes, "us_per_launch": t * 1e6}
            log("score16 lse %.1f us, backward %.1f us per launch" % (t_lse * 1e6, t_bwd * 1e6))
        if world == 1:
            result["host_inclusive"] = host_inclusive_rate(model, emb, records, min(args.steps, 300), torch)
            log("host-inclusive: %.0f sequences/s (%.3f ms per step)" % (result["host_inclusive"]["value"],
                                                                       result["host_inclusive"]["ms_per_step"]))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(batches[:8], FLAGS, arrays0, model_name=args.model)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
