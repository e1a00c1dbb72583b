#!/usr/bin/env python
"""Headline benchmark: MTAM training step, ml-1m shapes, synthetic data.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full training step (embedding gather -> time-aware GRU ->
time-aware attention -> full-catalog softmax -> backward -> global-norm clip ->
Adam on every variable) on a batch of 128 synthetic ml-1m-shaped sequences per
GPU (seq_len 50, emb 128, fp32).  Inputs are resident in HBM before the timed
region; the region is bracketed by barrier + synchronize, the slowest rank's
time counts, and rank 0 prints one JSON line.  value = sequences/s over all
ranks.  The staged feeds form a ring in HBM and the optimizer launch of step k
hands step k + 1 its feed (no copy in front of a step).  Ahead of the W warm-up
steps every rank runs a device pre-roll -- ~60 ms of the model's forward-only
evaluation pass, nothing of the training state touched -- because a GPU that
idled through model build needs ~10 ms of load to get its clocks back and a
short run (W = 5, K = 20) would otherwise sit on that ramp; W and K are exactly
the training steps run and timed.

Launching.  With --gpus N > 1 and no WORLD_SIZE in the environment this file is
its own launcher: the parent starts N children of itself (one process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) BEFORE
anything touches a GPU -- it imports neither torch nor the package -- relays
rank 0's JSON line to stdout and exits with the worst child's return code.
Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(WORLD_SIZE set) it is one rank.  The reference's only launcher starts one
process per GPU the same way (run_server.py:46-99).
`--launch-selftest` exercises launcher + rendezvous + line relay with gloo ranks
on a stub step (no GPU): tests/test_bench_launcher.py.

Extra objects on the result line:
  roofline            embedding gather kernel at the step's batch: algorithmic bytes / HIP-event time vs HBM peak
  roofline_at_scale   the same gather and scatter-add kernels at 512 and 2,048 sequences per launch
  gru_serial_model    the serial GRU pair against a dependent-chain model (not a roofline)
  cpu_baseline        the CPU restatement of the TF1.14 graph (oracle) timed on the host cores
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
    """HBM-side bytes per launch of `kernel` at the headline size (128 sequences, ml-1m tables) from the committed
    rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 read correction applied:
    tools/summarize_prof.py).  PMC counters cannot be collected inside this process, so the figure comes from the
    newest profiles/r*_pmc_emb_scale*.json (case "B128_V3709_zipf"); None when no such file travels with the tree."""
    return pmc_scale_traffic("B128_V3709_zipf", kernel.split("_")[1])


SCALE_LEGS = [(512, "run", "zipf"), (2048, "run", "zipf"), (512, "1m", "zipf"), (2048, "1m", "zipf"),
              (512, "1m", "uniform"), (2048, "1m", "uniform")]
ROT_BYTES = 768 << 20          # distinct streamed bytes in rotation per leg: 3 x the 256 MiB Infinity Cache


def pmc_scale_traffic(key, kernel):
    """Per-launch HBM bytes of one roofline_at_scale leg from the committed PMC passes
    (profiles/r*_pmc_emb_scale*.json: {"B512_V3709_zipf": {"gather": {..."traffic_bytes"}, "scatter": ...}})."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_emb_scale*.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))[key][kernel]["traffic_bytes"])
    except (KeyError, ValueError, OSError, TypeError):
        return None


def pmc_step_traffic(kernel):
    """Per-launch HBM-side bytes of one of the training step's kernels from the committed PMC passes over
    `bench.py` itself (profiles/r*_pmc_step_*.json, tools/gpu_r3_pmc_step.sh); None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step_*.json")))
    if not files:
        return None
    try:
        for name, row in json.load(open(files[-1])).items():
            if kernel in name:
                return float(row["traffic_bytes"])
    except (KeyError, ValueError, OSError, TypeError):
        pass
    return None


def emb_scale_legs(ops, torch, device, run_tables, reg):
    """`roofline_at_scale`: the stand-alone gather and scatter-add kernels (through their C entry points) at 512 and
    2,048 sequences per launch, L = 50, D = 128 -- the sizes at which the launch is bandwidth-bound rather than a
    latency chain -- over the run's own tables (ml-1m sized: table reads are L2 hits, as in the real step) and
    over a 1,000,003-row item table (512 MB; with uniform ids the row reads are HBM reads too).  Every launch of
    the timed graph works on its own id set and its own streamed buffers, ROT_BYTES of them in rotation.
    Algorithmic bytes per SURVEY.md 8(d); `traffic` from the committed --pmc passes of the same cases
    (tools/emb_roofline.py pmc)."""
    from mtamrecommender_amd.data.synthetic import make_id_batch
    Ls = 50
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=device)
    big = None
    out = []
    for B, which, dist in SCALE_LEGS:
        if which == "run":
            T = run_tables
        else:
            if big is None:
                big = dict(run_tables, item=f32(1000003, D).uniform_(-0.2, 0.2))
            T = big
        V = T["item"].shape[0]
        G = {k: torch.zeros_like(v) for k, v in T.items()} if which == "run" else \
            dict({k: torch.zeros_like(v) for k, v in run_tables.items() if k != "item"}, item=torch.zeros_like(T["item"]))
        R = B * Ls
        n_g = max(2, ROT_BYTES // (R * 3 * D * 4) + 1)           # gather streams its outputs
        n_s = max(2, ROT_BYTES // (R * 3 * D * 4 * 2) + 1)       # scatter-add its gradient rows AND the looked-up rows
        nset = max(n_g, n_s)
        ids = []
        for i in range(nset):
            c = make_id_batch(B, Ls, V, T["category"].shape[0], T["user"].shape[0], dist, seed=1234 + i)
            ids.append({k: torch.from_numpy(v).to(device) for k, v in c.items() if k != "live_rows"})
        outs = [(f32(R, 2 * D), f32(R, D), f32(B, D)) for _ in range(nset)]
        l2p = torch.zeros(ops.emb_gather_partials(B, Ls), device=device)

        def gather_fn(i):
            f, (ic, pos, user) = ids[i], outs[i]
            return lambda: ops.emb_gather_fwd(T["item"], T["category"], T["position"], T["user"], f["item_list"],
                                              f["category_list"], f["position_list"], f["user_id"], B, Ls, 1,
                                              ic, pos, user, l2p)
        reps = 40 if B <= 512 else 12
        t_g = time_kernel([gather_fn(i) for i in range(n_g)], torch, reps=reps, replays=10)
        grads_in = [(f32(R, 2 * D).normal_(0, 1e-3), f32(R, D).normal_(0, 1e-3)) for _ in range(n_s)]
        part = torch.zeros(ops.emb_scatter_partials(B, Ls), device=device)

        def scatter_fn(i):
            f, (ic, pos, user), (d_ic, d_x) = ids[i], outs[i], grads_in[i]
            return lambda: ops.emb_scatter_add_bwd(d_ic, d_x, ic, pos, user, f["item_list"], f["category_list"],
                                                   f["position_list"], f["user_id"], f["seq_length"], B, Ls, reg, 1,
                                                   G["item"], G["category"], G["position"], G["user"], part)
        t_s = time_kernel([scatter_fn(i) for i in range(n_s)], torch, reps=reps, replays=10)
        gb, sb = gather_bytes_per_seq(Ls, D) * B, scatter_bytes_per_seq(Ls, D) * B
        key = "B%d_V%d_%s" % (B, V, dist)
        for kernel, t, nb, n in (("emb_gather_kernel", t_g, gb, n_g), ("emb_scatter_kernel", t_s, sb, n_s)):
            out.append({"kernel": kernel, "sequences_per_launch": B, "seq_len": Ls, "item_rows": V, "id_dist": dist,
                        "bound": "hbm", "achieved": nb / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": nb / t / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": nb, "us_per_launch": t * 1e6,
                        "buffer_sets": n, "traffic": pmc_scale_traffic(key, kernel.split("_")[1])})
        log("at scale %s: gather %.1f us (%.2f of peak), scatter-add %.1f us (%.2f)"
            % (key, t_g * 1e6, gb / t_g / 8e12, t_s * 1e6, sb / t_s / 8e12))
        del ids, outs, grads_in, G
        torch.cuda.empty_cache()
    return out


def gru_serial_model(L_, ms_fwd, ms_bwd, clock_ghz=2.3):
    """The time-aware GRU against a dependent-chain model (latency-bound, not a roofline).
    One workgroup per sample walks L-1 steps; a step is one wave's chain of DEPENDENT instructions:
      gate contraction   32 v_pk_fma_f32 per wave at 8.5 cycles issue, two waves sharing a SIMD   (~770 cycles measured)
      octet reduction    3 DPP adds + sigmoid (exp, rcp) at 15-17 cycles per dependent VALU op     (~300)
      candidate          r.h through LDS, 16 v_pk_fma_f32, reduction, tanh                          (~410)
      update + barriers  u h + (1-u) c T, two s_barrier                                             (~490)
    Model cycles per step = the sum of the per-phase FLOORS (issue intervals from profiles/r02_valu_issue_lab.txt):
    32 x 8.5 x 2 + 7 x 16 + (16 x 8.5 x 2 + 7 x 16) + 2 x 64 + 10 x 16 = 1,328; achieved = kernel time / (L-1)
    at the clock the stamps were taken at."""
    steps = L_ - 1
    model = 32 * 8.5 * 2 + 7 * 16 + (16 * 8.5 * 2 + 7 * 16) + 2 * 64 + 10 * 16
    to_cycles = lambda ms: ms * 1e-3 * clock_ghz * 1e9 / steps
    return {"kernel": "tagru_fwd_kernel + tagru_bwd_kernel", "bound": "serial dependent chain (not a roofline)",
            "steps_per_launch": steps, "clock_ghz_assumed": clock_ghz, "model_cycles_per_step": model,
            "achieved_cycles_per_step_fwd": to_cycles(ms_fwd), "achieved_cycles_per_step_bwd": to_cycles(ms_bwd),
            "frac_fwd": model / to_cycles(ms_fwd), "us_fwd": ms_fwd * 1e3, "us_bwd": ms_bwd * 1e3,
            "stamps": "profiles/r02_gru_lab_stamps_v4_final.txt"}


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
            "steps": steps, "route": "RecordSet -> native packer thread -> pinned arena -> model.train(): ONE hipGraph "
                                     "launch per step (the arena's H2D copy is its first node, the loss's D2H copy its "
                                     "last; loss returned one step late, async_loss)"}


def host_inclusive_resident_rate(model, emb, records, steps, torch):
    """The same host loop over a DEVICE-RESIDENT epoch (FLAGS.resident_epoch; base_model.load_resident_epoch): the
    epoch's batches are packed and copied to HBM once -- inside the timed region -- and model.train() is then one graph
    launch per step with no feed copy; the loss still comes back every step, one step late."""
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, RecordSet
    rs = RecordSet.from_records(records)
    packer = BatchPacker(model.path, emb)
    model.async_loss = True
    n = len(records) // B_PER_GPU
    order = np.arange(n * B_PER_GPU, dtype=np.int64)
    done, t0, epochs = 0, None, 0
    nxt = model.prepare_resident_epoch(rs, order, B_PER_GPU, [1e-3] * n, packer)
    while done < steps + 2 * n:
        if epochs == 2:                         # two untimed epochs: ring allocated, graphs captured
            torch.cuda.synchronize()
            t0, done = time.perf_counter(), 2 * n
        handles = model.load_resident_epoch(prepared=nxt)
        nxt = model.prepare_resident_epoch(rs, order, B_PER_GPU, [1e-3] * n, packer)    # the next epoch, on a thread
        for h in handles:
            model.train(model.sess, h, 1e-3)
            done += 1
        epochs += 1
    model.last_loss()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    nxt.thread.join()
    timed = done - 2 * n
    model.async_loss = False
    model.path.batch(B_PER_GPU).feed_ring = None
    return {"value": B_PER_GPU * timed / elapsed, "unit": "sequences/s", "ms_per_step": elapsed / timed * 1e3,
            "steps": timed, "epochs": epochs - 2,
            "route": "RecordSet -> native packer (epoch e + 1 packed on a worker thread while epoch e trains) -> pinned "
                     "staging buffer -> H2D copy of the epoch's %d batches on a side stream (timed) -> "
                     "model.train(ResidentBatch): one hipGraph launch per step, the optimizer launch hands "
                     "the next step its feed; loss D2H copy as the graph's last node, returned one step late" % n}


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


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(n, argv):
    """Parent of an N-rank run: start N children of this file, one per GPU, and relay rank 0's result line.
    Nothing here may touch a GPU (no torch import, no package import): a process that has initialised HIP must
    not fork/exec workers on this pool.  Children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (what torch.distributed.run would set; the reference's launcher is likewise one process per GPU,
    run_server.py:46-99).  Rank 0's stdout is piped and its last JSON line re-printed; every other stream goes to
    this process's stderr.  If a child fails, the others are terminated by PID (they would otherwise wait in a
    collective until the RCCL timeout) and the worst return code is this process's."""
    import signal
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   MTAM_BENCH_CHILD="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                         stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    log("launcher: %d ranks started (pids %s), rendezvous 127.0.0.1:%d" % (n, [c.pid for c in children], port))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(children[0].stdout.read().decode().splitlines()))
    reader.start()
    rcs, failed, first_rc = [None] * n, False, 0
    while any(rc is None for rc in rcs):
        for i, c in enumerate(children):
            if rcs[i] is None:
                rcs[i] = c.poll()
                if rcs[i] not in (None, 0) and not failed:
                    failed, first_rc = True, abs(rcs[i])
                    log("launcher: rank %d exited with %d: stopping the other ranks" % (i, rcs[i]))
                    for j, o in enumerate(children):
                        if rcs[j] is None and o.poll() is None:
                            o.send_signal(signal.SIGTERM)
        time.sleep(0.05)
    reader.join()
    result = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    worst = first_rc if failed else 0             # (the ranks stopped by SIGTERM report -15: not the cause)
    if worst == 0 and len(result) != 1:
        log("launcher: expected one result line from rank 0, got %d" % len(result))
        worst = 1
    if worst == 0:
        sys.stdout.write(result[-1] + "\n")
        sys.stdout.flush()
    return worst


def selftest_rank(args):
    """--launch-selftest: one rank of a stub run -- gloo on the CPU, a sleep for a step -- through the same
    barrier / max-over-ranks / one-line protocol as the real thing (the launcher's CPU test)."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if os.environ.get("MTAM_SELFTEST_FAIL_RANK") == str(rank):
        raise SystemExit(3)                      # a rank that dies before the rendezvous: the launcher must not hang
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    print("rank %d chatter on stdout (must not reach the result line)" % rank)
    for _ in range(args.warmup):
        time.sleep(1e-3)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3 * (1 + rank))            # the slowest rank's time counts
    dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per_rank, torch.tensor([elapsed], dtype=torch.float64))
    seen = torch.ones(1)
    dist.all_reduce(seen)
    worst = max(float(t) for t in per_rank)
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest (stub step)", "value": B_PER_GPU * world * args.steps / worst,
                          "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": worst / args.steps * 1e3, "ranks_seen": int(seen.item()),
                          "ms_per_step_by_rank": [float(t) / args.steps * 1e3 for t in per_rank]}))
    dist.barrier()
    dist.destroy_process_group()


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
    ap.add_argument("--dp-exchange", default=None, choices=["flat", "sharded", "sharded-scoring", "sharded-table"],
                    help="data-parallel gradient exchange (default: by catalog size -- flat below 64 MiB of item "
                         "table, sharded above; sharded-scoring: the item table row-sharded for scoring too)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="stub ranks over gloo, no GPU: checks launcher, rendezvous and the one-line protocol")
    ap.add_argument("--no-scale-legs", action="store_true", help="skip roofline_at_scale (B = 512 / 2,048 legs)")
    args = ap.parse_args()
    global L, NB, H, B_PER_GPU
    L, NB, H, B_PER_GPU = args.seq_len, args.blocks, args.heads, args.batch
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args.gpus, sys.argv[1:]))          # parent: no GPU call before or after this line
    if args.launch_selftest:
        return selftest_rank(args)

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
        raise SystemExit("WORLD_SIZE=%d in the environment but --gpus %d" % (world, args.gpus))
    # MTAM_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box): every rank on cuda:0, collectives over gloo -- RCCL refuses
    # two ranks on one device.  Exercises launcher, rank code, exchanges and kernels with world > 1; not a measurement.
    share_gpu = os.environ.get("MTAM_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
        # the one-launch scoring kernel waits inside the launch for all of its workgroups to be resident: grids of
        # several processes on ONE device could wait on each other's CUs (csrc/score32.hip, small_form)
        os.environ["MTAM_SCORE32_FUSED"] = "0"
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
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
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
        data_parallel.attach(p, world, force=force_dp, exchange=args.dp_exchange)
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

    # Adam on one GPU or under the flat data-parallel exchange: the 32 staged feeds form a ring in HBM and the optimizer
    # launch of every step copies the NEXT feed into the arena (Model/time_aware_path.py FeedRing,
    # mtam_adam_images_clip_feed) -- on one GPU a step is one graph launch with nothing in front of it.
    # MTAM_BENCH_FEED_RING=0 (and the row-sharded exchanges, which update through their own launches): a
    # device -> device copy of the arena ahead of each step.
    use_ring = os.environ.get("MTAM_BENCH_FEED_RING", "1") != "0" and p.ring_supported(bt)
    ring = None
    if use_ring:
        ring = p.feed_ring(bt, n_batches)
        for j, st in enumerate(staged):
            ring.put(j, st)
        ring.prime(0)

    def step(i):
        if ring is None:
            bt.arena.copy_(staged[i % n_batches], non_blocking=True)      # device -> device, 128 KB
        model.step_train(bt)

    log("rank %d: model built, %d batches staged" % (rank, n_batches))
    # ---- device pre-roll.  The GPU has idled through model build and staging (seconds of host work) and its clocks
    # take ~10 ms of load to come back: measured with an event between the steps, the first 20 steps after the idle
    # phase run at 0.2325 ms, the next 20 at 0.229, everything from the 60th on at 0.2256 -- so W = 5
    # warm-up steps + K = 20 timed ones, the driver's command, would sit entirely on the ramp.  Before the W warm-up
    # steps every rank therefore replays the model's FORWARD-ONLY evaluation pass (scores + top-k of staged batch 0:
    # no parameter, no optimizer state, no feed-ring slot changes) for MTAM_BENCH_PREROLL_MS of device time (default
    # 60; 0 = none).  The warm-up and timed steps follow with no idle gap; they are exactly W and K training steps.
    preroll_ms, preroll_n = float(os.environ.get("MTAM_BENCH_PREROLL_MS", "60")), 0
    if preroll_ms > 0 and getattr(p, "sharded_scoring", None) is None:     # (those exchanges fetch rows collectively)
        bt.arena.copy_(staged[0])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(3):                                       # (eager pass, capture, first replay)
            model._run("eval", bt, p.eval_kernels)
        ev[0].record()
        for _ in range(4):
            model._run("eval", bt, p.eval_kernels)
        ev[1].record()
        torch.cuda.synchronize()
        per_pass = max(ev[0].elapsed_time(ev[1]) / 4.0, 1e-3)
        preroll_n = int(min(5000, max(1, preroll_ms / per_pass)))
        for _ in range(preroll_n):
            model._run("eval", bt, p.eval_kernels)
        if ring is not None:
            ring.prime(0)                                         # (the pass used the arena: slot 0 goes back in)
        log("rank %d: pre-roll of %d forward-only passes (%.3f ms each)" % (rank, preroll_n, per_pass))
    loss_first = None
    for i in range(args.warmup):
        step(i)
        if i == 0:
            loss_first = float(bt.loss[0].item())
    log("rank %d: warm-up done" % rank)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # MTAM_BENCH_STEP_EVENTS=1 (diagnostic, changes the timed region: an event between the steps): per-step device times
    step_events = ([torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
                   if os.environ.get("MTAM_BENCH_STEP_EVENTS", "0") == "1" else None)
    t0 = time.perf_counter()
    if step_events is not None:
        step_events[0].record()
    for i in range(args.warmup, total):
        step(i)
        if step_events is not None:
            step_events[i - args.warmup + 1].record()
    t_enqueued = time.perf_counter()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen, per_rank_ms = 1, [elapsed / args.steps * 1e3]
    if use_dist:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = torch.zeros(max(world, 1), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, mine)
        per_rank_ms = [float(t) / args.steps * 1e3 for t in every.cpu()]
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        elapsed = data_parallel.max_over_ranks(elapsed, device)

    log("rank %d: timed region %.3f s for %d steps (all steps enqueued after %.3f ms)"
        % (rank, elapsed, args.steps, (t_enqueued - t0) * 1e3))
    if step_events is not None:
        log("rank %d: per-step device ms: %s" % (rank, " ".join(
            "%.4f" % step_events[j].elapsed_time(step_events[j + 1]) for j in range(args.steps))))
    if use_dist:
        model._current_table()      # ("sharded-table": rank 0 evaluates below; bringing a replica up to date is a collective)
    loss_last = float(bt.loss[0].item())
    if not np.isfinite(loss_last):
        raise SystemExit("non-finite training loss")
    if ring is not None:
        if int(ring.cursor.item()) != total + 1:
            raise SystemExit("feed ring: cursor %d after %d steps" % (int(ring.cursor.item()), total))
        bt.feed_ring = None          # the legs below feed the arena themselves

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
            # (the K/V projection is part of this launch only when it does not ride with the GRU launch)
            kv_here = p.cfg["keys"] == "x" and p.cfg["attention"] and not p._kv_role_on()
            kvw, kvb = (p.seg("kv/w"), p.seg("kv/b")) if kv_here else (None, None)

            def fused_fn(i):
                f = ids[i % len(ids)]
                return lambda: ops.seq_chain_gather_fwd(
                    T["item"], T["category"], T["position"], T["user"], f["item_list"], f["category_list"],
                    f["position_list"], f["user_id"], B_PER_GPU, L, 1, p.seg("dense4emb/w"), kvw, kvb,
                    p.seg("gru/wx"), p.seg("gru/bx"), bt.ic, bt.user, bt.l2_partial, bt.zr, bt.x,
                    bt.kv if kvw is not None else None, bt.xproj, w_images=p.wimg)
            t_fused = time_kernel([fused_fn(i) for i in range(len(ids))], torch, reps=2 * len(ids))
            n_kv = kvw.shape[1] if kvw is not None else 0
            flops = 2.0 * R * (2 * D * D + D * n_kv + D * p.seg("gru/wx").shape[1])
            lookup_bytes = ((3 * L + 1) * (D * 4 + 4) + L * D * 4) * B_PER_GPU
            x3 = p.wimg is not None
            # split-bf16 form: six bf16-MFMA terms per fp32 product -> priced against the bf16 peak at 6 x the flops
            peak = 2500.0 if x3 else MFMA_F32_PEAK_TFLOPS
            issued = flops * (6 if x3 else 1)
            fused = {"kernel": "seq_chain_x3_kernel<true>" if x3 else "seq_chain_fwd_kernel<true>", "bound": "mfma",
                     "achieved": issued / t_fused / 1e12, "peak": peak, "unit": "TFLOP/s",
                     "frac": issued / t_fused / 1e12 / peak, "fp32_equivalent_tflops": flops / t_fused / 1e12,
                     "arith": "6 bf16-MFMA terms per fp32 product (split operands)" if x3 else "fp32 MFMA",
                     "traffic": pmc_step_traffic("seq_chain_x3_kernel") if x3 and (L, B_PER_GPU) == (50, 128) else None,
                     "flops_per_launch": flops, "us_per_launch": t_fused * 1e6,
                     "lookup_bytes_per_launch_fused_bound": lookup_bytes,
                     "what_bounds_it": "not the matrix pipe: every workgroup pulls all of the weights' images from L2 "
                                       "(~70 GB/s per CU) beside 22.9 MB of stores and the gather (DESIGN.md 5.0)",
                     "kv_projection": "inside this launch" if kv_here else
                                      "extra workgroups of the GRU launch (the CUs the recurrence leaves idle)",
                     "note": "the training step's embedding lookups run inside this kernel; the stand-alone gather "
                             "kernel of `roofline` is what mtam_emb_gather_fwd callers (PISTRec, bf16 mode) launch"}
        gb = gather_bytes_per_seq(L, D) * B_PER_GPU
        sb = scatter_bytes_per_seq(L, D) * B_PER_GPU
        log("gather %.2f us (one buffer set re-used: %.2f), scatter-add %.2f us (%.2f) per launch"
            % (t_gather * 1e6, t_gather_hot * 1e6, t_scatter * 1e6, t_scatter_hot * 1e6))
        recall = model.recall_at(model.sess, batches[0], 20)
        gru_model = None
        if args.model == "MTAM" and p.cfg["gru"] == "time":
            tvec = p.seg("gru/tvec")
            # (the recurrence alone, weights from their register-order image as in the step; in the step the launch also
            # carries the K/V projection on the CUs it leaves idle)
            t_gf = time_kernel(lambda: ops.tagru_fwd(bt.xproj, bt.x, fd["timelast_list"], fd["seq_length"],
                                                     p.seg("gru/wh_g"), p.seg("gru/wh_c"), tvec, B_PER_GPU, L, bt.hs,
                                                     bt.short, bt.gru_save, w_image=p.gru_img), torch, reps=20,
                               replays=10)
            t_gb = time_kernel(lambda: ops.tagru_bwd(bt.d_dec[0], bt.x, fd["timelast_list"], fd["seq_length"],
                                                     p.seg("gru/wh_g"), p.seg("gru/wh_c"), tvec, bt.gru_save,
                                                     B_PER_GPU, L, bt.d_xproj, bt.rh, bt.d_xt, bt.d_tvec_partial),
                               torch, reps=20, replays=10)
            gru_model = gru_serial_model(L, t_gf * 1e3, t_gb * 1e3)
            log("GRU forward %.1f us, backward %.1f us per launch" % (t_gf * 1e6, t_gb * 1e6))
        result = {
            "metric": "training sequences/sec", "value": B_PER_GPU * world * args.steps / elapsed,
            "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.score_dtype == "f32" else "bf16 scoring operands, f32 accumulate and elsewhere",
            "arith": ("training GEMMs and catalog scoring: fp32 via 3 x bf16 split operands (6 bf16-MFMA terms per "
                      "product, each exact), fp32 accumulate -- fp32-equivalent (tested to fp64 within 2x a native "
                      "fp32 matmul's error), the fused forward projections, the backward stripe kernel and the K/V "
                      "riders of the GRU launches included (weights pre-split into bf16 operand images by the Adam "
                      "launch); GRU / attention / optimizer / lookups: fp32 VALU; evaluation scores: k-ordered fp32 "
                      "fmaf chain"),
            "data": "synthetic", "ranks_seen": ranks_seen, "ms_per_step_by_rank": per_rank_ms,
            "rehearsal": ("every rank on cuda:0, gloo collectives (MTAM_BENCH_SHARE_GPU=1): NOT a measurement"
                          if share_gpu else None),
            "exchange": (None if not use_dist else
                         {"flat": "flat all-reduce of every gradient",
                          "sharded": "row-sharded item exchange (reduce-scatter + owned Adam + all-gather)",
                          "sharded-scoring": "item table row-sharded for scoring too (all-gather pred, reduced lse and "
                                             "d_pred, dE born sharded, slot exchange, owned Adam, all-gather)",
                          "sharded-table": "item table row-sharded for scoring AND left sharded (history rows fetched "
                                           "from their owners, no all-gather of the updated rows)"}
                         [p.dp_exchange]),
            "config": {"workload": "%s training step, %s synthetic (%d items, %d categories, %d users), seq_len=%d "
                                   "emb=128 num_blocks=%d num_heads=%d, batch=%d per GPU"
                                   % ("MTAMRec" if args.model == "MTAM" else "PISTRec (Time_Aware_self_Attention_model)",
                                      "ml-1m-shaped" if not args.items else "large-catalog", cat.item_count,
                                      cat.category_count, cat.user_count, L, NB, H, B_PER_GPU),
                       "global_batch": B_PER_GPU * world, "seq_len": L, "parallelism": "dp%d" % world,
                       "id_dist": args.id_dist, "optimizer": "adam", "hipgraph": bool(model.use_graph),
                       "preroll": ("%d forward-only evaluation passes on staged batch 0 (~%.0f ms of device time, no "
                                   "model or optimizer state touched) ahead of the warm-up steps: the clocks of a GPU "
                                   "that idled through model build take ~10 ms of load to come back"
                                   % (preroll_n, preroll_ms) if preroll_n else None),
                       "feed": ("ring of %d packed feeds resident in HBM; the optimizer launch of step k copies feed "
                                "k + 1 into the arena (%s)"
                                % (n_batches, "the update graph behind the all-reduce carries it" if use_dist else
                                   "one graph launch per step, nothing in front of it")
                                if ring is not None else
                                "%d packed feeds resident in HBM; a device-to-device copy of the arena ahead of each "
                                "step" % n_batches),
                       "dp_graph": getattr(model, "_dp_mode", None) if use_dist else None},
            "recall_at_20_train_batch_smoke": recall,
            "recall_note": "Recall@20 of TRAINING batch 0 after the timed steps on synthetic records: a smoke value "
                           "(the model fits its 32 batches), not BASELINE's ml-1m Recall@20 -- the dataset is not here",
            "loss_first": loss_first, "loss_last": loss_last,
            "roofline": {"kernel": "emb_gather_kernel", "bound": "hbm", "achieved": gb / t_gather / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / t_gather / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("emb_gather_kernel") if (L, B_PER_GPU) == (50, 128) else None,
                         "bytes_per_launch": gb, "us_per_launch": t_gather * 1e6,
                         "buffer_sets": n_rot, "us_per_launch_cache_hot": t_gather_hot * 1e6,
                         "note": "latency-bound at 128 sequences per launch (a 1.7 us empty kernel in the same "
                                 "harness); the bandwidth-bound sizes are in roofline_at_scale"},
            "roofline_scatter_add": {"kernel": "emb_scatter_kernel", "bound": "hbm", "achieved": sb / t_scatter / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": sb / t_scatter / 1e9 / HBM_PEAK_GBS,
                                     "traffic": pmc_traffic("emb_scatter_kernel")
                                     if (L, B_PER_GPU) == (50, 128) else None,
                                     "bytes_per_launch": sb, "us_per_launch": t_scatter * 1e6,
                                     "buffer_sets": n_rot_s, "us_per_launch_cache_hot": t_scatter_hot * 1e6},
        }
        if gru_model is not None:
            result["gru_serial_model"] = gru_model
        if not args.no_scale_legs and args.score_dtype == "f32" and world == 1:      # (the other ranks would only wait)
            result["roofline_at_scale"] = emb_scale_legs(ops, torch, device, T, p.reg)
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
            if use_ring and not use_dist:
                result["host_inclusive_resident_epoch"] = host_inclusive_resident_rate(model, emb, records,
                                                                                       min(args.steps, 300), torch)
                log("host-inclusive, resident epochs: %.0f sequences/s (%.3f ms per step)"
                    % (result["host_inclusive_resident_epoch"]["value"],
                       result["host_inclusive_resident_epoch"]["ms_per_step"]))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(batches[:8], FLAGS, arrays0, model_name=args.model)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
