"""Data-parallel logic with two gloo ranks on the CPU: sharding, gradient exchange, and the
identity it relies on -- summed shard gradients (loss scaled by the GLOBAL batch) equal the
full-batch gradient, so replicas that all-reduce stay identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from mtamrecommender_amd import data_parallel
    from tests.test_oracle import REG, small_case
    from oracle import mtam_oracle as O
    feed, arrays = small_case("MTAM", B=6, L=8, D=16, NB=1, H=2, seed=9)
    B = len(feed["user_id"])
    rows = data_parallel.shard(list(range(B)), rank, world)
    local = {k: v[rows] for k, v in feed.items()}
    _, grads, _ = O.loss_and_grads("MTAM", arrays, local, 2, 1, REG, torch.float64, global_batch=B)
    names = sorted(k for k, g in grads.items() if g is not None)
    bufs = [torch.from_numpy(np.ascontiguousarray(grads[k])) for k in names]
    data_parallel.allreduce_gradients(bufs)
    if rank == 0:
        _, full, _ = O.loss_and_grads("MTAM", arrays, feed, 2, 1, REG, torch.float64)
        worst = max(float(np.abs(b.numpy() - full[k]).max()) for k, b in zip(names, bufs))
        np.save(os.path.join(out_dir, "worst.npy"), np.array([worst]))
    t = torch.tensor([float(rank + 1)])
    assert data_parallel.max_over_ranks(float(rank + 1), "cpu") == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_partition():
    from mtamrecommender_amd import data_parallel
    recs = list(range(11))
    parts = [data_parallel.shard(recs, r, 4) for r in range(4)]
    assert sum(parts, []) == recs and max(map(len, parts)) - min(map(len, parts)) <= 1


def test_two_rank_gradient_exchange_matches_full_batch(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    worst = float(np.load(os.path.join(str(tmp_path), "worst.npy"))[0])
    assert worst < 1e-12
