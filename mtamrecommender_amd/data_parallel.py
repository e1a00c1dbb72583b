"""User-batch data parallelism over the GPUs of one node (RCCL over xGMI).

The reference has no multi-GPU path (run_server.py only fans out
independent experiments).  BASELINE.json's north_star asks for one: every rank
holds a full replica (tables included), takes its own slice of the global
batch, and gradients are summed with RCCL all-reduce before the clip + Adam
update, so that replicas stay bit-identical.

What is exchanged: the dense-parameter gradients AND the
table gradients -- the item table's gradient is dense because of the
full-catalog softmax, so replicas diverge if only the attention/GRU parameters
are reduced.  All gradients live in one flat buffer
[dense | category | position | user | item], so the exchange is ONE all-reduce
per step (one large collective suits the per-link-bound xGMI ring better than
many small ones).
The loss is a mean over the GLOBAL batch (each rank scales by 1/B_global), the
L2 term is a plain sum, so a sum all-reduce is exact.  The clip norm is the
true norm of the summed gradient (the TF IndexedSlices norm of App D-5 is a
single-process artefact and is only reproduced on one GPU).
"""
import torch
import torch.distributed as dist


def shard(records, rank, world):
    """Contiguous, equal slices of a global batch (drop nothing: sizes differ by at most 1)."""
    n = len(records)
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return records[lo:hi]


def allreduce_gradients(buffers, group=None):
    """Sum the given gradient buffers across ranks, in place."""
    for b in buffers:
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)


def attach(path, world_size, group=None, force=False):
    """Make ``path.train_kernels`` exchange gradients; call once after building the model.
    ``force`` installs the exchange even for one rank (tests the code path on one GPU)."""
    if world_size <= 1 and not force:
        return
    path.tf_compat = False
    path.world_size = world_size

    def exchange(p, bt):
        allreduce_gradients([p.flat_g], group)

    path.allreduce_fn = exchange


def broadcast_parameters(path, src=0, group=None):
    """Replicas start identical (they would anyway with equal seeds; this makes it explicit)."""
    dist.broadcast(path.flat_p, src=src, group=group)


def max_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
