"""Data parallelism over clips: one process per GPU, gradients summed with an RCCL all-reduce
(torch.distributed backend "nccl" IS RCCL on ROCm) over xGMI.

The reference is single process (run_task.py:132); this is what the build adds.
Clips are independent units: rank r takes clips [r*B/W, (r+1)*B/W) of each global batch.  Every
rank scales its loss gradient by 1/(local_rows * world) (the loss is a batch mean, train.py:123),
so a SUM all-reduce of the flat gradient buffer yields the global-batch gradient on every rank;
global-norm clipping and SGD then run identically everywhere (train.py:215-217).

Two buckets, in the order backward produces them: [classifier + fc6/7/8] (85 % of the bytes) is
reduced while conv5..conv1 backward (94 % of the FLOPs) still runs; [conv] follows at the end.
torch.distributed launches each all-reduce on RCCL's own stream behind the work already queued on
the compute stream, and wait() makes the compute stream wait for it -- no host synchronisation."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract).
    Returns (rank, world, local_rank); world == 1 means no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # VLTF_DIST_BACKEND=gloo: rehearse the multi-rank path where RCCL cannot run (several ranks on one GPU)
            backend = os.environ.get("VLTF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total, rank, world):
    """Contiguous clip range of `rank`: the first total % world ranks take one extra clip."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradAllReduce:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.pending = []

    def reduce_async(self, flat, offset, count):
        """Start summing flat[offset:offset+count] over ranks; overlaps whatever is enqueued next."""
        if self.world == 1 or count == 0:
            return
        self.pending.append(dist.all_reduce(flat[offset:offset + count], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def broadcast_params(self, flat, src=0):
        """Identical initial parameters on every rank (rank 0's)."""
        if self.world > 1:
            dist.broadcast(flat, src=src, group=self.group)

    def sum_scalars(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t
