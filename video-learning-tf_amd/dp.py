"""Data parallelism over clips: one process per GPU, gradients summed with an RCCL all-reduce
(torch.distributed backend "nccl" IS RCCL on ROCm) over xGMI.

The reference is single process (run_task.py:132); this is what the build adds.
Clips are independent units: rank r takes clips [r*B/W, (r+1)*B/W) of each global batch.  Every
rank scales its loss gradient by 1/(local_rows * world) (the loss is a batch mean, train.py:123),
so a SUM all-reduce of the flat gradient buffer yields the global-batch gradient on every rank;
global-norm clipping and SGD then run identically everywhere (train.py:215-217).

Chunks in the order backward produces them (engine.grad_chunks): [classifier + LSTM], then the fc
weight gradients in row blocks -- each block's all-reduce is issued right after the GEMM that
produces it, so the ring starts while the next block is still being computed and 85 % of the bytes
are in flight before conv5..conv1 backward (94 % of the FLOPs) begins --, and [conv] at the end.
xGMI is point to point: a ring all-reduce of S bytes moves 2 (W-1)/W S through one link per rank
(~153 GB/s), so chunks of 10-40 MB keep the per-launch latency (~20 us) negligible while the first
one starts ~0.1 ms after the backward pass does.
torch.distributed launches each all-reduce on RCCL's own stream behind the work already queued on
the compute stream, and wait() makes the compute stream wait for it -- no host synchronisation."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract).
    Returns (rank, world, local_rank); world == 1 means no process group unless `force` (or VLTF_DIST_FORCE=1) asks for a
    one-rank group -- the collective path (RCCL stream ordering against the launch stream) then runs on a single GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = force or os.environ.get("VLTF_DIST_FORCE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # VLTF_DIST_BACKEND=gloo: rehearse the multi-rank path where RCCL cannot run (several ranks on one GPU)
            backend = os.environ.get("VLTF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def pretouch_gpu(gpus=1):
    """Run a tiny GPU process to completion BEFORE this process (or its ranks) initialises HIP; True if one ran.
    Measured on this pool (round 4): the FIRST GPU process on a freshly acquired MI355X box executes short kernels slower
    for its whole life -- the 8-clip train step in 6.1 - 6.6 ms instead of 5.2 (4 of 5 fresh boxes; host issue time identical; 2,500
    warm-up steps or re-allocating the engine's memory do not cure it) -- while every later process runs at full speed, also when the
    first one only summed a vector (3 of 3 fresh boxes).  A benchmark that may be the first process on its box therefore sends such a
    process ahead.  Which cards it touches: under a launcher (LOCAL_RANK set) only this rank's own -- N ranks must not put N extra
    processes on card 0 -- else cards 0 .. gpus-1, one after the other in ONE process (the ranks `self_launch` starts inherit the
    marker and skip it).  VLTF_NO_PRETOUCH=1 skips it; the marker VLTF_GPU_PRETOUCHED keeps ranks from repeating it."""
    if os.environ.get("VLTF_GPU_PRETOUCHED") == "1" or os.environ.get("VLTF_NO_PRETOUCH") == "1":
        return False
    import subprocess
    import sys
    if "LOCAL_RANK" in os.environ:
        cards = [int(os.environ["LOCAL_RANK"])]
    else:
        cards = list(range(max(1, int(gpus))))
    code = ("import torch\n"
            "if torch.cuda.is_available():\n"
            "    for d in %r:\n"
            "        if d < torch.cuda.device_count():\n"
            "            x = torch.zeros(1 << 24, device='cuda:%%d' %% d)\n"
            "            float((x + 1).sum().item())\n" % (cards,))
    try:
        subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    except Exception:
        return False
    os.environ["VLTF_GPU_PRETOUCHED"] = "1"
    return True


def self_launch(gpus, argv=None):
    """`python <script> --gpus N` started WITHOUT a launcher (WORLD_SIZE unset, N > 1): start the N ranks ourselves -- one process per
    GPU under `python -m torch.distributed.run` on 127.0.0.1 with a free port -- pass their output through and return the launcher's
    exit code (the caller exits with it).  Returns None when nothing is to be done (N == 1, or a launcher already set WORLD_SIZE).
    Must be called BEFORE anything touches the GPU: the ranks are child processes of a parent that never initialises HIP (a process
    that has must not exec / fork into GPU work on this platform)."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    argv = list(sys.argv if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between the ranks' processes on this platform
    return subprocess.run(cmd, env=env).returncode


def shard_range(total, rank, world):
    """Contiguous clip range of `rank`: the first total % world ranks take one extra clip."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradAllReduce:
    def __init__(self, group=None, always=False):
        """always: issue the collectives even in a one-rank group (smoke test of the RCCL path on one GPU)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.always = bool(always) and dist.is_initialized()
        self.pending = []
        self.issued = 0                  # collectives launched (tests / logs)

    def reduce_async(self, flat, offset, count):
        """Start summing flat[offset:offset+count] over ranks; overlaps whatever is enqueued next."""
        if (self.world == 1 and not self.always) or count == 0:
            return
        self.issued += 1
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
