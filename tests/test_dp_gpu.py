"""Data-parallel train step END TO END on the device: two ranks (both on cuda:0, gloo backend -- RCCL refuses two ranks on one
GPU, and the box has one) each run LRCNEngine.train_step on their half of the clips with the bucketed gradient all-reduce
inside; the updated parameters must equal those of ONE engine stepping on all the clips (N-rank == 1-rank).
Exercises what the CPU gloo test cannot: the 1/(rows*world) loss scaling in softmax_xent, the bucket boundaries in the flat
gradient buffer, the async reduce overlapping the conv backward launches, wait() before the global-norm clip."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import lrcn_oracle as O

pytestmark = pytest.mark.gpu
MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vltf_amd import dp
    from vltf_amd.engine import LRCNEngine, NetConfig
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    shape, ncls, fpc, clips, hid = (67, 67, 3), 5, 2, 4, 6
    cfg = NetConfig(image_shape=shape, num_classes=ncls, fpc=fpc, lstm_hidden=hid)
    rng = np.random.default_rng(11)                                  # identical on every rank
    p = O.init_params(rng, ncls, "fc6", hid, 1, shape, well_scaled=True)
    frames = rng.integers(0, 256, (clips * fpc,) + shape, dtype=np.uint8)
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, clips)], ncls)
    lo, hi = dp.shard_range(clips, rank, world)
    gar = dp.GradAllReduce()
    eng = LRCNEngine(cfg, max_clips=hi - lo, device="cuda:0", dp=gar)
    eng.load_params(p)
    gar.broadcast_params(eng.w)
    out = eng.train_step_u8(torch.tensor(frames[lo * fpc:hi * fpc], device="cuda:0"), torch.tensor(onehot[lo:hi], device="cuda:0"),
                            lr=0.05, clip_norm=0.5, mean_bgr=MEAN)
    got = eng.get_params()
    if rank == 0:
        ref = LRCNEngine(cfg, max_clips=clips, device="cuda:0")       # one engine, all the clips
        ref.load_params(p)
        want_out = ref.train_step_u8(torch.tensor(frames, device="cuda:0"), torch.tensor(onehot, device="cuda:0"), lr=0.05,
                                     clip_norm=0.5, mean_bgr=MEAN)
        want = ref.get_params()
        err = {k: float(np.abs(got[k] - want[k]).max() / (np.abs(want[k] - p[k]).max() + 1e-12)) for k in want}
        q.put(("ok", max(err.values()), abs(out["grad_norm"] - want_out["grad_norm"]) / want_out["grad_norm"]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_equals_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0, "rank exited with %s" % pr.exitcode
    tag, worst, gn_err = q.get(timeout=10)
    # the UPDATE (new - old parameter) of every tensor agrees to 1e-3 of its largest element (measured 3e-4: the update of a
    # bias is ~1e-4 of the parameter, so one fp32 ulp of the parameter is already ~1e-4 of the update); the global gradient
    # norm agrees to fp32 rounding.  The only difference between the two runs is the summation order across the batch halves.
    assert tag == "ok" and worst < 1e-3 and gn_err < 1e-5, (worst, gn_err)


def nccl_worker(port, q, math="f32"):
    """ONE rank, backend nccl (= RCCL): the collective path of the data-parallel step on a single GPU."""
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vltf_amd import dp
    from vltf_amd.engine import LRCNEngine, NetConfig
    r, w, _ = dp.init_from_env(backend="nccl", force=True)
    assert (r, w) == (0, 1) and torch.distributed.get_backend() == "nccl"
    shape, ncls, fpc, clips, hid = (67, 67, 3), 5, 2, 4, 6
    cfg = NetConfig(image_shape=shape, num_classes=ncls, fpc=fpc, lstm_hidden=hid, conv_math=math)
    rng = np.random.default_rng(11)
    p = O.init_params(rng, ncls, "fc6", hid, 1, shape, well_scaled=True)
    frames = torch.tensor(rng.integers(0, 256, (clips * fpc,) + shape, dtype=np.uint8), device="cuda:0")
    onehot = torch.tensor(O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, clips)], ncls), device="cuda:0")
    gar = dp.GradAllReduce(always=True)
    eng = LRCNEngine(cfg, max_clips=clips, device="cuda:0", dp=gar)
    ref = LRCNEngine(cfg, max_clips=clips, device="cuda:0")
    eng.load_params(p)
    ref.load_params(p)
    gar.broadcast_params(eng.w)
    outs = []
    for _ in range(3):                                  # several steps: the RCCL stream must also order against the NEXT step's writes
        a = eng.train_step_u8(frames, onehot, lr=0.05, clip_norm=0.5, mean_bgr=MEAN)
        b = ref.train_step_u8(frames, onehot, lr=0.05, clip_norm=0.5, mean_bgr=MEAN)
        outs.append((a["loss"], b["loss"], a["grad_norm"], b["grad_norm"]))
    got, want = eng.get_params(), ref.get_params()
    same = all(np.array_equal(got[k], want[k]) for k in want)
    worst = max(float(np.abs(got[k] - want[k]).max()) for k in want)
    q.put(("ok", same, worst, outs, gar.issued, len(eng.grad_chunks)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_one_rank_rccl_step_equals_plain_step():
    """The first thing RCCL has to get right is ordering: every chunk's all-reduce runs on RCCL's stream behind the GEMM that
    produced it, and the clip + SGD kernels on the launch stream wait for all of them.  With one rank the sum is the identity, so
    three data-parallel steps must leave exactly the parameters three plain steps leave."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=nccl_worker, args=(free_port(), q))
    pr.start()
    pr.join(300)
    assert pr.exitcode == 0, "rank exited with %s" % pr.exitcode
    tag, same, worst, outs, issued, nchunks = q.get(timeout=10)
    assert tag == "ok" and issued == 3 * nchunks and nchunks >= 3, (issued, nchunks)
    assert same, "parameters differ from the plain step by up to %.3e; losses / norms per step: %s" % (worst, outs)


def test_one_rank_rccl_step_on_the_bf16_path():
    """BASELINE config 5 is the bf16 conv path under data parallelism: the same one-rank RCCL step with conv_math="bf16" (packed
    operands, csrc/conv_c8.hip).  The data-parallel form computes fc6's weight gradient in four row blocks on the packed-operand
    kernel (engine._backward: one vl_gemm_kc8 per block, its all-reduce issued right behind it) where the plain step runs one
    product over all rows; the split-k slab count may differ between the two, so the runs are held to bf16 level, not bitwise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=nccl_worker, args=(free_port(), q, "bf16"))
    pr.start()
    pr.join(300)
    assert pr.exitcode == 0, "rank exited with %s" % pr.exitcode
    tag, same, worst, outs, issued, nchunks = q.get(timeout=10)
    assert tag == "ok" and issued == 3 * nchunks and nchunks >= 3, (issued, nchunks)
    assert worst < 2e-2, "parameters differ from the plain step by up to %.3e; losses / norms per step: %s" % (worst, outs)
    for la, lb, ga, gb in outs:
        assert abs(la - lb) < 2e-2 * max(1.0, abs(lb)) and abs(ga - gb) < 5e-2 * gb, outs
