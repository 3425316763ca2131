"""Data-parallel path on CPU: world_size 2, gloo.  Checks that the bucketed SUM all-reduce of per-rank
gradients (each scaled by 1/(local_rows*world)) equals the full-batch gradient, i.e. N-rank == 1-rank
(SURVEY 8e), plus the shard arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lrcn_oracle as O


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def flat_from(grads, specs):
    return np.concatenate([np.asarray(grads[n], np.float64).ravel() for n, _ in specs])


def worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from vltf_amd import dp
    from vltf_amd.engine import NetConfig, param_specs
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    shape, ncls, fpc, clips, hid = (67, 67, 3), 5, 2, 4, 6
    cfg = NetConfig(image_shape=shape, num_classes=ncls, fpc=fpc, lstm_hidden=hid)
    specs = param_specs(cfg)
    rng = np.random.default_rng(7)                       # identical on every rank
    p = O.init_params(rng, ncls, "fc6", hid, 1, shape, well_scaled=True)
    frames = rng.integers(0, 256, (clips * fpc,) + shape).astype(np.float32) - 100.0
    onehot = O.labels_to_one_hot([[l] for l in rng.integers(0, ncls, clips)], ncls)
    lo, hi = dp.shard_range(clips, rank, world)
    # local backward with the loss gradient scaled by 1/(local_rows*world)
    logits, cache = O.lrcn_forward(p, frames[lo * fpc:hi * fpc], fpc, keep=True)
    _, dlog = O.softmax_xent_mean(logits, onehot[lo:hi])
    g = O.lrcn_backward(p, cache, dlog / world, fpc)
    flat = torch.from_numpy(flat_from(g, specs))
    total = flat.numel()
    first_conv = sum(int(np.prod(s)) for n, s in specs[:[n for n, _ in specs].index("dcnn/conv5W")])
    gar = dp.GradAllReduce()
    assert gar.world == world and gar.rank == rank
    gar.reduce_async(flat, 0, first_conv)                # bucket 0: classifier + fc
    gar.reduce_async(flat, first_conv, total - first_conv)
    gar.wait()
    params = torch.from_numpy(flat_from(p, specs)) + rank   # deliberately different, then broadcast
    gar.broadcast_params(params, src=0)
    if rank == 0:
        logits_f, cache_f = O.lrcn_forward(p, frames, fpc, keep=True)
        _, dlog_f = O.softmax_xent_mean(logits_f, onehot)
        want = flat_from(O.lrcn_backward(p, cache_f, dlog_f, fpc), specs)
        q.put((float(np.abs(flat.numpy() - want).max()), float(np.abs(want).max()),
               float(np.abs(params.numpy() - flat_from(p, specs)).max())))
    else:
        q.put((float(np.abs(params.numpy() - flat_from(p, specs)).max()),))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    full = [r for r in res if len(r) == 3][0]
    assert full[0] < 1e-12 * max(1.0, full[1]), full          # summed shard gradients == full-batch gradient
    for r in res:
        assert r[-1] == 0.0                                    # broadcast made parameters identical


def test_shard_range_covers_batch():
    from vltf_amd.dp import shard_range
    for total in (64, 7, 1, 0):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(64, 3, 8) == (24, 32)


def test_single_process_is_a_noop():
    from vltf_amd.dp import GradAllReduce
    gar = GradAllReduce()
    t = torch.ones(10)
    gar.reduce_async(t, 0, 10)
    gar.wait()
    assert gar.world == 1 and float(t.sum()) == 10.0
