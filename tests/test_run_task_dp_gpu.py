"""`run_task.py <config.yml>` data parallel: two ranks (gloo, both on the box's one GPU) train over the same TFRecords, each on
its videos of every global batch; the weights rank 0 saves must equal those of a one-process run of the same config
(center crop, no dropout, same seed), including a ragged last batch (3 videos over 2 ranks) and clips-per-video that vary."""
import glob
import os
import pickle
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp
import yaml

from tests.test_host_workflow import make_dataset
from tests.test_run_task_gpu import RAW, write_cfg

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, cfg_path, val_cfg=None):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      VLTF_DIST_BACKEND="gloo")
    from vltf_amd import run_task
    run_task.main(cfg_path, seed=3, device="cuda:0")
    if val_cfg:
        run_task.main(val_cfg, device="cuda:0")


def latest_weights(run):
    ck = sorted(glob.glob(os.path.join(run, "checkpoints", "*.weights.npz")), key=os.path.getmtime)
    with np.load(ck[-1], allow_pickle=False) as z:
        return {k: z[k] for k in z.files}, len(ck)


def cfg_for(folder, data_path):
    path = write_cfg(folder, "train.yml", data_path, "train", epochs=2)
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg["run"]["data"]["d"]["imgproc"] = ["defs.imgproc.center_crop", "defs.imgproc.sub_mean"]     # deterministic input
    cfg["run"]["train"]["batch_size"] = 4
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def val_cfg_for(folder, data_path):
    path = write_cfg(folder, "val.yml", data_path, "val", resume="latest")
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg["run"]["val"]["batch_size"] = 4
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def test_two_rank_run_task_equals_one_rank(tmp_path):
    single, multi = str(tmp_path / "one"), str(tmp_path / "two")
    os.makedirs(single)
    os.makedirs(multi)
    # 7 videos, global batch 4 -> batches of 4 and 3 videos: shards 2+2 and 2+1; clips per video vary
    p1, _, _ = make_dataset(single, "train.txt", nvid=7, cpv=(1, 2, 1, 1, 2, 1, 1), shape=RAW, seed=1)
    p2, _, _ = make_dataset(multi, "train.txt", nvid=7, cpv=(1, 2, 1, 1, 2, 1, 1), shape=RAW, seed=1)
    from vltf_amd import run_task
    run_task.main(cfg_for(single, p1), seed=3)
    want, nck1 = latest_weights(os.path.join(single, "run"))

    cfg2 = cfg_for(multi, p2)
    # validation of the latest checkpoint over 5 videos (batches of 4 + 1: the second rank's shard of the last batch is empty)
    v1, _, _ = make_dataset(single, "val.txt", nvid=5, cpv=(2, 1, 2, 1, 1), shape=RAW, seed=2)
    v2, _, _ = make_dataset(multi, "val.txt", nvid=5, cpv=(2, 1, 2, 1, 1), shape=RAW, seed=2)
    acc1 = run_task.main(val_cfg_for(single, v1))
    ctx = mp.get_context("spawn")
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, cfg2, val_cfg_for(multi, v2))) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0, "rank exited with %s" % pr.exitcode
    got, nck2 = latest_weights(os.path.join(multi, "run"))
    assert nck1 == nck2                                      # rank 0 alone wrote the checkpoints
    assert len(glob.glob(os.path.join(multi, "run", "log_e2e_train_*.log"))) == 1       # one log file per run, rank 0's
    assert len(glob.glob(os.path.join(multi, "run", "log_e2e_val_*.log"))) == 1
    for k in want:
        upd = np.abs(want[k]).max() + 1e-12
        assert np.abs(got[k] - want[k]).max() < 2e-5 * upd, k
    # sharded validation: same per-video logits file and accuracy as the one-process validation
    tot1 = glob.glob(os.path.join(single, "run", "validation_logits_*.total"))
    tot2 = glob.glob(os.path.join(multi, "run", "validation_logits_*.total"))
    assert len(tot1) == 1 and len(tot2) == 1
    with open(tot1[0], "rb") as f1, open(tot2[0], "rb") as f2:
        l1, l2 = pickle.load(f1), pickle.load(f2)                     # files these runs wrote
    assert l1.shape == l2.shape == (5, 4)
    np.testing.assert_allclose(l2, l1, rtol=1e-4, atol=1e-4)
    assert float(open(glob.glob(os.path.join(multi, "run", "accuracy_*"))[0]).read()) == acc1
