"""Train: learning-rate table and the per-step call with the reference's surface (train.py:50-149, 199-222)."""
import math
import os

import numpy as np
import torch

from .defs_ import defs
from .utils_ import error, info


def precompute_learning_rates(settings, num_batches):
    """train.py:50-109, including the schedule dump to <run_id>_lr_decay_schedule.txt.  Because idx advances by
    decay_freq per block, exp and staircase give the same piecewise-constant table."""
    base_lr, decay_params = settings.train.base_lr, settings.train.lr_decay
    total = num_batches * settings.train.epochs
    if decay_params is None:
        return [base_lr] * total
    offset = 0 if len(tuple(decay_params)) == 4 else decay_params[-1]
    strategy, scheme, freq, factor = tuple(decay_params[:4])
    if strategy not in (defs.decay.exp, defs.decay.staircase):
        error("Undefined decay strategy %s" % strategy)
    staircase = strategy == defs.decay.staircase
    if scheme == defs.periodicity.interval:
        period = freq
    elif scheme == defs.periodicity.drops:
        period = math.ceil(total / freq)
    else:
        error("Undefined decay scheme %s" % scheme)
    lrs, idx = [], 0
    while len(lrs) < total:
        fraction = idx // freq if staircase else idx / freq
        lrs.extend([base_lr * pow(factor, fraction)] * period)
        idx += freq
    lrs = lrs[:total]
    if offset:
        lrs = [base_lr] * offset + lrs[0:-offset]
    path = os.path.join(settings.run_folder, settings.run_id + "_lr_decay_schedule.txt")
    with open(path, "w") as f:
        k = 0
        for ep in range(settings.train.epochs):
            for b in range(num_batches):
                f.write("Epoch %d/%d, batch %d/%d, lr %2.8f\n" % (ep + 1, settings.train.epochs, b + 1, num_batches, lrs[k]))
                k += 1
    info("Dropping LR of %2.5f, mid / last lr is: %1.5f, %1.5f, total drops: %d" % (base_lr, lrs[len(lrs) // 2], lrs[-1], len(set(lrs))))
    return lrs


class Train:
    """train.py:112-149: owns the LR table and global_step; run_step is the train sess.run."""

    def __init__(self, settings, feeder, engine):
        self.engine = engine
        self.learning_rates = precompute_learning_rates(settings, feeder.get_num_batches())
        self.global_step = settings.global_step
        self.clip_norm = float(settings.train.clip_norm or 0)

    def run_step(self, fdict, others=None):
        """-> (loss, current_lr, global_step) like sess.run([.., loss, current_lr, global_step, optimizer]).
        others: {tag: batch} of the other datasets of a multi-pipeline model (fdict is the MAIN dataset's: the labels are its, train.py:117)."""
        if others is not None:
            return self.run_step_graph(fdict, others)
        if self.global_step >= len(self.learning_rates):
            error("global step %d exceeds the precomputed learning-rate table (%d)" % (self.global_step, len(self.learning_rates)))
        lr = float(self.learning_rates[self.global_step])
        dev = self.engine.dev
        eng, dpg = self.engine, self.engine.dp
        # data parallel: the loss is the mean over the GLOBAL batch, of which this rank holds a shard (possibly ragged or empty)
        per_clip = eng.cfg.classifier == "lstm" or eng.early or eng.late          # one logits row per clip
        grows = None
        if dpg is not None and fdict.get("global_clips") is not None:
            # per-frame head (classifier fc, no frame fusion): one logits row per frame -> global rows = global clips * fpc
            grows = fdict["global_clips"] * (1 if per_clip else eng.cfg.fpc)
        if len(fdict["labels"]) == 0:
            out = eng.train_step_empty(lr, self.clip_norm)
        elif "device" in fdict:          # uploaded ahead of time by the feeder's BatchPrefetcher: wait for the copy on the stream
            torch.cuda.current_stream(dev).wait_event(fdict["ready"])
            t = fdict["device"]
            out = eng.train_step_u8(t["frames_u8"], t["labels"], lr, self.clip_norm, fdict["mean_bgr"], t["crop_y"], t["crop_x"],
                                    t["mirror"], global_rows=grows, resize=fdict.get("resize"))
        else:
            out = eng.train_step_u8(torch.from_numpy(fdict["frames_u8"]).to(dev, non_blocking=True),
                                    torch.from_numpy(fdict["labels"]).to(dev),
                                    lr, self.clip_norm, fdict["mean_bgr"],
                                    torch.from_numpy(fdict["crop_y"]).to(dev), torch.from_numpy(fdict["crop_x"]).to(dev),
                                    torch.from_numpy(fdict["mirror"]).to(dev), global_rows=grows, resize=fdict.get("resize"))
        if dpg is not None:              # log the global-batch loss, not the shard's
            tot = dpg.sum_scalars(torch.tensor([out["loss_sum"], out["correct"], float(out["rows"])], device=dev, dtype=torch.float64))
            tot = tot.cpu().numpy()
            out = dict(out, loss=float(tot[0] / max(tot[2], 1)), accuracy=float(tot[1] / max(tot[2], 1)))
        self.global_step += 1
        self.last = out
        return out["loss"], lr, self.global_step

    def run_step_graph(self, fdict, others):
        """Multi-pipeline model (vltf_amd.graph): one batch per dataset tag; a per-step model (the last pipeline an LSTM with
        fusion reshape) takes the main dataset's per-record targets, any other its per-clip ones."""
        from .defs_ import defs
        from .run_task import graph_feeds
        if self.global_step >= len(self.learning_rates):
            error("global step %d exceeds the precomputed learning-rate table (%d)" % (self.global_step, len(self.learning_rates)))
        lr = float(self.learning_rates[self.global_step])
        eng, dev = self.engine, self.engine.dev
        labels = fdict["record_labels"] if (eng.per_step and "record_labels" in fdict) else fdict["labels"]
        grows = None
        if eng.dp is not None and fdict.get("global_clips") is not None and len(fdict["labels"]):
            # rows of the GLOBAL batch: the main dataset's global clips times this model's logits rows per main clip
            grows = fdict["global_clips"] * len(labels) // len(fdict["labels"])
        if len(labels) == 0:                 # this rank's shard of a short last batch is empty
            out = eng.train_step_empty(lr, self.clip_norm)
        else:
            fdicts = dict(others)
            fdicts[defs.dataset_tag.main] = fdict
            out = eng.train_step(graph_feeds(fdicts, sorted(fdicts), dev), torch.from_numpy(np.ascontiguousarray(labels)).to(dev), lr,
                                 self.clip_norm, global_rows=grows)
        if eng.dp is not None:
            tot = eng.dp.sum_scalars(torch.tensor([out["loss_sum"], out["correct"], float(out["rows"])], device=dev, dtype=torch.float64))
            tot = tot.cpu().numpy()
            out = dict(out, loss=float(tot[0] / max(tot[2], 1)), accuracy=float(tot[1] / max(tot[2], 1)))
        self.global_step += 1
        self.last = out
        return out["loss"], lr, self.global_step
