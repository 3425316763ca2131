"""Validation: clip -> video logit fusion, accuracy, pickled logits with the reference's file names
(val.py:59-203).  Host numpy; the logits come from LRCNEngine.forward_u8."""
import os
import pickle

import numpy as np

from .defs_ import defs
from .utils_ import debug, error, info


class Validation:
    def __init__(self, settings):
        self.item_logits = np.zeros([0, settings.num_classes], np.float32)
        self.item_labels = np.zeros([0, settings.num_classes], np.float32)
        self.save_counter = 0
        self.save_interval = settings.val.logits_save_interval
        self.run_folder, self.run_id, self.timestamp = settings.run_folder, settings.run_id, settings.timestamp

    def apply_clip_fusion(self, clips_logits, cpv, video_labels, clip_fusion):
        """val.py:158-167."""
        cur = clips_logits[0:cpv, :]
        if clip_fusion == defs.fusion_method.avg:
            video_logits = np.mean(cur, axis=0)
        elif clip_fusion == defs.fusion_method.last:
            video_logits = cur[-1, :]
        else:
            error("Undefined clip fusion [%s]" % clip_fusion)
        self.item_logits = np.vstack((self.item_logits, video_logits))
        self.item_labels = np.vstack((self.item_labels, video_labels[0, :]))

    def process_validation_logits(self, dataset, settings, logits, labels, batch_index=None):
        """val.py:59-113, video batch mode (the only working one in the reference).  batch_index: the dataset's batch index right
        after THIS batch was read (the reference reads dataset.batch_index here; with a read-ahead feeder that has moved on)."""
        maxvid = (dataset.batch_index if batch_index is None else batch_index) * dataset.batch_size
        for vidx in range(maxvid - dataset.batch_size, maxvid):
            if vidx >= dataset.num_items:
                break
            cpv = dataset.clips_per_video[vidx]
            self.apply_clip_fusion(logits, cpv, labels, settings.val.clip_fusion_method)
            logits, labels = logits[cpv:, :], labels[cpv:, :]
        if len(logits) or len(labels):
            error("Logits and/or labels non empty at the end of video item mode aggregation!")
        info("Incremental accuracy up to current batch: %2.3f" % np.mean(self.get_chunk_accuracy(self.item_logits, self.item_labels)))

    def add_items(self, logits, labels):
        """Logit rows that are items of their own (per-step logits of a `reshape`-fusion model: no clip -> video fusion)."""
        self.item_logits = np.vstack((self.item_logits, np.asarray(logits, np.float32)))
        self.item_labels = np.vstack((self.item_labels, np.asarray(labels, np.float32)))
        info("Incremental accuracy up to current batch: %2.3f" % np.mean(self.get_chunk_accuracy(self.item_logits, self.item_labels)))

    def save_validation_logits_chunk(self, save_all=False):
        """val.py:115-148: '<run_folder>/validation_logits_<run_id>_<ts>.total' or '.part_k'."""
        if self.save_interval is None or len(self.item_logits) == 0:
            return
        if self.save_interval <= 0:
            if save_all:
                path = os.path.join(self.run_folder, "validation_logits_%s_%s.total" % (self.run_id, self.timestamp))
                info("Saving all %d extracted validation logits to %s" % (len(self.item_logits), path))
                with open(path, "wb") as f:
                    pickle.dump(self.item_logits, f)
            return
        if len(self.item_logits) >= self.save_interval or save_all:
            path = os.path.join(self.run_folder, "validation_logits_%s_%s.part_%d" % (self.run_id, self.timestamp, self.save_counter))
            info("Saving a %d-sized chunk of validation logits to %s" % (len(self.item_logits), path))
            with open(path, "wb") as f:
                pickle.dump(self.item_logits, f)
            self.item_logits = np.zeros([0, self.item_logits.shape[-1]], np.float32)
            self.save_counter += 1

    def load_validation_logits_chunk(self, idx):
        path = os.path.join(self.run_folder, "validation_logits_%s_%s.part_%d" % (self.run_id, self.timestamp, idx))
        with open(path, "rb") as f:
            return pickle.load(f)          # a file this code wrote

    def get_accuracy(self):
        """val.py:174-197: mean of per-chunk accuracies."""
        accuracies, cur = [], 0
        for k in range(self.save_counter):
            logits = self.load_validation_logits_chunk(k)
            accuracies.append(self.get_chunk_accuracy(logits, self.item_labels[cur:cur + len(logits), :]))
            cur += len(logits)
        if len(self.item_logits) > 0:
            n = len(self.item_logits)
            accuracies.append(self.get_chunk_accuracy(self.item_logits, self.item_labels[cur:cur + n, :]))
        return float(np.mean(accuracies))

    def get_chunk_accuracy(self, logits, labels):
        return np.mean(np.equal(np.argmax(logits, axis=1), np.argmax(labels, axis=1)))
