"""Feeder: batch delivery and checkpoint save / resume with the reference's surface (feeder.py).
The feed dict of the reference (placeholder -> numpy) becomes a plain dict of device-ready arrays."""
import os
import pickle
import queue
import threading

import numpy as np

from . import dataset_
from .defs_ import defs
from .utils_ import debug, error, get_datetime_str, get_run_checkpoints, info, warning


class BatchPrefetcher:
    """Reads the remaining batches of the current epoch ahead of the training loop.

    The reference feeds synchronously: parse a batch on the host, sess.run, repeat (run_task.py:25-81).  At the device's rate
    that leaves the GPU idle more than half of the time: one 64-clip batch is 236 MB of records, ~45 ms to read and checksum on
    one thread and 4 ms to upload, against a 37 ms train step.  Here a background thread produces batch k+1 .. k+depth while
    batch k trains: the native reader (several threads) writes the frames straight into a pinned slot, the slot is uploaded on a
    side stream, and the training loop receives device tensors plus the event that marks the upload done.
    Everything order-dependent stays on the producer thread in batch order (record offsets, the per-frame crop / mirror draws
    of dataset_.py:444-461), so a run is identical to the synchronous one.  `consumed` counts batches handed out: checkpoints
    record it, not the read-ahead position."""

    def __init__(self, dataset, device, depth=2):
        import torch
        self.torch, self.d, self.device, self.depth = torch, dataset, torch.device(device), max(1, int(depth))
        self.consumed = dataset.batch_index
        self.q, self.thread, self.stop_flag = None, None, None
        h, w, c = dataset.stored_image_shape
        nmax = max(dataset.frames_in_batch(b) for b in range(len(dataset.batches))) if dataset.batches else 0
        self.slots = []
        # batch j trains while j+1 .. j+depth wait in the queue and the producer fills j+depth+1: depth + 2 live batches
        for _ in range(self.depth + 2):
            self.slots.append({"pin": torch.empty((nmax, h, w, c), dtype=torch.uint8, pin_memory=True),
                               "dev": torch.empty((nmax, h, w, c), dtype=torch.uint8, device=self.device)})
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _run(self, q, stop, first_slot):
        torch, d = self.torch, self.d
        torch.cuda.set_device(self.device)
        k = first_slot
        try:
            while d.loop() and not stop.is_set():
                slot = self.slots[k % len(self.slots)]
                frames, cy, cx, mirror, onehot = d.get_next_batch(out=slot["pin"].numpy())
                n = len(frames)
                with torch.cuda.stream(self.copy_stream):
                    slot["dev"][:n].copy_(slot["pin"][:n], non_blocking=True)
                    dev = {key: torch.from_numpy(v).to(self.device, non_blocking=True)
                           for key, v in (("crop_y", cy), ("crop_x", cx), ("mirror", mirror), ("labels", onehot))}
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                item = {"frames_u8": frames, "crop_y": cy, "crop_x": cx, "mirror": mirror, "labels": onehot, "mean_bgr": d.mean_bgr,
                        "resize": d.resize_chain, "dataset": d, "batch_index": d.batch_index, "global_clips": d.global_clips, "device": dict(dev, frames_u8=slot["dev"][:n]), "ready": ev}
                k += 1
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        pass
            q.put(None)
        except BaseException as ex:          # surfaces in the training loop, like a synchronous read error would
            q.put(ex)

    def next(self):
        if self.thread is None:
            self.q, self.stop_flag = queue.Queue(maxsize=self.depth), threading.Event()
            self.thread = threading.Thread(target=self._run, args=(self.q, self.stop_flag, self.consumed), daemon=True)
            self.thread.start()
        item = self.q.get()
        if isinstance(item, BaseException):
            self.stop()
            raise item
        if item is None:
            error("No batch left in this epoch.")
        self.consumed += 1
        return item

    def stop(self):
        """End of epoch / rewind: the producer has read exactly the epoch's batches (or is told to quit)."""
        if self.thread is not None:
            self.stop_flag.set()
            while self.thread.is_alive():       # drain so a blocked put() returns
                try:
                    self.q.get_nowait()
                except queue.Empty:
                    pass
                self.thread.join(0.05)
            self.thread = None


class Feeder:
    def __init__(self, input_mode, phases, trainval, save_freq_per_epoch, run_folder, resume):
        """feeder.py:16-29."""
        self.datasets = {}
        self.input_mode, self.phases, self.phase = input_mode, phases, None
        self.run_folder, self.resume = run_folder, resume
        self.train, self.val = trainval
        self.save_freq_per_epoch = save_freq_per_epoch
        self.save_interval, self.num_saves = -1, 0
        self.saved = []
        self.prefetch = None          # the read-ahead of the MAIN dataset (None: synchronous feed)
        self.prefetchers = {}         # tag -> BatchPrefetcher: every frame dataset a multi-pipeline model reads

    def add_dataset(self, dataset_phase, id, path, mean_image, prepend_folder, image_shape, imgproc, raw_image_shape, data_format,
                    frame_format, batch_item, num_classes, tag, read_tries, captioning_config=None):
        """feeder.py:31-38."""
        dset = dataset_.Dataset()
        self.datasets.setdefault(dataset_phase, []).append(dset)
        dset.initialize(id, path, mean_image, prepend_folder, image_shape, imgproc, raw_image_shape, data_format, frame_format,
                        batch_item, num_classes, tag, read_tries)

    def set_phase(self, phase):
        self.phase = phase

    def initialize_datasets(self):
        """feeder.py:43-54."""
        if not self.datasets:
            error("No dataset configured to active phase [%s]" % self.phase)
        for phase in self.phases:
            for i, dset in enumerate(self.datasets.get(phase, [])):
                info("Reading dataset %d / %d : [%s]" % (i + 1, len(self.datasets[phase]), dset.id))
                bs = self.train.batch_size if (defs.phase.train in self.phases and self.train) else self.val.batch_size
                dset.calculate_batches(bs, self.input_mode)

    def _any_prefetch(self):
        return next(iter(self.prefetchers.values())) if self.prefetchers else None

    def loop(self):
        pf = self._any_prefetch()
        if pf is not None:            # a dataset that is read ahead is past the batch the loop is at: count what was handed out
            return pf.consumed < self.get_num_batches()
        return self.datasets[self.phase][0].loop()

    def get_dataset_by_tag(self, tag):
        return [d for d in self.datasets[self.phase] if d.tag == tag]

    def get_datasets(self):
        return self.datasets[self.phase]

    def get_num_batches(self):
        return len(self.datasets[self.phase][0].batches) if self.datasets else -1

    def get_batch_index(self):
        pf = self._any_prefetch()
        if pf is not None:
            return pf.consumed                  # batches handed to the training loop, not batches read ahead
        return self.datasets[self.phase][0].batch_index

    def get_batch_sizes(self):
        return [d.batch_size for d in self.datasets[self.phase]]

    def rewind_datasets(self):
        for pf in self.prefetchers.values():
            pf.stop()
        for d in self.datasets[self.phase]:
            d.rewind()
        for pf in self.prefetchers.values():
            pf.consumed = 0

    def enable_prefetch(self, device, depth=2, tag=defs.dataset_tag.main):
        """Read ahead of the training loop (BatchPrefetcher).  get_feed_dict / loop / get_batch_index keep their meaning."""
        dsets = self.get_dataset_by_tag(tag)
        if len(dsets) != 1:
            error("%d datasets satisfy the network input requirement [%s], but exactly one must." % (len(dsets), tag))
        if dsets[0].input_mode == defs.input_mode.vectors:
            return                                  # a few KB per batch: read synchronously
        self.prefetchers[tag] = BatchPrefetcher(dsets[0], device, depth)
        if tag == defs.dataset_tag.main:
            self.prefetch = self.prefetchers[tag]

    def get_feed_dict(self, tag=defs.dataset_tag.main):
        """feeder.py:84-106: -> (fdict, num_data, num_labels, padding).  fdict holds the raw uint8 frames and the
        device-side imgproc arguments instead of float32 frames."""
        dsets = self.get_dataset_by_tag(tag)
        if len(dsets) != 1:
            error("%d datasets satisfy the network input requirement [%s], but exactly one must." % (len(dsets), tag))
        d = dsets[0]
        if tag in self.prefetchers:
            fdict = self.prefetchers[tag].next()
            return fdict, [len(fdict["frames_u8"])], len(fdict["labels"]), 0
        frames, cy, cx, mirror, onehot = d.get_next_batch()
        if d.input_mode == defs.input_mode.vectors:            # float32 vectors instead of frames; per-record targets beside the per-clip ones
            fdict = {"vectors": frames, "labels": onehot, "record_labels": d.record_onehot, "dataset": d, "batch_index": d.batch_index,
                     "global_clips": d.global_clips}
            return fdict, [len(frames)], len(onehot), 0
        fdict = {"frames_u8": frames, "crop_y": cy, "crop_x": cx, "mirror": mirror, "labels": onehot, "mean_bgr": d.mean_bgr,
                 "resize": d.resize_chain, "dataset": d, "batch_index": d.batch_index, "global_clips": d.global_clips}
        return fdict, [len(frames)], len(onehot), 0

    # ---- save cadence (feeder.py:111-129) ---------------------------------------------------------------
    def compute_save_interval(self):
        if not self.train:
            self.save_interval, self.num_saves = -1, 0
            return
        for d in self.datasets[self.phase]:
            self.save_interval, self.num_saves = d.compute_dataset_portion(self.save_freq_per_epoch, self.train.epochs)

    def should_save(self, step):
        if self.save_interval < 0 or self.phase == defs.phase.val:
            return False
        return step % self.save_interval == 0

    # ---- checkpoints (feeder.py:143-288) -------------------------------------------------------------------
    # <run_folder>/checkpoints/<ddmmyy_HHMMSS>_ep_E_btch_B_gs_G.graph-<gs>.weights.npz  {tf variable name: array}
    #                                                            ...  .graph-<gs>.snap   pickle [batch_index, epoch_index, global_step]
    def _resolve(self, resume_file):
        if resume_file == defs.names.latest_savefile:
            ck = get_run_checkpoints(self.run_folder)
            if not ck:
                error("Specified resume file: [%s], but no checkpoint exists in %s" % (resume_file, self.run_folder))
            return ck[-1]
        return resume_file.strip("\"'")

    def resume_snap(self, resume_file):
        """feeder.py:143-194 -> (epoch_index, global_step); fast-forwards the record iterators."""
        if not self.resume:
            return None
        base = self._resolve(resume_file)
        snap = base + ".snap"
        info("Resuming metadata: [%s]" % snap)
        if not os.path.exists(snap):
            error("Metaparameters savefile does not exist: %s" % snap)
        with open(snap, "rb") as f:
            params = pickle.load(f)            # a file this code wrote
        batch_info, epoch = params[:2]
        global_step = params[2] if len(params) > 2 else int(os.path.basename(base).split("-")[-1])
        for d in self.get_datasets():
            idx = batch_info.get(d.tag, 0) if isinstance(batch_info, dict) else batch_info
            d.restore(idx, epoch)
        info("Restored training snapshot of epoch %d, train index %s, global step %d" % (epoch + 1, str(batch_info), global_step))
        return epoch, global_step

    def init_saveload(self, engine, resume_file, ignorable_variable_names=()):
        """feeder.py:198-257: restore variables; a variable-set mismatch fails instead of prompting."""
        self.compute_save_interval()
        if not self.resume:
            return
        base = self._resolve(resume_file)
        wfile = base + ".weights.npz"
        info("Resuming weights: [%s]" % wfile)
        if not os.path.exists(wfile) or not os.path.exists(base + ".snap"):
            error("Missing weights or snap part of savefile: %s" % base)
        with np.load(wfile, allow_pickle=False) as z:
            stored = {k: z[k] for k in z.files}
        opt_state = {k: stored.pop(k) for k in list(stored) if k.startswith(engine.OPT_PREFIX)}
        ignor = set(ignorable_variable_names) | {defs.names.global_step}
        want = {n for n, _ in engine.specs}
        missing = sorted(want - set(stored) - ignor)
        extra = sorted(set(stored) - want - ignor)
        if missing or extra:
            error("Failed to load checkpoint: variables missing from it %s, unknown to the network %s" % (missing, extra))
        engine.load_params({k: v for k, v in stored.items() if k in want})
        # tf.train.Saver() saves every global variable (feeder.py:201), i.e. also the Adam slots and beta powers: without them a
        # resumed adam run restarts its moments and bias correction.  Their absence (a weights-only file) means a fresh optimizer.
        if engine.training:
            snap_gs = None
            try:
                with open(base + ".snap", "rb") as f:
                    sp = pickle.load(f)          # a file this code wrote
                snap_gs = sp[2] if len(sp) > 2 else None
            except (OSError, pickle.UnpicklingError, IndexError):
                pass
            absent = engine.load_opt_state(opt_state, global_step=snap_gs)
            if absent:
                warning("Checkpoint holds no optimizer state %s: the optimizer starts fresh (step count from global_step)" % absent)

    def save(self, engine, progress, global_step):
        """feeder.py:263-288."""
        folder = os.path.join(self.run_folder, "checkpoints")
        os.makedirs(folder, exist_ok=True)
        base = os.path.join(folder, get_datetime_str() + "_" + progress) + ".graph-%d" % global_step
        info("Saving graph  to [%s]" % base)
        np.savez(base + ".weights.npz", **engine.get_params(), **(engine.get_opt_state() if engine.training else {}))
        info("Saving params for epoch index %d, train index %d" % (self.train.epoch_index + 1, self.get_batch_index()))
        with open(base + ".snap", "wb") as f:
            pickle.dump([self.get_batch_index(), self.train.epoch_index, global_step], f)
        with open(os.path.join(folder, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % base)
        self.saved.append(base)
        while self.num_saves > 0 and len(self.saved) > self.num_saves:      # tf.train.Saver(max_to_keep=num_saves)
            old = self.saved.pop(0)
            for suf in (".weights.npz", ".snap"):
                if os.path.exists(old + suf):
                    os.remove(old + suf)
        return base
