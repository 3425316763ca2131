"""Dataset: the reference's TFRecord-backed video dataset (dataset_.py) with the device-side split of
process_image: the host reads raw uint8 frames (native reader) and draws crop offsets / mirror flags
exactly as the reference does; crop, mean subtraction, mirror and the NHWC->NCHW float conversion run on
the device (vl_input_prep_u8).  Only input_mode video + data_format tfrecord + batch_item default are on
the path (Feeder hard-codes video, settings_.py:302; batch_item clip is broken in the reference)."""
import math
import os
import random

import numpy as np

from . import _hostio, tfrecord
from .defs_ import defs
from .utils_ import debug, error, info, labels_to_one_hot, warning


class Dataset:
    def initialize(self, id, path, mean_image, prepend_folder, desired_image_shape, imgproc, raw_image_shape, data_format,
                   frame_format, batch_item, num_classes, tag, read_tries):
        """dataset_.py:503-519."""
        info("Initializing dataset [%s]" % id)
        self.id, self.path, self.tag = id, path, tag
        self.data_format, self.frame_format, self.prepend_folder = data_format, frame_format, prepend_folder
        self.mean_image = mean_image
        self.desired_image_shape = tuple(desired_image_shape) if desired_image_shape is not None else None
        self.raw_image_shape = tuple(raw_image_shape) if raw_image_shape is not None else None
        self.imgproc = list(imgproc)
        self.batch_item, self.num_classes, self.read_tries = batch_item, num_classes, read_tries
        self.batch_index = self.epoch_index = 0
        self.batches, self.offset = None, 0
        self._vec_iter = None
        self.frames, self.labels = [], []
        self.rng = random.Random()          # the reference uses the unseeded `random` module (dataset_.py:454,498)
        if data_format != defs.data_format.tfrecord:
            # `data_format: raw` cannot run in the reference either: Dataset.get_next_batch (dataset_.py:246-262) indexes the current
            # batch as (paths, labels) -- `currentBatch[0]` -- but calculate_batches (dataset_.py:603-609) fills self.batches with batch
            # SIZES (ints) since the legacy initialize_data (616-696) went dead: the first batch raises TypeError.  Serialise the frame
            # folders instead (serialize.py writes the TFRecord + .size this reader takes).
            error("data_format [%s] is broken in the reference (dataset_.py:246-262 indexes an int batch size) and not built; "
                  "serialize the frames to a TFRecord (serialize.py) and use defs.data_format.tfrecord" % data_format)
        if batch_item != defs.batch_item.default:
            error("batch_item [%s] is not supported (broken in the reference, dataset_.py:415)" % batch_item)

    def seed(self, s):
        self.rng = random.Random(s)

    def get_image_shape(self):
        return self.desired_image_shape if self.desired_image_shape is not None else self.raw_image_shape

    def read_frames_metadata(self):
        """dataset_.py:71-84: 'path label...' per line."""
        self.frames, self.labels = [], []
        with open(self.path, "r") as f:
            for line in f:
                parts = line.split()
                if not parts:
                    continue
                item = parts[0]
                if self.prepend_folder:
                    item = os.path.join(self.prepend_folder, item)
                self.frames.append(item)
                self.labels.append(parts[1:])

    def get_input_data_count(self):
        """dataset_.py:701-756."""
        size_file = self.record_path + ".size"
        if not os.path.exists(size_file):
            error("Could not file data size file: %s" % size_file)
        d = tfrecord.read_size_file(size_file)
        if d["type"] == defs.input_mode.vectors:
            self.input_mode = defs.input_mode.vectors        # the size file decides (dataset_.py:709-714)
        elif d["type"] != self.input_mode:
            error("Specified input mode is [%s] but the size file contains [%s]" % (self.input_mode, d["type"]))
        if d["cpv"] is None or d["fpc"] is None:
            error("Read cpi: %s / fpc: %s but input mode is %s" % (d["cpv"], d["fpc"], self.input_mode))
        self.num_items = d["items"]
        self.clips_per_video = d["cpv"] if isinstance(d["cpv"], list) else [d["cpv"]] * self.num_items
        self.num_frames_per_clip = d["fpc"]
        self.max_caption_length = d["labelcount"]
        info("Read [%s] data, count: %d, fpc: %s, type: %s" % (self.id, self.num_items, self.num_frames_per_clip, self.input_mode))

    def compute_crop(self, raw_image_shape, image_shape, mode):
        """dataset_.py:571-577.  rand range(0, raw - want - 1) excludes the last two legal offsets, as the reference does."""
        if mode == defs.imgproc.center_crop:
            return tuple(int(np.floor((r - w) / 2)) for r, w in zip(raw_image_shape, image_shape))[:2]
        return (list(range(0, raw_image_shape[0] - image_shape[0] - 1)), list(range(0, raw_image_shape[1] - image_shape[1] - 1)))

    def initialize_imgproc(self):
        """dataset_.py:540-560 (+ build_mean_image 521-530: mean_image[0] is blue)."""
        if self.input_mode == defs.input_mode.vectors:
            if self.imgproc:
                info("Ignoring imgproc due to input mode: [%s]" % self.input_mode)
            self.imgproc, self.mean_bgr, self.crop_mode, self.resize_chain = [], None, None, []
            return
        self.mean_bgr = np.asarray(self.mean_image, np.float32) if defs.imgproc.sub_mean in self.imgproc else None
        self.crop_mode = None
        if defs.imgproc.rand_crop in self.imgproc:
            self.crop_mode = defs.imgproc.rand_crop
        elif defs.imgproc.center_crop in self.imgproc:
            self.crop_mode = defs.imgproc.center_crop
        if self.raw_image_shape is None:
            self.raw_image_shape = self.desired_image_shape
        # imresize steps of process_image (dataset_.py:481-495), run on the device (vl_resize_u8, PIL bilinear bit for bit):
        # `raw_resize`: stored frame -> raw_image_shape, then crop; `resize`: -> the network input size instead of a crop
        self.stored_image_shape = tuple(self.raw_image_shape)
        self.resize_chain = []
        if defs.imgproc.raw_resize in self.imgproc:
            it = tfrecord.tf_record_iterator(self.record_path)
            try:
                img, _ = tfrecord.parse_frame_example(next(it))
            finally:
                it.close()
            self.stored_image_shape = tuple(img.shape)
            self.resize_chain.append((tuple(img.shape[:2]), tuple(self.raw_image_shape[:2])))
        if defs.imgproc.resize in self.imgproc and self.crop_mode is None:
            self.resize_chain.append((tuple(self.raw_image_shape[:2]), tuple(self.desired_image_shape[:2])))
        elif self.crop_mode is None and tuple(self.raw_image_shape) != tuple(self.desired_image_shape):
            error("Encountered image shape %s but desired shape is %s" % (self.raw_image_shape, self.desired_image_shape))
        if self.crop_mode:
            self.crop_h, self.crop_w = self.compute_crop(self.raw_image_shape, self.desired_image_shape, self.crop_mode)

    def calculate_batches(self, batch_size, input_mode):
        """dataset_.py:582-613."""
        self.batch_size, self.input_mode = batch_size, input_mode
        if not os.path.exists(self.path):
            error("Dataset path does not exist: %s" % self.path)
        self.read_frames_metadata()
        self.record_path = self.path + ".tfrecord"
        if not os.path.exists(self.record_path):
            error("TFRecord file path does not exist: %s" % self.record_path)
        self.reset_iterator()
        self.get_input_data_count()
        self.initialize_imgproc()
        whole = self.num_items // self.batch_size
        left = self.num_items - whole * self.batch_size
        self.batches = [self.batch_size] * whole + ([left] if left else [])
        self.tell()

    def tell(self):
        clips = sum(self.clips_per_video)
        info("Dataset batch information per epoch: items %d, clips %d, frames %d, b-size %d, b-num %d, b-index %d, imgprc %s" %
             (self.num_items, clips, clips * self.num_frames_per_clip, self.batch_size, len(self.batches), self.batch_index,
              defs.imgproc.to_str(self.imgproc)))

    def compute_dataset_portion(self, freq_per_epoch, epochs):
        """dataset_.py:562-568."""
        save_interval = math.ceil(len(self.batches) / freq_per_epoch)
        num_saves = math.ceil(freq_per_epoch * epochs)
        info("Computed batch save interval (from %2.4f per %d-batched epoch) to %d batches and %d total saves" %
             (freq_per_epoch, len(self.batches), save_interval, num_saves))
        return save_interval, num_saves

    # ---- iteration ---------------------------------------------------------------------------------
    def reset_iterator(self):
        self.offset = 0
        if getattr(self, "_vec_iter", None) is not None:
            self._vec_iter.close()
        self._vec_iter = None

    def _read_vectors(self, count):
        """deserialize_vector (dataset_.py:137-168): the next `count` vector records -> (float32 [count, dim], labels per record)."""
        if self._vec_iter is None:
            self._vec_iter = tfrecord.tf_record_iterator(self.record_path)
        vecs, labels = [], []
        for _ in range(count):
            try:
                payload = next(self._vec_iter)
            except StopIteration:
                break
            try:
                v, l = tfrecord.parse_vector_example(payload)
            except Exception as ex:
                warning(str(ex))
                error("Error reading tfrecord vector.")
            vecs.append(v)
            labels.append(l)
        if len(vecs) != count:
            error("End of %s after %d of %d vector records" % (self.record_path, len(vecs), count))
        return np.stack(vecs).astype(np.float32), labels

    def vector_dim(self):
        """Width of a vectors dataset (the `dimension` feature of its first record)."""
        it = tfrecord.tf_record_iterator(self.record_path)
        try:
            v, _ = tfrecord.parse_vector_example(next(it))
        finally:
            it.close()
        return int(v.size)

    def rewind(self):
        self.reset_iterator()
        self.batch_index = 0

    def loop(self):
        return self.batch_index < len(self.batches)

    def _read(self, count, out=None):
        """deserialize_from_tfrecord (dataset_.py:171-217): on EOF mid-batch the reference rewinds and re-reads from
        the start of the file (reread_serialized, 219-230); parse errors are retried read_tries times."""
        tries = 0
        while True:
            try:
                imgs, labels, self.offset = _hostio.read_frames(self.record_path, self.offset, count, self.stored_image_shape,
                                                                out=None if out is None else out[:count])
                return imgs, labels
            except EOFError as ex:
                warning("Unexpected EOF after %d/%d records of the batch; rewinding the iterator" % (ex.records_read, count))
                self.reset_iterator()
                tries += 1
                if tries > max(1, self.read_tries):
                    error("Failed to troubleshoot serialization error.")
            except _hostio.HostIOError as ex:
                tries += 1
                warning("Encountered exception while reading TFRecord batch (%s); try %d" % (ex, tries))
                if tries > self.read_tries:
                    error("Failed to troubleshoot serialization error.")

    def frames_in_batch(self, batch_index):
        """Upper bound of the frames one get_next_batch() call returns (a rank's shard is never larger than the batch)."""
        v0 = batch_index * self.batch_size
        return sum(self.num_frames_per_clip * c for c in self.clips_per_video[v0:v0 + self.batch_size])

    def set_shard(self, rank, world):
        """Data parallel: this process trains on videos shard_range(len(batch), rank, world) of every global batch."""
        self.shard = (int(rank), int(world))

    def get_next_batch(self, out=None):
        """get_next_batch_video_tfr (dataset_.py:386-420): batch = batch_size videos; reads sum(cpv)*fpc consecutive
        frame records; one label per clip (its first frame's).  Returns (frames uint8 [n,H,W,C] raw, crop_y, crop_x,
        mirror, onehot int32 [clips, classes]).  With a shard set, only this rank's videos of the batch are read (the records
        of the other ranks are skipped by their length headers) and `self.global_clips` holds the clip count of the WHOLE batch
        (the loss is its mean, train.py:123)."""
        v0 = self.batch_index * self.batch_size
        cpv_all = self.clips_per_video[v0:v0 + self.batch_size]
        fpc = self.num_frames_per_clip
        n_all = sum(fpc * c for c in cpv_all)
        if not n_all:
            error("Computed 0 frames in next batch.")
        rank, world = getattr(self, "shard", (0, 1))
        base, rem = divmod(len(cpv_all), world)
        lo = rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        cpv = cpv_all[lo:hi]
        before, n = sum(fpc * c for c in cpv_all[:lo]), sum(fpc * c for c in cpv)
        after = n_all - before - n
        self.global_clips = sum(cpv_all)
        if self.input_mode == defs.input_mode.vectors:
            if self._vec_iter is None:
                self._vec_iter = tfrecord.tf_record_iterator(self.record_path)
            self._vec_iter.skip(before)
            vecs, labels_per_frame = self._read_vectors(n) if n else (np.zeros((0, 1), np.float32), [])
            self._vec_iter.skip(after)
            labels, first = [], 0
            for c in cpv:
                labels.extend([labels_per_frame[first]] * c)            # one label set per clip: its first record's (dataset_.py:400-408)
                first += c * fpc
            # per-record targets as well (word-level cross-entropy of a per-step model: every record's own label)
            self.record_onehot = labels_to_one_hot(labels_per_frame, self.num_classes) if labels_per_frame else \
                np.zeros((0, self.num_classes), np.int32)
            ground_truth = labels_to_one_hot(labels, self.num_classes) if labels else np.zeros((0, self.num_classes), np.int32)
            self.batch_index += 1
            return vecs, None, None, None, ground_truth
        if before:
            self.offset = _hostio.skip_records(self.record_path, self.offset, before)
        if n:
            frames, labels_per_frame = self._read(n, out)      # out: caller's (pinned) uint8 buffer [>= n, H, W, C]
        else:                                                  # fewer videos than ranks in a short last batch
            frames, labels_per_frame = np.empty((0,) + tuple(self.stored_image_shape), np.uint8), []
        if after:
            self.offset = _hostio.skip_records(self.record_path, self.offset, after)
        labels, first = [], 0
        for c in cpv:
            labels.extend([labels_per_frame[first]] * c)
            first += c * fpc
        # per-frame draws, like process_image per frame (dataset_.py:481-501), for EVERY frame of the global batch in order, so
        # that N ranks with one seed apply exactly the augmentation one process would
        cy = np.zeros(n_all, np.int32)
        cx = np.zeros(n_all, np.int32)
        mirror = np.zeros(n_all, np.uint8)
        for i in range(n_all):
            if self.crop_mode == defs.imgproc.rand_crop:
                cy[i], cx[i] = self.rng.choice(self.crop_h), self.rng.choice(self.crop_w)
            elif self.crop_mode == defs.imgproc.center_crop:
                cy[i], cx[i] = self.crop_h, self.crop_w
            if defs.imgproc.rand_mirror in self.imgproc:
                mirror[i] = 0 if self.rng.randrange(2) else 1          # `if not randrange(2)` mirrors
        cy, cx, mirror = cy[before:before + n].copy(), cx[before:before + n].copy(), mirror[before:before + n].copy()
        ground_truth = labels_to_one_hot(labels, self.num_classes) if labels else np.zeros((0, self.num_classes), np.int32)
        self.batch_index += 1
        return frames, cy, cx, mirror, ground_truth

    # ---- resume ------------------------------------------------------------------------------------
    def restore(self, batch_index, epoch_index):
        """dataset_.py:534-538."""
        self.batch_index, self.epoch_index = batch_index, epoch_index
        self.fast_forward_iter()

    def fast_forward_iter(self):
        """dataset_.py:772-811: skip the records of the first batch_index batches (variable cpv aware)."""
        if len(self.batches) <= self.batch_index:
            info("Fast-forward not necessary for batch index %d with a total of %d batches." % (self.batch_index + 1, len(self.batches)))
            return
        item_index = self.batch_index * self.batch_size
        num_forward = sum(self.clips_per_video[:item_index]) * self.num_frames_per_clip
        info("Fast forwarding to batch # %d/%d ( image # %d )" % (self.batch_index + 1, len(self.batches), num_forward + 1))
        if self.input_mode == defs.input_mode.vectors:
            self.reset_iterator()
            self._vec_iter = tfrecord.tf_record_iterator(self.record_path)
            self._vec_iter.skip(num_forward)
            return
        self.offset = _hostio.skip_records(self.record_path, 0, num_forward)
