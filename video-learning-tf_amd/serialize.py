"""Writer side of the input format: the reference's serializer (serialize.py) -- `python serialize.py <config.yml>` over the
`serialize:` block: paths file -> clip / frame selection per video (rand_frames | rand_clips | iterative, serialize.py:269-408) with
the generation_error policies (abort | compromise | report), clips-per-video consistency check (:586-595), optional shuffling with
its `.shuffled` / `.unshuffled` side files and the `.<cpv>.cpv.<fpc>.fpc.<mode>.cfm` clip listing (:764-822), threaded frame reads
(:153-244), TFRecord + `.size` writing (:138-151,246-266) and the read-back validator (:677-762).  Frames are read with Pillow --
what scipy.misc.imread / imresize of the reference are -- converted RGB -> BGR and resized to raw_image_shape (:411-425).
`write_video_dataset` / `write_vector_dataset` write the same files from in-memory arrays (tests, examples)."""
import itertools
import os
import random
import shutil
import threading
import time

import numpy as np

from . import tfrecord
from .defs_ import defs
from .parse_opts import parse_seq
from .utils_ import CustomLogger, debug, error, get_datetime_str, info, warning


class SerializationSettings:
    """serialize.py:20-117: the `serialize:` YAML block."""
    init_file = None
    run_id = None
    input_files = []
    path_prepend_folder = None
    output_folder = None
    num_threads = 4
    num_items_per_thread = 500
    num_frames_per_clip = 16
    raw_image_shape = (240, 320, 3)
    clipframe_mode = defs.clipframe_mode.rand_clips
    clip_offset_or_num = 1
    generation_error = defs.generation_error.abort
    frame_format = "jpg"
    do_shuffle = False
    do_serialize = False
    do_validate = True
    validate_pcnt = 10
    seed = None

    def __init__(self):
        self.generation_log = []           # (message, path) of every video that could not supply the requested clips

    def initialize_from_file(self, init_file, configure_logging=True):
        import yaml
        self.init_file = init_file
        if not os.path.exists(init_file):
            error("Initialization file [%s] does not exist" % init_file)
        if init_file.endswith(".ini"):
            error("Ini files deprecated")
        if ".yml" not in init_file:
            error("Need a yml initialization file")
        with open(init_file, "r") as f:
            config = yaml.safe_load(f)["serialize"]
        self.output_folder = config.get("output_folder")
        self.path_prepend_folder = config.get("path_prepend_folder")
        self.input_files = [x.strip() for x in parse_seq(config["input_files"])]
        self.run_id = (config.get("run_id") or "").strip()
        self.num_threads = int(config["num_threads"])
        self.num_items_per_thread = int(config["num_items_per_thread"])
        self.raw_image_shape = tuple(parse_seq(config["raw_image_shape"])) if config.get("raw_image_shape") not in (None, "None") else None
        self.clip_offset_or_num = int(config["clip_offset_or_num"])
        self.num_frames_per_clip = int(config["num_frames_per_clip"])
        self.clipframe_mode = defs.check(config["clipframe_mode"], defs.clipframe_mode)
        self.generation_error = defs.check(config["generation_error"], defs.generation_error)
        self.do_shuffle, self.do_serialize, self.do_validate = bool(config["do_shuffle"]), bool(config["do_serialize"]), bool(config["do_validate"])
        self.frame_format = config["frame_format"].strip()
        level = (config.get("logging_level") or "logging.INFO").strip()
        if level not in ["logging." + x for x in ("INFO", "DEBUG", "WARN")]:
            error("Invalid logging level: [%s]" % level)
        if not self.run_id:
            self.run_id = "serialize_%s" % get_datetime_str()
        self.logfile = "log_" + self.run_id + ".log"
        if configure_logging:
            if self.output_folder:
                os.makedirs(self.output_folder, exist_ok=True)
            self.logger = CustomLogger()
            self.logger.configure_logging(os.path.join(self.output_folder or ".", self.logfile) if self.output_folder else self.logfile, level)
        if "seed" in config and config["seed"] is not None:
            try:
                self.seed = float(config["seed"])
            except (TypeError, ValueError):
                error("Invalid seed value: %s - numeric expected" % str(config["seed"]))
            info("Using supplied seed: %f" % self.seed)
        else:
            self.seed = random.random()
            info("Using randomized seed: %f" % self.seed)
        random.seed(self.seed)
        info("Starting serialization run: [%s]" % self.run_id)
        return self


# ---- clip / frame selection per video (serialize.py:269-408) ----------------------------------------------------------------------
# The three clipframe modes share two stages.  Stage 1: a video with fewer frames than one clip (num_frames_per_clip) meets the
# `generation_error` policy -- one helper for all modes.  Stage 2, per mode: where the clips start.
def _shortfall(settings, path, message):
    """The generation_error policy for a video that cannot supply what was asked (serialize.py:277-289,309-335,366-376).
    abort: raise.  Otherwise the problem is logged in settings.generation_log and the caller is told whether to PAD the request
    (compromise -> True) or to DROP the video (report -> False)."""
    policy = settings.generation_error
    if policy == defs.generation_error.abort:
        error(message)
    if policy not in (defs.generation_error.compromise, defs.generation_error.report):
        error("Undefined generation error strategy: %s" % policy)
    settings.generation_log.append((message, path))
    return policy == defs.generation_error.compromise


def _pad_with_random_frames(frames, missing):
    return frames + [random.choice(frames) for _ in range(missing)]


def _pad_with_start_frame(frames, missing):
    return [0] * missing + frames                       # rand_clips duplicates the first frame up to the clip length


#                     message of stage 1 (clip length, video, frames it has)                                        padding
_TOO_SHORT = {
    defs.clipframe_mode.rand_frames: ("Attempted to get a %d-framed clip from video %s which has %d frames.", _pad_with_random_frames),
    defs.clipframe_mode.rand_clips: ("Video %s cannot sustain a number of %d fpc, as it has %d frames", _pad_with_start_frame),
    defs.clipframe_mode.iterative: ("Attempted to get %d-framed sequential clips from video %s which has %d frames.", _pad_with_random_frames),
}


def _fit_to_clip_length(frames, mode, settings, path):
    """Stage 1.  -> (frames to continue with, padded?) or (None, False) when the policy drops the video."""
    fpc, n = settings.num_frames_per_clip, len(frames)
    if fpc - n <= 0:
        return frames, False
    text, pad = _TOO_SHORT[mode]
    name = os.path.basename(path)
    message = text % ((name, fpc, n) if mode == defs.clipframe_mode.rand_clips else (fpc, name, n))
    if not frames or not _shortfall(settings, path, message):
        return None, False
    return pad(frames, fpc - n), True


def _spread_random_starts(possible, want, fpc):
    """rand_clips, stage 2 (serialize.py:337-357): `want` random starts; a chosen start takes the starts within one clip length of
    it out of the pool, and the pool refills when it runs dry."""
    starts, pool = [], list(possible)
    for _ in range(want):
        st = random.choice(pool)
        starts.append(st)
        for i in range(st - fpc + 1, st + fpc):          # ONE occurrence each: a pool padded with repeated starts keeps the others
            if i in pool:
                pool.remove(i)
        if not pool:
            pool = list(possible)
    return starts


def select_clips(avail_frame_idxs, settings, path, mode=None):
    """Frame indices per clip of one video under settings.clipframe_mode (serialize.py:269-378).
      rand_frames  ONE clip of num_frames_per_clip frames drawn at random.  (The reference assigns the None that random.shuffle returns
                   and hands back a flat index list generate_frames_for_video cannot iterate as clips; the evident intent is built.)
      rand_clips   clip_offset_or_num clips of consecutive frames at random, spread starts
      iterative    every clip of consecutive frames whose starts are num_frames_per_clip + clip_offset_or_num frames apart"""
    mode = mode or settings.clipframe_mode
    if mode not in _TOO_SHORT:
        error("Undefined clipframe mode [%s]" % mode)
    frames = list(avail_frame_idxs)
    n, fpc, k = len(frames), settings.num_frames_per_clip, settings.clip_offset_or_num
    if mode == defs.clipframe_mode.rand_clips and n == 0:
        error("No frames for path [%s]" % path)
    if mode == defs.clipframe_mode.rand_frames:
        random.shuffle(frames)
    fitted, padded = _fit_to_clip_length(frames, mode, settings, path)
    if mode == defs.clipframe_mode.rand_frames:
        return [] if fitted is None else [fitted[:fpc]]
    if mode == defs.clipframe_mode.iterative:
        # (like the reference, the start range uses the ORIGINAL frame count: a compromised short video yields no clip here and is
        # caught by check_cpv_per_item)
        return [] if fitted is None else [list(range(st, st + fpc)) for st in range(0, n - fpc + 1, fpc + k)]
    if padded:                                           # rand_clips on a padded video: the one possible clip, k times
        return [list(fitted) for _ in range(k)]
    possible = list(range(n - fpc + 1))                  # (a dropped video goes on: with no start left it is reported a second time)
    if k - len(possible) > 0:
        message = "Video %s cannot sustain a number of %d cpv as it has %d frames" % (os.path.basename(path), k, n)
        if not _shortfall(settings, path, message):
            return []
        possible = _pad_with_random_frames(possible, k - len(possible))
    return [list(range(st, st + fpc)) for st in _spread_random_starts(possible, k, fpc)]


def get_random_frames(avail_frame_idxs, settings, path):
    return select_clips(avail_frame_idxs, settings, path, defs.clipframe_mode.rand_frames)


def get_random_clips(avail_frame_idxs, settings, path):
    return select_clips(avail_frame_idxs, settings, path, defs.clipframe_mode.rand_clips)


def get_sequential_clips(avail_frame_idxs, settings, path):
    return select_clips(avail_frame_idxs, settings, path, defs.clipframe_mode.iterative)


def generate_frames_for_video(path, settings):
    """serialize.py:381-408: the sorted frame files of the folder `path` -> frame paths per clip."""
    files = sorted(f for f in os.listdir(path) if os.path.isfile(os.path.join(path, f)))
    return [[os.path.join(path, files[i]) for i in clip] for clip in select_clips(range(len(files)), settings, path)]


def check_cpv_per_item(paths_per_item, items_list, settings):
    """serialize.py:586-595: every video must have produced exactly clip_offset_or_num clips."""
    erratic = [i for i, p in enumerate(paths_per_item) if len(p) != settings.clip_offset_or_num]
    if erratic:
        for e in erratic:
            warning("Item %d/%d : %s has cpv of len %d:" % (e + 1, len(paths_per_item), items_list[e], len(paths_per_item[e])))
            for p in paths_per_item[e]:
                warning(str(p))
        error("Erratic item(s) encountered")


# ---- reading ----------------------------------------------------------------------------------------------------------------------
def read_image(imagepath, settings):
    """serialize.py:411-434: imread -> 3 channels -> BGR -> imresize to raw_image_shape (PIL bilinear on uint8)."""
    from PIL import Image
    try:
        img = Image.open(imagepath)
        image = np.asarray(img)
        if image.ndim <= 2:
            image = np.repeat(image[:, :, np.newaxis], 3, 2)
        image = np.ascontiguousarray(image[:, :, :3][:, :, ::-1])
        if settings.raw_image_shape is not None:
            h, w = settings.raw_image_shape[:2]
            image = np.asarray(Image.fromarray(image).resize((w, h), resample=Image.BILINEAR))
        return image
    except Exception as ex:
        warning("Error reading image %s: %s" % (imagepath, ex))
        return None


def read_file(inp, settings):
    """serialize.py:513-558: 'path label...' per line -> (paths, labels, mode, max_num_labels); a first token without letters
    means a vectors file, a frame-format suffix image mode, anything else video mode."""
    import string
    mode, max_num_labels, paths, labels = None, -1, [], []
    with open(inp, "r") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            path, label = line.split(" ", 1)
            if not any(x in string.ascii_letters for x in path):
                return [], [], defs.input_mode.vectors, max_num_labels
            label = [int(l) for l in label.split()]
            max_num_labels = max(max_num_labels, len(label))
            if mode is None:
                mode = defs.input_mode.image if path.lower().endswith("." + settings.frame_format.lower()) else defs.input_mode.video
            if settings.path_prepend_folder is not None:
                path = os.path.join(settings.path_prepend_folder, path)
            paths.append(path)
            labels.append(label)
    return paths, labels, mode, max_num_labels


def read_vectors(input_file):
    """serialize.py:824-847: lines '<v0,v1,...> <label[,label...]>' -> (float32 [n, dim], labels, max_num_labels)."""
    vectors, labels, max_num_labels, dim0 = [], [], 1, None
    with open(input_file) as f:
        for i, line in enumerate(f):
            parts = line.split()
            if not parts:
                continue
            row = np.asarray(parts[0].split(","), np.float32)
            lab = [int(x) for x in parts[-1].split(",")]
            if dim0 is None:
                dim0 = len(row)
            if len(row) != dim0:
                error("Inconsistent dimension: Encountered dim: %d at line %d, had stored %d." % (len(row), i + 1, dim0))
            vectors.append(row)
            labels.append(lab[0] if len(lab) == 1 else lab)
            max_num_labels = max(max_num_labels, len(lab))
    return np.stack(vectors), labels, max_num_labels


def _sublists(lst, n):
    return [lst[i:i + n] for i in range(0, len(lst), n)]


def serialize_multithread(item_paths, clips_per_item, frame_paths, labels, outfile, mode, max_num_labels, settings):
    """serialize.py:153-220: `.size` first, then runs of num_threads reader threads of num_items_per_thread frames each;
    every run is written in thread order, so the record order is the path order."""
    fpc = None if mode == defs.input_mode.image else settings.num_frames_per_clip
    tfrecord.write_size_file(outfile + ".size", len(item_paths), mode, clips_per_item, fpc, max_num_labels)
    per_run = settings.num_items_per_thread * settings.num_threads
    tic, count = time.time(), 0
    with tfrecord.TFRecordWriter(outfile) as writer:
        for run_paths, run_labels in zip(_sublists(frame_paths, per_run), _sublists(labels, per_run)):
            chunks = _sublists(run_paths, settings.num_items_per_thread)
            lchunks = _sublists(run_labels, settings.num_items_per_thread)
            frames = [[] for _ in chunks]

            def work(t):
                for fp in chunks[t]:
                    im = read_image(fp, settings)
                    if im is None:
                        frames[t] = []
                        return
                    frames[t].append(im)
            threads = [threading.Thread(target=work, args=(t,)) for t in range(len(chunks))]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            for t in range(len(chunks)):
                if len(frames[t]) != len(chunks[t]):
                    error("Thread # %d encountered an error." % t)
                for im, lab in zip(frames[t], lchunks[t]):
                    writer.write(tfrecord.frame_example(im, lab))
                count += len(frames[t])
    info("Serialized %d frames to %s in %.1f s" % (count, outfile, time.time() - tic))


def serialize_ascii(input_file, settings):
    """serialize.py:849-884: a vectors file (+ `.ids`) -> TFRecord of vector records, one item (1 clip) per vector."""
    outfile = input_file + ".tfrecord"
    if settings.output_folder:
        os.makedirs(settings.output_folder, exist_ok=True)
        outfile = os.path.join(settings.output_folder, os.path.basename(outfile))
    vectors, labels, max_num_labels = read_vectors(input_file)
    ids_file = input_file + ".ids"
    ids = [l.split()[0] for l in open(ids_file).read().splitlines() if l.strip()] if os.path.exists(ids_file) else [str(i) for i in range(len(vectors))]
    shuffle_idx = None
    if settings.do_shuffle:
        shuffle_idx = list(range(len(vectors)))
        random.shuffle(shuffle_idx)
        vectors, labels, ids = vectors[shuffle_idx], [labels[i] for i in shuffle_idx], [ids[i] for i in shuffle_idx]
    tfrecord.write_size_file(outfile + ".size", len(vectors), defs.input_mode.vectors, [1] * len(vectors), settings.num_frames_per_clip,
                             max_num_labels)
    with tfrecord.TFRecordWriter(outfile) as w:
        for v, l in zip(vectors, labels):
            w.write(tfrecord.vector_example(v, l))
    return (input_file, shuffle_idx), ids, labels, outfile


def _shuffle_together(*lists):
    z = list(zip(*lists))
    random.shuffle(z)
    return [list(x) for x in zip(*z)] if z else [list(l) for l in lists]


def _output_path(inp, settings, suffix=""):
    out = inp + suffix
    if settings.output_folder is not None:
        os.makedirs(settings.output_folder, exist_ok=True)
        out = os.path.join(settings.output_folder, os.path.basename(out))
    return out


def write_serialization(settings):
    """serialize.py:597-675 -> (written data per input file, generation-error flag per input file)."""
    written, errors = [], [False] * len(settings.input_files)
    for idx, inp in enumerate(settings.input_files):
        info("Reading input file %d/%d: [%s] " % (idx + 1, len(settings.input_files), inp))
        item_paths, item_labels, mode, max_num_labels = read_file(inp, settings)
        if mode == defs.input_mode.vectors:
            sidx, ids, labels, _ = serialize_ascii(inp, settings)
            written.append([sidx, labels, ids, None, mode])
            continue
        clips_per_item = None
        if mode == defs.input_mode.image:
            if settings.do_shuffle:
                item_paths, item_labels = _shuffle_together(item_paths, item_labels)
            paths_to_serialize, labels_to_serialize = item_paths, item_labels
            written.append([item_paths, item_labels, None, None, mode])
        elif mode == defs.input_mode.video:
            settings.generation_log = []
            paths = [generate_frames_for_video(v, settings) for v in item_paths]
            if settings.generation_log:
                errors[idx] = True
                warning("%d generation errors occured, that were resolved with the [%s] strategy:" % (len(settings.generation_log), settings.generation_error))
                for i, (msg, _) in enumerate(settings.generation_log):
                    warning("%d/%d: %s" % (i + 1, len(settings.generation_log), msg))
                if settings.generation_error == defs.generation_error.report:
                    probl = _output_path("generation_errors_files_%s_%s" % (settings.run_id, get_datetime_str()), settings)
                    with open(probl, "w") as f:
                        for _, pf in settings.generation_log:
                            f.write(pf + "\n")
                    info("Writing problematic files in %s" % probl)
                    info("Omitting serialization due to generation error setting [%s]." % defs.generation_error.report)
                    settings.generation_log = []
                    written.append(None)
                    continue
                if settings.generation_error == defs.generation_error.compromise:
                    settings.generation_log, errors[idx] = [], False
                else:
                    error("Generated paths with errors, but error strategy is [%s]" % settings.generation_error)
            if settings.clipframe_mode != defs.clipframe_mode.iterative:       # iterative yields as many clips as fit (cpv varies)
                check_cpv_per_item(paths, item_paths, settings)
            if settings.do_shuffle:
                item_paths, paths, item_labels = _shuffle_together(item_paths, paths, item_labels)
                for vid in paths:
                    if settings.clipframe_mode == defs.clipframe_mode.rand_frames:
                        for clip in vid:
                            random.shuffle(clip)
                    else:
                        random.shuffle(vid)
            clips_per_item = [len(v) for v in paths]
            labels_to_serialize = [item_labels[i] for i in range(len(item_labels)) for clip in paths[i] for _ in clip]
            paths_to_serialize = [p for video in paths for clip in video for p in clip]
            written.append([item_paths, item_labels, paths_to_serialize, labels_to_serialize, mode])
        else:
            error("Unknown data type: %s" % mode)
        if settings.do_serialize:
            out = _output_path(inp, settings, ".tfrecord")
            info("Serializing to %s " % out)
            serialize_multithread(item_paths, clips_per_item, paths_to_serialize, labels_to_serialize, out, mode, max_num_labels, settings)
        info("Done processing input file %s" % inp)
    return written, errors


def write_paths_file(data, errors, settings):
    """serialize.py:764-822: `.shuffled` / `.unshuffled` copies of the paths file and the clip listing
    `<paths>[.<cpv>.cpv].<fpc>.fpc.<mode>.cfm` with one 'frame-path labels' line per serialized frame."""
    for i, inp in enumerate(settings.input_files):
        if errors[i] or data[i] is None:
            info("Skipping file %s due to generation errors and strategy [%s]" % (os.path.basename(inp), settings.generation_error))
            continue
        item_paths, item_labels, paths, labels, mode = data[i]
        output_file = _output_path(inp, settings)
        if settings.do_shuffle:
            if mode == defs.input_mode.vectors:
                with open(output_file + ".shuffled", "w") as f:
                    for item_id, label in zip(paths, item_labels):
                        f.write("%s %s\n" % (item_id, str(label)))
            else:
                shutil.copyfile(inp, output_file + ".unshuffled")
                with open(output_file + ".shuffled", "w") as f:
                    for item, lab in zip(item_paths, item_labels):
                        f.write("%s %s\n" % (item, " ".join("%d" % l for l in (lab if isinstance(lab, list) else [lab]))))
        elif settings.output_folder is not None and os.path.abspath(inp) != os.path.abspath(output_file):
            shutil.copyfile(inp, output_file)
        if mode != defs.input_mode.video:
            continue
        clip_info = "" if settings.clipframe_mode == defs.clipframe_mode.rand_frames else ".%d.cpv" % settings.clip_offset_or_num
        outfile = "%s%s.%d.fpc.%s.cfm" % (output_file, clip_info, settings.num_frames_per_clip, settings.clipframe_mode)
        with open(outfile, "w") as f:
            for path, label in zip(paths, labels):
                f.write("%s %s\n" % (path, " ".join(map(str, label))))


def validate(written_data, errors, settings):
    """serialize.py:677-762: re-reads the TFRecord and compares sampled records (all of them below 10,000) with a fresh read of
    the source frame / vector and its label."""
    for index, inp in enumerate(settings.input_files):
        if errors[index] or written_data[index] is None:
            info("Skipping file %s due to generation errors and strategy [%s]" % (os.path.basename(inp), settings.generation_error))
            continue
        output_file = _output_path(inp, settings, ".tfrecord")
        if not os.path.isfile(output_file):
            error("TFRecord file %s does not exist." % output_file)
        item_paths, item_labels, paths, labels, mode = written_data[index]
        if mode == defs.input_mode.video and not settings.do_serialize:
            error("Cannot validate-only in video mode, as frame selection is not known.")
        if settings.do_shuffle and not settings.do_serialize:
            error("Cannot validate-only with shuffle enabled, as serialization shuffling is not known.")
        vectors = None
        if mode == defs.input_mode.image:
            paths, labels = item_paths, item_labels
        if mode == defs.input_mode.vectors:
            _, shuffle_idx = item_paths
            vectors, vlabels, _ = read_vectors(inp)
            if shuffle_idx is not None:
                vectors, vlabels = vectors[shuffle_idx], [vlabels[s] for s in shuffle_idx]
            paths, labels = list(range(len(vectors))), vlabels
        total = len(paths)
        num_validate = round(total * settings.validate_pcnt / 100) if total >= 10000 else total
        info("Will validate %d%% of a total of %d items (but at least 10K), i.e. %d items." % (settings.validate_pcnt, total, num_validate))
        idx_list = list(range(total))
        random.shuffle(idx_list)
        check = set(idx_list[:num_validate])
        error_free, tic = True, time.time()
        it = tfrecord.tf_record_iterator(output_file)
        try:
            for i in range(total):
                payload = next(it)
                if i not in check:
                    continue
                if mode == defs.input_mode.vectors:
                    vec, lab = tfrecord.parse_vector_example(payload)
                    want_lab = labels[i] if isinstance(labels[i], list) else [labels[i]]
                    if not np.array_equal(vec, vectors[i]):
                        warning("Unequal vector @ idx %d" % i)
                        error_free = False
                    if list(lab) != want_lab:
                        warning("Unequal label @ %d. Found %s, expected %s" % (i, lab, want_lab))
                        error_free = False
                else:
                    frame = read_image(paths[i], settings)
                    img, lab = tfrecord.parse_frame_example(payload)
                    if frame is None or not np.array_equal(frame, img):
                        warning("Unequal image @ %s" % paths[i])
                        error_free = False
                    if list(lab) != list(labels[i]):
                        warning("Unequal label @ %s. Found %s, expected %s" % (paths[i], lab, labels[i]))
                        error_free = False
            try:
                next(it)
                warning("%s holds more records than the %d that were written" % (output_file, total))
                error_free = False
            except StopIteration:
                pass
        except StopIteration:
            warning("%s ends before its %d records" % (output_file, total))
            error_free = False
        finally:
            it.close()
        if not error_free:
            error("errors exist.")
        info("Validation for %s completed successfully in %.1f s." % (os.path.basename(inp) + ".tfrecord", time.time() - tic))
    info("Validation completed error-free for all files.")


def main(init_file):
    """serialize.py:887-902: python serialize.py <config.yml>."""
    settings = SerializationSettings().initialize_from_file(init_file)
    written, errors = write_serialization(settings)
    write_paths_file(written, errors, settings)
    if settings.do_validate:
        info("Validating serialization")
        validate(written, errors, settings)
    if settings.output_folder is not None and settings.do_serialize and not any(errors):
        shutil.copyfile(settings.init_file, os.path.join(settings.output_folder, os.path.basename(settings.init_file)))
    info("Serialization complete")
    return written, errors


# ---- the same files from in-memory arrays (tests, examples) ----------------------------------------------------------------------
def write_video_dataset(paths_file, videos, labels, fpc, clips_per_video):
    """videos: list of uint8 arrays [frames, H, W, 3] (BGR); writes <paths_file>, .tfrecord and .tfrecord.size with
    `clips_per_video[i]` clips of `fpc` consecutive frames each (clip j starts at frame j*fpc)."""
    cpv = clips_per_video if isinstance(clips_per_video, (list, tuple)) else [clips_per_video] * len(videos)
    with open(paths_file, "w") as f:
        for i, lab in enumerate(labels):
            lab = lab if isinstance(lab, (list, tuple)) else [lab]
            f.write("video_%04d %s\n" % (i, " ".join(str(l) for l in lab)))
    maxlab = 1
    with tfrecord.TFRecordWriter(paths_file + ".tfrecord") as w:
        for vid, lab, c in zip(videos, labels, cpv):
            lab = lab if isinstance(lab, (list, tuple)) else [lab]
            maxlab = max(maxlab, len(lab))
            if len(vid) < c * fpc:
                error("video has %d frames, needs %d" % (len(vid), c * fpc))
            for k in range(c * fpc):
                w.write(tfrecord.frame_example(vid[k], lab))
    tfrecord.write_size_file(paths_file + ".tfrecord.size", len(videos), defs.input_mode.video, list(cpv), fpc, maxlab)
    info("Serialized %d videos to %s.tfrecord" % (len(videos), paths_file))


def write_vector_dataset(paths_file, sequences, labels, fpc, clips_per_item=1):
    """Vector-mode dataset (serialize.py:258-266,605-606; `.size` type `vectors`): sequences = list of float32 arrays
    [clips * fpc, dim] (e.g. the word embeddings of a caption), labels = per item either ONE label list (stored with every record,
    like a video's label) or a list of per-record label lists (word-level targets)."""
    cpv = clips_per_item if isinstance(clips_per_item, (list, tuple)) else [clips_per_item] * len(sequences)
    maxlab = 1
    with open(paths_file, "w") as f:
        for i in range(len(sequences)):
            f.write("item_%04d\n" % i)
    with tfrecord.TFRecordWriter(paths_file + ".tfrecord") as w:
        for seq, lab, c in zip(sequences, labels, cpv):
            seq = np.asarray(seq, np.float32)
            if len(seq) != c * fpc:
                error("vector item has %d records, needs %d" % (len(seq), c * fpc))
            per_record = len(lab) == len(seq) and all(isinstance(x, (list, tuple, np.ndarray)) for x in lab)
            for k in range(len(seq)):
                l = list(lab[k]) if per_record else (list(lab) if isinstance(lab, (list, tuple, np.ndarray)) else [lab])
                maxlab = max(maxlab, len(l))
                w.write(tfrecord.vector_example(seq[k], l))
    tfrecord.write_size_file(paths_file + ".tfrecord.size", len(sequences), defs.input_mode.vectors, list(cpv), fpc, maxlab)
    info("Serialized %d vector items to %s.tfrecord" % (len(sequences), paths_file))
