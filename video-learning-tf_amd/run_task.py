"""run_task: `python run_task.py <config.yml>` -- the reference's driver (run_task.py:117-160) over the
MI355X engine.  Same YAML, same TFRecord/.size inputs, same log lines and result files."""
import argparse
import math
import os
import pickle
import time

import numpy as np
import torch

from . import dp as dpmod
from .defs_ import defs
from .engine import LRCNEngine, NetConfig, init_params
from .parse_opts import parse_seq
from .settings_ import Settings
from .train import Train
from .utils_ import elapsed_str, error, info, warning
from .val import Validation


def print_iter_info(settings, feeder, num_images, num_labels, padding):
    """run_task.py:15-21."""
    dataset = feeder.datasets[settings.phase][0]
    epoch_str = "" if settings.val else "epoch: %2d/%2d," % (settings.train.epoch_index + 1, settings.train.epochs)
    info("Mode: [%s], %s batch %4d / %4d : %s images, %3d labels" %
         (settings.phase, epoch_str, feeder.get_batch_index(), len(dataset.batches), str(num_images), num_labels))


def graph_feeds(fdicts, tags, dev):
    """{tag: feed dict of Feeder.get_feed_dict} -> GraphEngine feeds: frames with their imgproc arguments, or the vectors tensor."""
    feeds = {}
    for tag in tags:
        fd = fdicts[tag]
        if "vectors" in fd:
            feeds[tag] = torch.from_numpy(fd["vectors"]).to(dev)
        elif "device" in fd:             # read and uploaded ahead by the feeder's BatchPrefetcher: wait for the copy on the stream
            torch.cuda.current_stream(dev).wait_event(fd["ready"])
            t = fd["device"]
            feeds[tag] = dict(frames_u8=t["frames_u8"], mean_bgr=fd["mean_bgr"], crop_y=t["crop_y"], crop_x=t["crop_x"], mirror=t["mirror"],
                              resize=fd.get("resize"))
        else:
            feeds[tag] = dict(frames_u8=torch.from_numpy(fd["frames_u8"]).to(dev, non_blocking=True), mean_bgr=fd["mean_bgr"],
                              crop_y=torch.from_numpy(fd["crop_y"]).to(dev), crop_x=torch.from_numpy(fd["crop_x"]).to(dev),
                              mirror=torch.from_numpy(fd["mirror"]).to(dev), resize=fd.get("resize"))
    return feeds


def pipeline_net_config(settings, p, dataset):
    """NetConfig of one dcnn pipeline over `dataset` (model.py:81-151)."""
    kw = dict(image_shape=tuple(dataset.get_image_shape()), num_classes=settings.num_classes, fpc=dataset.num_frames_per_clip,
              frame_encoding_layer=p.frame_encoding_layer, classifier=p.classifier or defs.classifier.fc,
              dropout_keep_prob=settings.get_dropout(), optimizer=settings.train.optimizer if settings.train else "sgd",
              conv_math=os.environ.get("VLTF_CONV_MATH", "f32"))     # "bf16x3": opt-in split-bf16 conv products (not a reference key)
    if p.classifier == defs.classifier.lstm:
        if p.frame_fusion and p.frame_fusion[0] != defs.fusion_type.none:
            error("The LSTM classifier should be used only with [none] fusion, but it's [%s]" % p.frame_fusion[0])
        kw.update(lstm_hidden=p.lstm_params[0], lstm_layers=p.lstm_params[1], fusion=p.lstm_params[2])
    else:
        kw.update(frame_fusion=tuple(p.frame_fusion) if p.frame_fusion else None)
    return NetConfig(**kw)


def net_config(settings, dataset):
    """Single-pipeline model: the last pipeline defines the logits (model.py:161)."""
    p = settings.pipelines[settings.pipeline_names[-1]]
    if len(settings.pipeline_names) != 1 or p.representation != defs.representation.dcnn or p.input != [defs.dataset_tag.main]:
        error("A single pipeline must be a dcnn pipeline on the main dataset.")
    return pipeline_net_config(settings, p, dataset), p


def graph_config(settings, feeder, batch):
    """Any list of pipelines (model.py:18-162) -> ([PipelineSpec], {tag: DatasetInfo}, {tag: Dataset}) for vltf_amd.graph.GraphEngine.
    batch: videos per batch on this rank; a dataset's clips per batch follow from its clips-per-video list."""
    from .graph import DatasetInfo, PipelineSpec
    specs = []
    for name in settings.pipeline_names:
        p = settings.pipelines[name]
        specs.append(PipelineSpec(name=name, input=list(p.input), representation=p.representation,
                                  frame_encoding_layer=p.frame_encoding_layer, fc_output_dim=getattr(p, "fc_output_dim", None),
                                  classifier=p.classifier, lstm_params=tuple(p.lstm_params) if p.lstm_params else None,
                                  frame_fusion=tuple(p.frame_fusion) if p.frame_fusion else None, input_fusion=p.input_fusion))
    infos, dsets = {}, {}
    for tag in sorted({i for sp in specs for i in sp.input if i not in settings.pipelines}):
        found = feeder.get_dataset_by_tag(tag)
        if len(found) != 1:
            error("%d datasets satisfy the network input requirement [%s], but exactly one must." % (len(found), tag))
        d = found[0]
        cpv = d.clips_per_video
        if not all(c == cpv[0] for c in cpv):
            warning("Non equal clips per item")                        # model.py:57-58
        if d.input_mode == defs.input_mode.vectors:
            infos[tag] = DatasetInfo("vectors", d.num_frames_per_clip, cpv[0], batch * max(cpv), dim=d.vector_dim())
        else:
            infos[tag] = DatasetInfo("video", d.num_frames_per_clip, cpv[0], batch * max(cpv), image_shape=tuple(d.get_image_shape()))
        dsets[tag] = d
    # `input_shape` (model.py:47-54): the placeholder's shape, when given, "has to be the same as the image shape in the dataset
    # configuration" -- a different one would be fed frames it cannot hold
    for sp in specs:
        shapes = getattr(settings.pipelines[sp.name], "input_shape", None) or []
        for src, shp in zip(sp.input, shapes):
            if shp in (None, "None") or src not in dsets:
                continue
            want = tuple(parse_seq(shp)) if isinstance(shp, str) else tuple(shp)
            have = tuple(dsets[src].get_image_shape()) if infos[src].mode == "video" else (infos[src].dim,)
            if tuple(int(v) for v in want) != have:
                error("Pipeline [%s]: input_shape %s of input [%s] differs from the dataset's %s" % (sp.name, want, src, have))
    n_items = {d.num_items for d in dsets.values()}
    if len(n_items) > 1:
        error("The datasets of the pipelines do not pair up item by item (%s items)." % sorted(n_items))
    return specs, infos, dsets


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Unpickler for the reference's weights file: admits the reconstruction of numpy arrays / dtypes / scalars and nothing
    else -- no other global can be looked up, so no code from the file can run."""
    ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
               ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}

    @staticmethod
    def _latin1(text, encoding="latin1"):
        """Stand-in for _codecs.encode, which protocol-2 pickles written by Python 3 use to carry array bytes as latin-1 text."""
        if encoding not in ("latin1", "latin-1"):
            raise pickle.UnpicklingError("weights file asks for the codec %r: only latin1 byte strings are admitted" % (encoding,))
        return text.encode("latin1")

    def find_class(self, module, name):
        if (module, name) == ("_codecs", "encode"):
            return self._latin1
        if (module, name) in self.ALLOWED:
            import importlib
            mod = importlib.import_module("numpy._core.multiarray" if module.endswith("multiarray") else module)
            return getattr(mod, name)
        raise pickle.UnpicklingError("weights file refers to %s.%s: only numpy arrays are admitted" % (module, name))


def load_weights_file(path):
    """alexnet.py:50-52,69-71: `numpy.load(weights_file, encoding="latin1").item()` -- the bvlc_alexnet.npy layout, a pickled
    dict {layer: [W, b]} (conv W in HWIO, fc W [in, out]) inside an object-dtype .npy -- -> {tf variable name: array}.
    The pickle stream is read by an unpickler that can only rebuild numpy arrays (_ArraysOnlyUnpickler).  An .npz of arrays named
    like the TF variables (dcnn/conv1W, ...) loads without any pickle."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    with open(path, "rb") as f:
        try:
            version = np.lib.format.read_magic(f)
            shape, _, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
        except ValueError as ex:
            error("weights_file %s is not a .npy / .npz file: %s" % (path, ex))
        if dtype.hasobject:
            if shape != ():
                error("weights_file %s: expected a 0-d object array holding the {layer: [W, b]} dict" % path)
            try:
                net_data = _ArraysOnlyUnpickler(f, encoding="latin1").load()
            except pickle.UnpicklingError as ex:
                error("weights_file %s refused: %s" % (path, ex))
        else:
            error("weights_file %s holds a plain array, not the {layer: [W, b]} dict" % path)
    if isinstance(net_data, np.ndarray) and net_data.shape == () and net_data.dtype == object:
        net_data = net_data.item()                               # numpy pickles the 0-d object array itself; `.item()` as alexnet.py:51
    if not isinstance(net_data, dict):
        error("weights_file %s: expected a dict {layer: [W, b]}, found %s" % (path, type(net_data).__name__))
    out = {}
    for layer, wb in net_data.items():
        layer = layer.decode("latin1") if isinstance(layer, bytes) else str(layer)
        if not isinstance(wb, (list, tuple)) or len(wb) != 2:
            error("weights_file %s: entry [%s] is not a [W, b] pair" % (path, layer))
        out["dcnn/%sW" % layer] = np.asarray(wb[0], np.float32)
        out["dcnn/%sb" % layer] = np.asarray(wb[1], np.float32)
    return out


def do_train(settings, train, feeder, engine):
    """run_task.py:25-81."""
    run_batch_count, min_train_loss = 0, (1000, -1)
    fed_clips, fed_time = 0, 0.0          # feed + step time of every batch but the first (checkpoint writes excluded)
    info("Starting train")
    for _ in range(settings.train.epoch_index, settings.train.epochs):
        while feeder.loop():
            tic = time.perf_counter()
            fdict, num_data, num_labels, padding = feeder.get_feed_dict()
            others = None
            if getattr(settings, "graph_tags", None) is not None:      # multi-pipeline model: every other dataset's batch of the same items
                others = {}
                for tag in settings.graph_tags:
                    if tag != defs.dataset_tag.main:
                        others[tag], nd2, _, _ = feeder.get_feed_dict(tag)
                        num_data = num_data + nd2
            print_iter_info(settings, feeder, num_data, num_labels, padding)
            run_batch_count += 1
            batch_loss, learning_rate, settings.global_step = train.run_step(fdict, others)
            if run_batch_count > 1:
                fed_clips, fed_time = fed_clips + num_labels, fed_time + time.perf_counter() - tic
            if min_train_loss[0] > batch_loss:
                min_train_loss = (batch_loss, settings.global_step)
            nats = batch_loss / math.log(settings.num_classes)
            info("Learning rate %2.8f, global step: %d, batch loss/nats : %2.5f / %2.3f " % (learning_rate, settings.global_step, batch_loss, nats))
            info("Dataset global step %d, epoch index %d, batch sizes %s, batch index train %d" %
                 (settings.global_step, settings.train.epoch_index + 1, str(feeder.get_batch_sizes()), feeder.get_batch_index()))
            if feeder.should_save(run_batch_count):
                feeder.save(engine, "ep_%d_btch_%d_gs_%d" % (1 + settings.train.epoch_index, feeder.get_batch_index(), settings.global_step),
                            settings.global_step)
        info("Epoch [%d] training run complete." % (1 + settings.train.epoch_index) if run_batch_count > 0 else
             "Resumed epoch [%d] is already complete." % (1 + settings.train.epoch_index))
        settings.train.epoch_index += 1
        feeder.rewind_datasets()
    info("Minimum training loss: %2.2f on global index %d" % (min_train_loss[0], min_train_loss[1]))
    if fed_time > 0:
        info("Training throughput: %.1f clips/s over %d batches (input feed + train step)" % (fed_clips / fed_time, run_batch_count - 1))
    if run_batch_count > 0 and not feeder.should_save(run_batch_count):
        info("Saving model checkpoint out of turn, since training's finished.")
        feeder.save(engine, "ep_%d_btch_%d_gs_%d" % (1 + settings.train.epoch_index, feeder.get_num_batches(), settings.global_step),
                    settings.global_step)


def do_test(settings, val, feeder, engine, rank=0, world=1):
    """run_task.py:84-114.  Data parallel: every rank evaluates its videos of each batch (Dataset.set_shard) and the clip logits are
    gathered to rank 0 in rank order = the unsharded order; rank 0 alone aggregates per video and writes the result files."""
    tic = time.time()
    settings.global_step = 0
    dev = engine.dev
    graph = getattr(settings, "graph_tags", None) is not None
    while feeder.loop():
        fdict, num_data, num_labels, padding = feeder.get_feed_dict()
        fdicts = {defs.dataset_tag.main: fdict}
        if graph:
            for tag in settings.graph_tags:
                if tag != defs.dataset_tag.main:
                    fdicts[tag], nd2, _, _ = feeder.get_feed_dict(tag)
                    num_data = num_data + nd2
        print_iter_info(settings, feeder, num_data, num_labels, padding)
        if num_labels == 0:
            logits = np.zeros((0, settings.num_classes), np.float32)
        elif graph:
            logits = engine.forward(graph_feeds(fdicts, settings.graph_tags, dev)).cpu().numpy()
            if engine.per_step and "record_labels" in fdict:      # one logits row per record of the vectors dataset: its own targets
                fdict = dict(fdict, labels=fdict["record_labels"])
        elif "device" in fdict:          # read and uploaded ahead by the feeder's BatchPrefetcher
            torch.cuda.current_stream(dev).wait_event(fdict["ready"])
            t = fdict["device"]
            logits = engine.forward_u8(t["frames_u8"], fdict["mean_bgr"], t["crop_y"], t["crop_x"], t["mirror"],
                                       resize=fdict.get("resize")).cpu().numpy()
        else:
            logits = engine.forward_u8(torch.from_numpy(fdict["frames_u8"]).to(dev), fdict["mean_bgr"],
                                       torch.from_numpy(fdict["crop_y"]).to(dev), torch.from_numpy(fdict["crop_x"]).to(dev),
                                       torch.from_numpy(fdict["mirror"]).to(dev), resize=fdict.get("resize")).cpu().numpy()
        if num_labels:
            engine.check_status()                        # a timed-out LSTM cluster launch invalidates these logits: fail, do not save
        labels = fdict["labels"].astype(np.float32)
        if world > 1:
            parts = [None] * world
            torch.distributed.all_gather_object(parts, (logits, labels))
            logits, labels = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        if rank == 0:
            if graph and engine.per_step:                # per-step logits are their own items (no clip -> video fusion applies)
                val.add_items(logits, labels)
            else:
                val.process_validation_logits(fdict["dataset"], settings, logits, labels, fdict["batch_index"])
            val.save_validation_logits_chunk()
    if rank != 0:
        return None
    val.save_validation_logits_chunk(save_all=True)
    accuracy = val.get_accuracy()
    info("Validation run complete in [%s], accuracy: %2.5f" % (elapsed_str(tic), accuracy))
    if val.save_interval is not None:
        with open(os.path.join(settings.run_folder, "accuracy_" + settings.run_id), "w") as f:
            f.write(str(accuracy))
    return accuracy


def main(init_file, seed=0, device=None):
    """run_task.py:117-152."""
    settings = Settings()
    feeder = settings.initialize(init_file)
    rank, world, local = dpmod.init_from_env()
    batch = settings.train.batch_size if settings.train else settings.val.batch_size
    gar = dpmod.GradAllReduce() if world > 1 else None
    dev_name = device or "cuda:%d" % local
    settings.graph_tags = None
    p0 = settings.pipelines[settings.pipeline_names[0]]
    single = len(settings.pipeline_names) == 1 and p0.representation == defs.representation.dcnn and p0.input == [defs.dataset_tag.main]
    if world > 1:
        batch = -(-batch // world)           # data parallel: `batch_size` stays the GLOBAL batch of the config
    if single:
        dataset = feeder.get_dataset_by_tag(defs.dataset_tag.main)[0]
        cfg, _ = net_config(settings, dataset)
        datasets = [dataset]
    else:
        # any other list of pipelines (model.py:18-162): vltf_amd.graph -- BASELINE config 4's encoder-decoder, two-stream models ...
        from .graph import GraphEngine
        specs, infos, by_tag = graph_config(settings, feeder, batch)
        if defs.dataset_tag.main not in by_tag:
            error("No pipeline reads the [%s] dataset, whose labels the loss is taken on (train.py:117)." % defs.dataset_tag.main)
        datasets = list(by_tag.values())
        settings.graph_tags = sorted(by_tag)
    if world > 1:
        # every rank works on its videos of the global batch; rank 0 writes the checkpoints (parameters are identical everywhere)
        for d in datasets:
            d.set_shard(rank, world)
        if rank != 0:
            feeder.save = lambda *a, **k: None
    if single:
        engine = LRCNEngine(cfg, max_clips=batch * max(datasets[0].clips_per_video), device=dev_name, training=bool(settings.train), dp=gar)
        params = init_params(cfg, seed=seed)
    else:
        engine = GraphEngine(specs, infos, settings.num_classes, device=dev_name, training=bool(settings.train), dp=gar,
                             optimizer=settings.train.optimizer if settings.train else "sgd", dropout_keep_prob=settings.get_dropout(),
                             conv_math=os.environ.get("VLTF_CONV_MATH", "f32"))
        for name in engine.skipped:
            warning("Pipeline [%s] does not feed the output pipeline [%s]: it is never evaluated and is not built." %
                    (name, settings.pipeline_names[-1]))
        params = engine.init_params(seed=seed)
    for name in settings.pipeline_names:
        pl = settings.pipelines[name]
        if pl.weights_file and pl.representation == defs.representation.dcnn:
            scope = "" if single or not engine.scoped else name + "/"
            loaded = {scope + k: v for k, v in load_weights_file(pl.weights_file).items()}
            # fc8 is re-initialised (alexnet.py:273)
            params.update({k: v for k, v in loaded.items() if k in params and not k.startswith(scope + "dcnn/fc8")})
    engine.load_params(params)
    feeder.init_saveload(engine, settings.resume_file)
    if os.environ.get("VLTF_PREFETCH", "2") != "0":      # batches read + uploaded ahead of the loop (0 = the reference's synchronous feed)
        for tag in ([defs.dataset_tag.main] if single else settings.graph_tags):       # every frame dataset the model reads
            feeder.enable_prefetch(engine.dev, depth=int(os.environ.get("VLTF_PREFETCH", "2")), tag=tag)
    if gar is not None:
        gar.broadcast_params(engine.w)
    result = None
    if settings.train:
        train = Train(settings, feeder, engine)
        do_train(settings, train, feeder, engine)
        if world > 1:
            # nobody leaves before rank 0 has written the last checkpoint: a phase that follows (validation with `resume: latest`)
            # reads it on every rank
            torch.distributed.barrier()
    elif settings.val:
        result = do_test(settings, Validation(settings) if rank == 0 else None, feeder, engine, rank, world)
        if world > 1:
            torch.distributed.barrier()
    info("Run [%s] complete." % settings.run_id)
    return result


def cli(argv=None):
    """`python3 run_task.py <config.yml>` as in the reference (run_task.py:155-160), plus `--gpus N` (or VLTF_GPUS=N): data parallel over
    N GPUs of this node -- the process becomes the launcher of its N ranks (dp.self_launch) before anything touches a GPU.  Under an
    external launcher (WORLD_SIZE set) the flag is ignored and the launcher's world is used."""
    parser = argparse.ArgumentParser()
    parser.add_argument("init_file", help="Configuration .yml file for the run.")
    parser.add_argument("--gpus", type=int, default=int(os.environ.get("VLTF_GPUS", "1")), help="ranks to start on this node (default 1)")
    args = parser.parse_args(argv)
    rc = dpmod.self_launch(args.gpus)
    if rc is not None:
        raise SystemExit(rc)
    return main(args.init_file)


if __name__ == "__main__":
    cli()
