"""run_task: `python run_task.py <config.yml>` -- the reference's driver (run_task.py:117-160) over the
MI355X engine.  Same YAML, same TFRecord/.size inputs, same log lines and result files."""
import argparse
import math
import os
import time

import numpy as np
import torch

from . import dp as dpmod
from .defs_ import defs
from .engine import LRCNEngine, NetConfig, init_params
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


def net_config(settings, dataset):
    """The last pipeline defines the logits (model.py:161); only the dcnn -> lstm | fc form is on the path."""
    p = settings.pipelines[settings.pipeline_names[-1]]
    if len(settings.pipeline_names) != 1 or p.representation != defs.representation.dcnn or p.input != [defs.dataset_tag.main]:
        error("Only a single dcnn pipeline on the main dataset is built (multi-pipeline description models are out of scope).")
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
    return NetConfig(**kw), p


def load_weights_file(path):
    """alexnet.py:50-52: numpy dict {layer: [W, b]} (bvlc_alexnet.npy layout) -> tf variable names.  Loaded without
    pickle execution only if it is an .npz; the pickled .npy dict of the reference needs allow_pickle and is refused."""
    if not path.endswith(".npz"):
        error("weights_file must be an .npz of arrays named like the TF variables (a pickled .npy is not loaded)")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def do_train(settings, train, feeder, engine):
    """run_task.py:25-81."""
    run_batch_count, min_train_loss = 0, (1000, -1)
    fed_clips, fed_time = 0, 0.0          # feed + step time of every batch but the first (checkpoint writes excluded)
    info("Starting train")
    for _ in range(settings.train.epoch_index, settings.train.epochs):
        while feeder.loop():
            tic = time.perf_counter()
            fdict, num_data, num_labels, padding = feeder.get_feed_dict()
            print_iter_info(settings, feeder, num_data, num_labels, padding)
            run_batch_count += 1
            batch_loss, learning_rate, settings.global_step = train.run_step(fdict)
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
    while feeder.loop():
        fdict, num_data, num_labels, padding = feeder.get_feed_dict()
        print_iter_info(settings, feeder, num_data, num_labels, padding)
        if num_labels == 0:
            logits = np.zeros((0, settings.num_classes), np.float32)
        elif "device" in fdict:          # read and uploaded ahead by the feeder's BatchPrefetcher
            torch.cuda.current_stream(dev).wait_event(fdict["ready"])
            t = fdict["device"]
            logits = engine.forward_u8(t["frames_u8"], fdict["mean_bgr"], t["crop_y"], t["crop_x"], t["mirror"]).cpu().numpy()
        else:
            logits = engine.forward_u8(torch.from_numpy(fdict["frames_u8"]).to(dev), fdict["mean_bgr"],
                                       torch.from_numpy(fdict["crop_y"]).to(dev), torch.from_numpy(fdict["crop_x"]).to(dev),
                                       torch.from_numpy(fdict["mirror"]).to(dev)).cpu().numpy()
        labels = fdict["labels"].astype(np.float32)
        if world > 1:
            parts = [None] * world
            torch.distributed.all_gather_object(parts, (logits, labels))
            logits, labels = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        if rank == 0:
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
    dataset = feeder.get_dataset_by_tag(defs.dataset_tag.main)[0]
    cfg, pipeline = net_config(settings, dataset)
    batch = settings.train.batch_size if settings.train else settings.val.batch_size
    gar = dpmod.GradAllReduce() if world > 1 else None
    if world > 1:
        # data parallel: `batch_size` stays the GLOBAL batch of the config; every rank works on its videos of it
        dataset.set_shard(rank, world)
        batch = -(-batch // world)
        if rank != 0:
            feeder.save = lambda *a, **k: None              # rank 0 writes the checkpoints (parameters are identical everywhere)
    max_clips = batch * max(dataset.clips_per_video)
    engine = LRCNEngine(cfg, max_clips=max_clips, device=device or "cuda:%d" % local, training=bool(settings.train), dp=gar)
    params = init_params(cfg, seed=seed)
    if pipeline.weights_file:
        loaded = load_weights_file(pipeline.weights_file)
        params.update({k: v for k, v in loaded.items() if k in params and not k.startswith("dcnn/fc8")})   # fc8 is re-initialised (alexnet.py:273)
    engine.load_params(params)
    feeder.init_saveload(engine, settings.resume_file)
    if os.environ.get("VLTF_PREFETCH", "2") != "0":      # batches read + uploaded ahead of the loop (0 = the reference's synchronous feed)
        feeder.enable_prefetch(engine.dev, depth=int(os.environ.get("VLTF_PREFETCH", "2")))
    if gar is not None:
        gar.broadcast_params(engine.w)
    result = None
    if settings.train:
        train = Train(settings, feeder, engine)
        do_train(settings, train, feeder, engine)
    elif settings.val:
        result = do_test(settings, Validation(settings) if rank == 0 else None, feeder, engine, rank, world)
        if world > 1:
            torch.distributed.barrier()
    info("Run [%s] complete." % settings.run_id)
    return result


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("init_file", help="Configuration .yml file for the run.")
    main(parser.parse_args().init_file)
