"""Settings: the `run:` block of the reference's YAML (settings_.py:210-444; key list in SURVEY.md Appendix A)."""
import logging
import os
from shutil import copyfile

import yaml

from .defs_ import defs
from .feeder import Feeder
from .parse_opts import parse_seq
from .utils_ import CustomLogger, debug, error, get_datetime_str, info, warning


class Network:
    description = "Network representative class"


class TrainSettings:
    batch_size, epochs, epoch_index = 100, 15, 0
    optimizer, base_lr, lr_mult = defs.optim.sgd, 0.001, None
    lr_decay, clip_norm, dropout_keep_prob = None, 0, 0.5


class ValSettings:
    batch_size, logits_save_interval = 88, -1
    clip_fusion_type, clip_fusion_method = defs.fusion_type.late, defs.fusion_method.avg


class Settings:
    def __init__(self):
        self.run_id, self.resume_file, self.run_folder = "", None, None
        self.global_step = 0
        self.feeder = None
        self.pipelines, self.pipeline_names = {}, []
        self.train = self.val = None

    def should_resume(self):
        return bool(self.resume_file)

    def get_dropout(self):
        return self.train.dropout_keep_prob if self.phase == defs.phase.train else 0.0

    # ---- pipelines (settings_.py:134-208) ---------------------------------------------------------------
    def read_field(self, config, fieldname, validate=None, required=False, listify=False):
        self.pipeline_field_cache.append(fieldname)
        val = config.get(fieldname)
        if val is None:
            if required:
                error("No default value specified for missing field [%s]" % fieldname)
            return [None] if listify else None
        if validate is not None:
            if isinstance(validate, (list, tuple)):
                val = list(parse_seq(val))
                if len(validate) != len(val):
                    error("Field [%s] required %d entries, found: [%s]" % (fieldname, len(validate), str(val)))
                val = [defs.check(el, v) for el, v in zip(val, validate)]
            else:
                val = defs.check(val, validate)
        if listify and not isinstance(val, (list, tuple)):
            val = [val]
        return val

    def read_network(self, content):
        network = Network()
        self.pipeline_field_cache = []
        network.input = list(self.read_field(content, "input", listify=True))
        if any(x is None for x in network.input):
            error("<None> or undefined <input> tag in pipeline: %s" % content)
        for i, inp in enumerate(network.input):
            if inp not in self.pipelines:
                ok, tagname = defs.check(inp, defs.dataset_tag, do_boolean=True)
                if not ok:
                    error("Input identifier [%s] is not a dataset tag, but no such pipeline has been declared yet." % inp)
                network.input[i] = tagname
        network.representation = self.read_field(content, "representation", required=True, validate=defs.representation)
        network.frame_encoding_layer = None
        if network.representation == defs.representation.dcnn:
            network.frame_encoding_layer = self.read_field(content, "frame_encoding_layer", required=True)
        if network.representation == defs.representation.fc:
            network.fc_output_dim = self.read_field(content, "fc_output_dim", required=True)
        network.classifier = self.read_field(content, "classifier", validate=defs.classifier)
        network.lstm_params = None
        if network.classifier == defs.classifier.lstm:
            params = parse_seq(self.read_field(content, "lstm_params", required=True))
            network.lstm_params = [int(params[0]), int(params[1]), defs.check(params[2], defs.fusion_method)]
        network.weights_file = self.read_field(content, "weights_file")
        network.frame_fusion = self.read_field(content, "frame_fusion", validate=(defs.fusion_type, defs.fusion_method))
        network.input_shape = self.read_field(content, "input_shape", listify=True)
        network.input_fusion = self.read_field(content, "input_fusion", validate=defs.fusion_method)
        unread = [x for x in content if x not in self.pipeline_field_cache]
        if unread:
            error("Undefined pipeline field(s):" + str(unread))
        return network

    # ---- run block (settings_.py:210-366) -------------------------------------------------------------------
    def read_config(self, config, init_file):
        self.resume_file = config.get("resume_file")
        self.run_folder = config["run_folder"]
        self.run_id = config.get("run_id") or ""
        self.phases = defs.check(config["phase"], defs.phase)
        if not isinstance(self.phases, list):
            self.phases = [self.phases]
        self.phase = self.phases[0]
        trainval = ("train" if defs.phase.train in self.phases else "") + ("val" if defs.phase.val in self.phases else "")
        trainval += "_resume" if self.should_resume() else "_scratch"
        self.run_id = "_".join([self.run_id or os.path.basename(init_file), trainval])
        os.makedirs(self.run_folder, exist_ok=True)       # (every rank of a data-parallel launch gets here at the same moment)
        lg = config["logging"]
        self.save_freq_per_epoch = lg["save_freq_per_epoch"]
        self.logging_level = lg["level"]
        self.tensorboard_folder = lg.get("tensorboard_folder", "tensorboard")
        self.print_tensors = lg.get("print_tensors", False)
        self.configure_logging()

        for pipeline in config["network"]["pipelines"]:
            pname, content = list(pipeline.items())[0]
            debug("Reading network [%s]" % pname)
            self.pipelines[pname] = self.read_network(content)
            self.pipeline_names.append(pname)
        self.num_classes = int(config["network"]["num_classes"])

        for phase in self.phases:
            obj = config[phase]
            if phase == defs.phase.train:
                t = self.train = TrainSettings()
                t.batch_size, t.epochs = int(obj["batch_size"]), int(obj["epochs"])
                t.optimizer = defs.check(obj["optimizer"], defs.optim)
                t.base_lr = float(obj["base_lr"])
                t.lr_mult = float(obj["lr_mult"]) if obj.get("lr_mult") not in (None, "None") else None
                if t.lr_mult is not None:
                    error("Two-tier learning rates (lr_mult) are broken in the reference (train.py:152-197) and not built.")
                if obj.get("lr_decay") in (None, "None"):
                    t.lr_decay = None
                else:
                    d = parse_seq(obj["lr_decay"])
                    t.lr_decay = [defs.check(d[0], defs.decay), defs.check(d[1], defs.periodicity), int(d[2]), float(d[3])] + \
                        ([int(d[4])] if len(d) > 4 else [])
                t.clip_norm = int(obj["clip_norm"]) if obj.get("clip_norm") not in (None, "None") else 0
                t.dropout_keep_prob = float(obj["dropout_keep_prob"])
            if phase == defs.phase.val:
                v = self.val = ValSettings()
                v.batch_size = int(obj["batch_size"])
                v.logits_save_interval = int(obj["logits_save_interval"])
                cf = parse_seq(obj["clip_fusion"])
                v.clip_fusion_type, v.clip_fusion_method = defs.check(cf[0], defs.fusion_type), defs.check(cf[1], defs.fusion_method)

        self.feeder = Feeder(defs.input_mode.video, self.phases, (self.train, self.val), self.save_freq_per_epoch, self.run_folder,
                             self.should_resume())
        for dataid, dataobj in config["data"].items():
            dataset_phase = defs.check(dataobj["phase"], defs.phase)
            if dataset_phase not in self.phases:
                info("Omitting dataset [%s] due to its phase [%s]" % (dataid, dataset_phase))
                continue
            mean_image = parse_seq(dataobj["mean_image"]) if "mean_image" in dataobj else None
            batch_item = defs.check(dataobj["batch_item"], defs.batch_item) if "batch_item" in dataobj else defs.batch_item.default
            image_shape = parse_seq(dataobj["image_shape"]) if "image_shape" in dataobj else None
            imgproc = [defs.check(o, defs.imgproc) for o in (parse_seq(dataobj["imgproc"]) if "imgproc" in dataobj else [])]
            if defs.imgproc.sub_mean in imgproc and not mean_image:
                error("[%s] option requires a supplied mean image intensity." % defs.imgproc.sub_mean)
            raw_image_shape = parse_seq(dataobj["raw_image_shape"]) if "raw_image_shape" in dataobj else None
            ncrop = sum(o in imgproc for o in (defs.imgproc.rand_crop, defs.imgproc.center_crop, defs.imgproc.resize))
            if ncrop > 1:
                error("Need at most one image processing parameter. Imgproc params : %s" % imgproc)
            if mean_image is not None and defs.imgproc.sub_mean not in imgproc:
                imgproc.append(defs.imgproc.sub_mean)                      # settings_.py:341-342
            if self.val and (defs.imgproc.rand_crop in imgproc or defs.imgproc.rand_mirror in imgproc):
                error("Random cropping / mirroring is enabled in validation mode (the reference prompts; we fail).")
            self.feeder.add_dataset(dataset_phase, dataid, dataobj["data_path"], mean_image, dataobj.get("prepend_folder"),
                                    image_shape, imgproc, raw_image_shape, defs.check(dataobj["data_format"], defs.data_format),
                                    dataobj.get("frame_format"), batch_item, self.num_classes,
                                    defs.check(dataobj["tag"], defs.dataset_tag), int(dataobj.get("read_tries", 1)))

    def configure_logging(self):
        self.timestamp = get_datetime_str()
        logfile = os.path.join(self.run_folder, "log_" + self.run_id + "_" + self.timestamp + ".log")
        if int(os.environ.get("RANK", "0")) > 0:      # data parallel: rank 0 owns the run's log file, the others log to the console
            logfile = None
        self.logger = CustomLogger()
        self.logger.configure_logging(logfile, self.logging_level)

    def initialize(self, init_file):
        """settings_.py:404-444 -> Feeder."""
        if not os.path.exists(init_file):
            raise Exception("Unable to read initialization file [%s]." % init_file)
        if init_file.endswith(".ini"):
            raise Exception(".ini files deprecated.")
        with open(init_file, "r") as f:
            config = yaml.safe_load(f)["run"]
        self.read_config(config, init_file)
        info("Initialized from configuration file: [%s]" % init_file)
        if os.path.abspath(os.path.dirname(init_file)) != os.path.abspath(self.run_folder):
            copyfile(init_file, os.path.join(self.run_folder, os.path.basename(init_file)))
        if self.train and self.val:
            error("Cannot specify simultaneous training and validation run, for now.")
        if not (self.train or self.val):
            error("Neither training nor validation is enabled.")
        self.tensorboard_folder = os.path.join(self.run_folder, self.tensorboard_folder, self.phase)
        self.feeder.set_phase(self.phase)
        self.feeder.initialize_datasets()
        if self.should_resume():
            if self.train:
                info("Resuming training.")
                self.train.epoch_index, self.global_step = self.feeder.resume_snap(self.resume_file)
            if self.val:
                info("Evaluating trained network.")
        else:
            info("Starting training from scratch." if self.train else "Starting validation-only run with an untrained network.")
        info("Starting run on folder [%s]." % self.run_folder)
        return self.feeder
