"""Logging and small helpers with the reference's surface (utils_.py): info / debug / warning / error,
CustomLogger format, labels_to_one_hot, read_file_dict, get_datetime_str.  E-mail notification and
TensorBoard summaries are out of scope."""
import logging
import os
import time

import numpy as np

LOGGER_NAME = "vltf"


def get_datetime_str():
    """utils_.py:24-26 format ddmmyy_HHMMSS (used in log / checkpoint / logits file names)."""
    return time.strftime("%d%m%y_%H%M%S")


def elapsed_str(tic):
    return time.strftime("%H:%M:%S", time.gmtime(time.time() - tic))


class CustomLogger:
    """utils_.py:41-93: file + console handlers, format '%(asctime)s| %(levelname)7s - %(message)s'."""
    instance = None

    def configure_logging(self, logfile, level="logging.INFO"):
        lvl = {"logging.INFO": logging.INFO, "logging.DEBUG": logging.DEBUG, "logging.WARN": logging.WARN}.get(level)
        if lvl is None:
            raise Exception("Invalid logging level: %s" % level)
        logger = logging.getLogger(LOGGER_NAME)
        logger.handlers = []
        logger.setLevel(lvl)
        logger.propagate = False
        fmt = logging.Formatter("%(asctime)s| %(levelname)7s - %(message)s", "%d/%m %H:%M:%S")
        for h in ([logging.FileHandler(logfile)] if logfile else []) + [logging.StreamHandler()]:
            h.setFormatter(fmt)
            h.setLevel(lvl)
            logger.addHandler(h)
        self.logfile = logfile
        CustomLogger.instance = self
        return logger


def info(msg):
    logging.getLogger(LOGGER_NAME).info(msg)


def debug(msg):
    logging.getLogger(LOGGER_NAME).debug(msg)


def warning(msg):
    logging.getLogger(LOGGER_NAME).warning(msg)


def error(msg):
    """utils_.py:133-136: log, then raise -- fail fast, no error codes, no interactive prompts."""
    logging.getLogger(LOGGER_NAME).error(msg)
    raise Exception(msg)


def labels_to_one_hot(labels, num_classes):
    """utils_.py:160-169: labels = list of per-item label lists -> int32 [items, num_classes]."""
    if not isinstance(labels, list):
        labels = [labels]
    labels = [l if isinstance(l, (list, tuple)) else [l] for l in labels]
    maxlbl = max(lbl for item in labels for lbl in item)
    if maxlbl >= num_classes:
        error("Encountered label %d but the number of labels was set to %d" % (maxlbl, num_classes))
    onehots = np.zeros((len(labels), num_classes), np.int32)
    for i, item in enumerate(labels):
        onehots[i][list(item)] = 1
    return onehots


def read_file_dict(filename):
    """utils_.py:232-243: 'key<TAB>value' lines."""
    d = {}
    with open(filename, "r") as f:
        for line in f:
            if not line.strip():
                continue
            key, value = line.strip().split("\t")
            d[key.strip()] = value.strip()
    return d


def read_file_lines(filename):
    with open(filename, "r") as f:
        return [line.strip() for line in f]


def get_run_checkpoints(run_folder):
    """utils_.py:221-228 adapted to our weight container: <name>.graph-<gs>.weights.npz files, oldest first."""
    folder = os.path.join(run_folder, "checkpoints")
    files = [os.path.join(folder, x) for x in os.listdir(folder) if x.endswith(".weights.npz")]
    files.sort(key=os.path.getmtime)
    return [f[:-len(".weights.npz")] for f in files]
