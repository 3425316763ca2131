"""ComposedEngine: two chained pipelines of the reference's Model (models/model.py:18-162) -- BASELINE config 4, the video
description encoder-decoder -- behind the two-pipeline interface of round 2.  The graph itself is vltf_amd.graph.GraphEngine
(any list of pipelines); this class only names its two pipelines and two datasets:

  pipeline 1 ("enc" by default): {input: main (frames), representation: dcnn, classifier: lstm | fc} -> [rows, C].
  pipeline 2 ("dec"): {input: [aux (vectors), <pipeline 1>], representation: nop | fc, classifier: lstm}.
      * no input_fusion: the second input is the LSTM's state vector (model.py:128-134): replicate_auxilliary_tensor
        (tf_util.py:182-192), convert_dim_fc "input_state_fc" when its width differs from the hidden size (lstm.py:74-77), and
        c = h = state in every layer (get_state_tuple, lstm.py:34-42);
      * input_fusion concat | ibias (tf_util.py:146-176): the two inputs are fused into the sequence instead (the pipeline-1
        vector of a clip is prepended to every word vector / inserted as an extra first time step) and the state is zero;
      * lstm_params fusion avg | last | state give one logits row per clip, `reshape` (tf_util.py:26-27) one per time step
        (word-level cross-entropy: labels [B*T, classes]).

Every variable is scoped by its pipeline's name: "<pipeline>/<tf name>" (see graph.py).  The oracle of this composition is
oracle.lrcn_oracle.encdec_forward / lstm_classifier_* / tensor_list_fusion (= model_forward on these two pipelines)."""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from ._ffi import VltfError
from .engine import NetConfig
from .graph import DatasetInfo, GraphEngine, PipelineSpec, _trunc_normal


@dataclass
class HeadConfig:
    """Second pipeline: keys of settings_.py:167-208 that shape it."""
    in_dim: int                              # width of the vectors dataset (dimension feature of its records)
    fpc: int                                 # its sequence length (frames per clip of the vectors dataset)
    num_classes: int
    lstm_hidden: int = 256
    lstm_layers: int = 1
    fusion: str = "reshape"                  # lstm_params[2]: avg | last | reshape | state
    representation: str = "nop"              # nop | fc
    fc_output_dim: Optional[int] = None
    input_fusion: Optional[str] = None       # None (pipeline 1 = LSTM state) | concat | ibias
    dropout_keep_prob: float = 0.0
    cpv_ratio: int = 1                       # clips per video of the sequence dataset / of pipeline 1: its vectors are replicated
                                             # that many times (replicate_auxilliary_tensor, tf_util.py:182-192: whole batch tiled)


def head_specs(h: HeadConfig, enc_dim: int, scope: str):
    """[(name, shape)] of pipeline 2 (head fc, LSTM layers last to first, state fc, representation fc)."""
    fused = h.in_dim + enc_dim if h.input_fusion == "concat" else h.in_dim       # the input fusion comes first (model.py:69-73)
    seq_dim = h.fc_output_dim if h.representation == "fc" else fused            # then the representation (model.py:81-96)
    H, C = h.lstm_hidden, h.num_classes
    specs = []
    if H != C:
        head = "fc_convert" if h.fusion == "state" else "output_fc"
        specs += [(scope + head + "_w", (H, C)), (scope + head + "_b", (C,))]
    dims = [seq_dim] + [H] * (h.lstm_layers - 1)
    for l in reversed(range(h.lstm_layers)):
        pre = scope + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
        specs += [(pre + "kernel", (dims[l] + H, 4 * H)), (pre + "bias", (4 * H,))]
    if h.input_fusion is None and enc_dim != H:
        specs += [(scope + "input_state_fc_w", (enc_dim, H)), (scope + "input_state_fc_b", (H,))]
    if h.representation == "fc" and h.fc_output_dim != fused:
        specs += [(scope + "fc_convert_w", (fused, h.fc_output_dim)), (scope + "fc_convert_b", (h.fc_output_dim,))]
    return specs


def init_head_params(h: HeadConfig, enc_dim, scope, seed=0, stddev=0.05, well_scaled=False):
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in head_specs(h, enc_dim, scope):
        if name.endswith("kernel"):
            lim = math.sqrt(6.0 / (shp[0] + shp[1]))
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = np.zeros(shp, np.float32)
        elif len(shp) > 1:
            out[name] = _trunc_normal(rng, shp, math.sqrt(2.0 / shp[0]) if well_scaled else stddev)
        else:
            out[name] = np.full(shp, 0.1, np.float32)
    return out


class ComposedEngine(GraphEngine):
    def __init__(self, enc_cfg: NetConfig, head: HeadConfig, max_clips: int, device="cuda:0", training=True, dp=None,
                 scopes=("enc", "dec")):
        if head.fusion not in ("avg", "last", "reshape", "state"):
            raise VltfError("Undefined frame fusion type : %s" % head.fusion)
        if head.input_fusion not in (None, "concat", "ibias"):
            raise VltfError("input_fusion [%s] is not built for a vector sequence fused with a per-clip vector (concat | ibias)"
                            % head.input_fusion)
        if enc_cfg.classifier not in ("lstm", "fc"):
            raise VltfError("pipeline 1 of the two-pipeline model needs a classifier (lstm | fc)")
        self.h = head
        p1 = PipelineSpec(scopes[0], ["main"], "dcnn", frame_encoding_layer=enc_cfg.frame_encoding_layer, classifier=enc_cfg.classifier,
                          lstm_params=(enc_cfg.lstm_hidden, enc_cfg.lstm_layers, enc_cfg.fusion) if enc_cfg.classifier == "lstm" else None,
                          frame_fusion=enc_cfg.frame_fusion if enc_cfg.classifier == "fc" else None)
        p2 = PipelineSpec(scopes[1], ["aux", scopes[0]], head.representation, fc_output_dim=head.fc_output_dim, classifier="lstm",
                          lstm_params=(head.lstm_hidden, head.lstm_layers, head.fusion), input_fusion=head.input_fusion)
        datasets = {"main": DatasetInfo("video", enc_cfg.fpc, 1, max_clips, image_shape=tuple(enc_cfg.image_shape)),
                    "aux": DatasetInfo("vectors", head.fpc, head.cpv_ratio, max_clips * head.cpv_ratio, dim=head.in_dim)}
        super().__init__([p1, p2], datasets, head.num_classes, device, training, dp, optimizer=enc_cfg.optimizer,
                         dropout_keep_prob=head.dropout_keep_prob or enc_cfg.dropout_keep_prob, conv_math=enc_cfg.conv_math)
        self.enc_cfg = enc_cfg
        self.Ts = self.last.fpc                  # steps the second LSTM runs (one more under ibias)

    @staticmethod
    def _feeds(frames_u8, words, mean_bgr, crop_y, crop_x, mirror, resize):
        return {"main": dict(frames_u8=frames_u8, mean_bgr=mean_bgr, crop_y=crop_y, crop_x=crop_x, mirror=mirror, resize=resize),
                "aux": words}

    def forward(self, frames_u8, words, mean_bgr=None, crop_y=None, crop_x=None, mirror=None, resize=None):
        """sess.run(model.logits, fdict) for the two-pipeline model.  words: device float32 [clips * fpc, in_dim]."""
        return GraphEngine.forward(self, self._feeds(frames_u8, words, mean_bgr, crop_y, crop_x, mirror, resize))

    def train_step(self, frames_u8, words, onehot, lr, clip_norm=0.0, mean_bgr=None, crop_y=None, crop_x=None, mirror=None,
                   fetch=True, global_rows=None, resize=None):
        """sess.run([.., loss, .., optimizer], fdict): labels int32 one-hot [rows, classes], rows = clips (fusion avg | last |
        state) or clips * steps (fusion reshape: one row per time step, clip-major)."""
        return GraphEngine.train_step(self, self._feeds(frames_u8, words, mean_bgr, crop_y, crop_x, mirror, resize), onehot, lr,
                                      clip_norm, fetch, global_rows)
