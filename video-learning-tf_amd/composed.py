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
from dataclasses import dataclass
from typing import Optional

from ._ffi import VltfError
from .engine import NetConfig
from .graph import DatasetInfo, GraphEngine, PipelineSpec, init_params_for, model_specs


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


def two_pipelines(enc_cfg: NetConfig, head: HeadConfig, max_clips: int, scopes=("enc", "dec")):
    """([PipelineSpec, PipelineSpec], {tag: DatasetInfo}) of the two-pipeline model: what ComposedEngine hands to GraphEngine."""
    p1 = PipelineSpec(scopes[0], ["main"], "dcnn", frame_encoding_layer=enc_cfg.frame_encoding_layer, classifier=enc_cfg.classifier,
                      lstm_params=(enc_cfg.lstm_hidden, enc_cfg.lstm_layers, enc_cfg.fusion) if enc_cfg.classifier == "lstm" else None,
                      frame_fusion=enc_cfg.frame_fusion if enc_cfg.classifier == "fc" else None)
    p2 = PipelineSpec(scopes[1], ["aux", scopes[0]], head.representation, fc_output_dim=head.fc_output_dim, classifier="lstm",
                      lstm_params=(head.lstm_hidden, head.lstm_layers, head.fusion), input_fusion=head.input_fusion)
    datasets = {"main": DatasetInfo("video", enc_cfg.fpc, 1, max_clips, image_shape=tuple(enc_cfg.image_shape)),
                "aux": DatasetInfo("vectors", head.fpc, head.cpv_ratio, max_clips * head.cpv_ratio, dim=head.in_dim)}
    return [p1, p2], datasets


def head_specs(enc_cfg: NetConfig, head: HeadConfig, scopes=("enc", "dec")):
    """[(name, shape)] of pipeline 2's variables, from the ONE place that defines a model's variable list: the graph's plan
    (graph.model_specs -> PipeNode.head_specs).  Round 3 kept a second copy of that logic here."""
    pipes, datasets = two_pipelines(enc_cfg, head, 1, scopes)
    return [(n, s) for n, s in model_specs(pipes, datasets, head.num_classes) if n.startswith(scopes[1] + "/")]


def init_head_params(enc_cfg: NetConfig, head: HeadConfig, scopes=("enc", "dec"), seed=0, stddev=0.05, well_scaled=False):
    """Reference initialisers (graph.init_params_for) for pipeline 2's variables only."""
    return init_params_for(head_specs(enc_cfg, head, scopes), seed=seed, stddev=stddev, well_scaled=well_scaled)


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
        pipes, datasets = two_pipelines(enc_cfg, head, max_clips, scopes)
        super().__init__(pipes, datasets, head.num_classes, device, training, dp, optimizer=enc_cfg.optimizer,
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
