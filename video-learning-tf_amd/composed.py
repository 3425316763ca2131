"""ComposedEngine: two chained pipelines of the reference's Model (models/model.py:18-162) -- BASELINE config 4, the video
description encoder-decoder.

  pipeline 1 ("enc" by default): any single LRCN pipeline (LRCNEngine): frames -> dcnn -> lstm | fc -> [rows, C1].
  pipeline 2 ("dec"): {input: [<vectors dataset>, <pipeline 1>], representation: nop | fc, classifier: lstm}.
      * no input_fusion: the second input is the LSTM's state vector (model.py:128-134): replicate_auxilliary_tensor
        (tf_util.py:182-192), convert_dim_fc "input_state_fc" when its width differs from the hidden size (lstm.py:74-77), and
        c = h = state in every layer (get_state_tuple, lstm.py:34-42);
      * input_fusion concat | ibias (tf_util.py:146-176): the two inputs are fused into the sequence instead (the pipeline-1
        vector of a clip is prepended to every word vector / inserted as an extra first time step) and the state is zero;
      * lstm_params fusion avg | last | state give one logits row per clip, `reshape` (tf_util.py:26-27) one per time step
        (word-level cross-entropy: labels [B*T, classes]).

In the reference both LSTMs would claim the same TF variable names ("rnn/multi_rnn_cell/...", "output_fc_w", ...) and the graph
could not be built; here every pipeline of a multi-pipeline model is scoped by its name: "<pipeline>/<tf name>".  The oracle of
this composition is oracle.lrcn_oracle.encdec_forward / lstm_classifier_* / tensor_list_fusion.

All parameters live in ONE flat buffer ordered [pipeline 2 | pipeline 1] = the order backward produces their gradients, so the
global-norm clip, the update and the data-parallel exchange treat them exactly as LRCNEngine treats a single pipeline."""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import ops
from ._ffi import VltfError
from .engine import FORGET_BIAS, LRCNEngine, NetConfig, param_specs


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
    """[(name, shape)] of pipeline 2 in flat order (= backward order: head fc, LSTM layers last to first, state fc, representation fc)."""
    e = h.fc_output_dim if h.representation == "fc" else h.in_dim
    seq_dim = e + enc_dim if h.input_fusion == "concat" else e
    if h.input_fusion == "ibias" and enc_dim != e:
        raise VltfError("input_fusion ibias needs equal widths (pipeline 1 gives %d, the sequence %d)" % (enc_dim, e))
    H, C = h.lstm_hidden, h.num_classes
    specs = []
    if H != C:
        head = "fc_convert" if h.fusion == "state" else "output_fc"
        if head == "fc_convert" and h.representation == "fc":
            raise VltfError("representation fc and lstm fusion state would both create the variable fc_convert")
        specs += [(scope + head + "_w", (H, C)), (scope + head + "_b", (C,))]
    dims = [seq_dim] + [H] * (h.lstm_layers - 1)
    for l in reversed(range(h.lstm_layers)):
        pre = scope + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
        specs += [(pre + "kernel", (dims[l] + H, 4 * H)), (pre + "bias", (4 * H,))]
    if h.input_fusion is None and enc_dim != H:
        specs += [(scope + "input_state_fc_w", (enc_dim, H)), (scope + "input_state_fc_b", (H,))]
    if h.representation == "fc" and h.fc_output_dim != h.in_dim:
        specs += [(scope + "fc_convert_w", (h.in_dim, h.fc_output_dim)), (scope + "fc_convert_b", (h.fc_output_dim,))]
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
            v = rng.standard_normal(shp)
            bad = np.abs(v) > 2.0
            while bad.any():
                v[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(v) > 2.0
            out[name] = (v * (math.sqrt(2.0 / shp[0]) if well_scaled else stddev)).astype(np.float32)
        else:
            out[name] = np.full(shp, 0.1, np.float32)
    return out


class ComposedEngine:
    def __init__(self, enc_cfg: NetConfig, head: HeadConfig, max_clips: int, device="cuda:0", training=True, dp=None,
                 scopes=("enc", "dec")):
        if not torch.cuda.is_available():
            raise VltfError("ComposedEngine needs a HIP device; there is no CPU fallback")
        if head.fusion not in ("avg", "last", "reshape", "state"):
            raise VltfError("Undefined frame fusion type : %s" % head.fusion)
        if head.input_fusion == "concat" and head.cpv_ratio == 1 and head.fpc > 1:
            # apply_tensor_list_fusion takes tf.concat(inputs, axis=1) at cpv ratio 1 (tf_util.py:147-148): [B*T, E] with [B, C] cannot be
            raise VltfError("input_fusion concat of a %d-step sequence with a per-clip vector needs a clips-per-video ratio > 1 "
                            "(at ratio 1 the reference concatenates row by row and the row counts differ)" % head.fpc)
        if head.input_fusion not in (None, "concat", "ibias"):
            raise VltfError("input_fusion [%s] is not built for a vector sequence fused with a per-clip vector (concat | ibias)"
                            % head.input_fusion)
        self.h, self.dev, self.training, self.dp = head, torch.device(device), training, dp
        self.B = max_clips
        self.scope1, self.scope2 = scopes[0] + "/", scopes[1] + "/"
        enc_dim = enc_cfg.num_classes          # every classifier ends at num_classes (model.py:115-119,140-141), one value per model
        self.enc_dim = enc_dim
        self.specs2 = head_specs(head, enc_dim, self.scope2)
        enc_specs = param_specs(enc_cfg)
        n2 = sum(int(np.prod(s)) for _, s in self.specs2)
        n1 = sum(int(np.prod(s)) for _, s in enc_specs)
        dev = self.dev
        torch.cuda.set_device(dev)
        self.w = torch.zeros(n1 + n2, device=dev)
        self.g = torch.zeros(n1 + n2, device=dev) if training else None
        self.enc = LRCNEngine(enc_cfg, max_clips, device, training, dp=None,
                              flat=(self.w[n2:], self.g[n2:] if training else None))
        self.specs = self.specs2 + [(self.scope1 + n, s) for n, s in enc_specs]
        self.P, self.G, off = {}, {}, 0
        for name, shp in self.specs2:
            n = int(np.prod(shp))
            self.P[name] = self.w[off:off + n].view(shp)
            if training:
                self.G[name] = self.g[off:off + n].view(shp)
            off += n
        for n, _ in enc_specs:
            self.P[self.scope1 + n] = self.enc.P[n]
            if training:
                self.G[self.scope1 + n] = self.enc.G[n]
        # data-parallel chunks: pipeline 2 first (its gradients are complete before pipeline 1's backward starts)
        self.grad_chunks = ([(0, n2)] if n2 else []) + [(n2 + lo, cnt) for lo, cnt in self.enc.grad_chunks]
        if enc_cfg.optimizer == "adam" and training:
            self.adam_m, self.adam_v = torch.zeros(n1 + n2, device=dev), torch.zeros(n1 + n2, device=dev)
        self.optimizer = enc_cfg.optimizer
        self.step_count = 0

        def buf(*shape, dtype=torch.float32):
            return torch.empty(shape, dtype=dtype, device=dev)

        self.B1 = max_clips                                                  # clips of pipeline 1
        B, T, H, C = max_clips * head.cpv_ratio, head.fpc, head.lstm_hidden, head.num_classes
        self.rep = buf(B, enc_dim) if head.cpv_ratio > 1 else None
        self.E = head.fc_output_dim if head.representation == "fc" else head.in_dim
        self.Ts = T + 1 if head.input_fusion == "ibias" else T               # steps the LSTM runs
        self.seq_dim = self.E + enc_dim if head.input_fusion == "concat" else self.E
        R = B * self.Ts
        self.xfc = buf(B * T, self.E) if (head.representation == "fc" and head.fc_output_dim != head.in_dim) else None
        self.xseq = buf(R, self.seq_dim) if head.input_fusion else None
        self.state = buf(B, H) if (head.input_fusion is None and enc_dim != H) else None
        self.lstm = []
        for l in range(head.lstm_layers):
            S = dict(gx=buf(R, 4 * H), act=buf(R, 4 * H), cseq=buf(R, H), hseq=buf(R, H), hprev=buf(R, H))
            if training:
                S.update(dz=buf(R, 4 * H), dout=buf(R, H), dh0=buf(B, H), dc0=buf(B, H))
            self.lstm.append(S)
        self.lstm_ws = ops.lstm_seq_ws(B, self.Ts, H, dev)
        self.per_step = head.fusion == "reshape"
        rows = R if self.per_step else B
        self.fused = buf(rows, H) if not self.per_step else None
        self.dropped = buf(rows, H)
        self.drop_mask = buf(rows, H, dtype=torch.uint8)
        self.logits = buf(rows, C) if H != C else self.dropped
        self.small_ws = buf(64 * max(4 * H, C, 1024, self.seq_dim, H, enc_dim, self.B1 * enc_dim if head.cpv_ratio > 1 else 1))
        if training:
            self.dlogits = buf(rows, C)
            self.dpre = buf(rows, H)                       # d(dropout output), d(dropout input)
            self.dfused = buf(rows, H)
            self.dstate = buf(B, H)
            self.drep = buf(B, enc_dim) if head.cpv_ratio > 1 else None
            self.dxseq = buf(R, self.seq_dim) if (head.input_fusion or self.xfc is not None) else None
            self.dxfc = buf(B * T, self.E) if self.xfc is not None else None
            self.tmp_enc = buf(B, enc_dim) if head.input_fusion == "concat" else None
        self.stats = torch.zeros(2, device=dev)
        self.loss_rows = torch.zeros(2 * rows, device=dev)              # per-row losses | hits (vl_softmax_xent workspace)
        self.ss = torch.zeros(1, device=dev)

    # ---- parameters --------------------------------------------------------------------------------------------------------------
    def load_params(self, params: dict):
        missing = [n for n, _ in self.specs if n not in params]
        extra = [n for n in params if n not in self.P]
        if missing or extra:
            raise VltfError("parameter set mismatch: missing %s, unexpected %s" % (missing, extra))
        for name, shp in self.specs:
            a = np.asarray(params[name], np.float32)
            if tuple(a.shape) != tuple(shp):
                raise VltfError("parameter %s has shape %s, expected %s" % (name, a.shape, shp))
            self.P[name].copy_(torch.from_numpy(np.ascontiguousarray(a)))

    def get_params(self):
        torch.cuda.synchronize(self.dev)
        return {n: self.P[n].detach().cpu().numpy().copy() for n, _ in self.specs}

    def get_grads(self):
        torch.cuda.synchronize(self.dev)
        return {n: self.G[n].detach().cpu().numpy().copy() for n, _ in self.specs}

    def check_status(self):
        """LRCNEngine.check_status over both pipelines' LSTM workspaces."""
        ops.lstm_seq_check(self.lstm_ws, getattr(self.enc, "lstm_ws", None))

    def logits_host(self):
        torch.cuda.synchronize(self.dev)
        self.check_status()
        return self.logits[:self._rows].detach().cpu().numpy().copy()

    # checkpoints (feeder.py): same reserved keys as LRCNEngine
    OPT_PREFIX = LRCNEngine.OPT_PREFIX
    get_opt_state = LRCNEngine.get_opt_state
    load_opt_state = LRCNEngine.load_opt_state

    @property
    def cfg(self):
        return self.enc.cfg

    # ---- forward -------------------------------------------------------------------------------------------------------------------
    def _head_forward(self, words, b1, train):
        h, P, sc = self.h, self.P, self.scope2
        b = b1 * h.cpv_ratio                                            # clips of this pipeline
        T, Ts, H, C, E = h.fpc, self.Ts, h.lstm_hidden, h.num_classes, self.E
        if tuple(words.shape) != (b * T, h.in_dim) or words.dtype != torch.float32:
            raise VltfError("word vectors must be float32 [%d, %d], got %s" % (b * T, h.in_dim, tuple(words.shape)))
        enc_out = self.enc.logits                                       # [b1, enc_dim]: pipeline 1's output
        if self.rep is not None:                                        # replicate_auxilliary_tensor: the batch of vectors, cpv_ratio times over
            ops.copy2d(enc_out, self.rep, h.cpv_ratio, b1 * self.enc_dim, src_ld=0, dst_ld=b1 * self.enc_dim)
            enc_out = self.rep
        self._enc_out, self._b = enc_out, b
        x = words
        if self.xfc is not None:                                        # representation fc = convert_dim_fc (vectorizer.py:77-78)
            ops.gemm(x, P[sc + "fc_convert_w"], self.xfc, b * T, E, h.in_dim, bias=P[sc + "fc_convert_b"])
            x = self.xfc
        s0 = None
        if h.input_fusion == "concat":                                  # vec_seq_concat, vector first (tf_util.py:99-124)
            for t in range(T):                                          # each clip's vector in front of each of its T word vectors
                ops.copy2d(enc_out, self.xseq[t:], b, self.enc_dim, src_ld=self.enc_dim, dst_ld=T * self.seq_dim)
            ops.copy2d(x, self.xseq[:, self.enc_dim:], b * T, E, src_ld=E, dst_ld=self.seq_dim)
            x = self.xseq
        elif h.input_fusion == "ibias":                                 # the vector becomes time step 0 (tf_util.py:154-176)
            ops.copy2d(enc_out, self.xseq, b, E, src_ld=E, dst_ld=Ts * E)
            ops.copy2d(x, self.xseq[1:], b, T * E, src_ld=T * E, dst_ld=Ts * E)
            x = self.xseq
        else:                                                           # pipeline 1 = the state vector (model.py:128-134)
            s0 = enc_out
            if self.state is not None:
                ops.gemm(enc_out, P[sc + "input_state_fc_w"], self.state, b, H, self.enc_dim, bias=P[sc + "input_state_fc_b"])
                s0 = self.state
        self._x, self._s0 = x, s0
        xin, d = x, self.seq_dim
        for l, S in enumerate(self.lstm):
            pre = sc + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
            K = P[pre + "kernel"]
            ops.gemm(xin, K, S["gx"], b * Ts, 4 * H, d, bias=P[pre + "bias"])
            ops.lstm_seq_fwd(S["gx"], K[d:], S["act"], S["cseq"], S["hseq"], S["hprev"], b, Ts, H, FORGET_BIAS, ws=self.lstm_ws,
                             h0=s0, c0=s0)
            xin, d = S["hseq"], H
        rows = b * Ts if self.per_step else b
        if self.per_step:
            v = xin                                                     # reshape fusion: every step's output (tf_util.py:26-27)
        else:
            ops.temporal_fusion_fwd(xin, self.fused, b, Ts, H, "last" if h.fusion == "state" else h.fusion)
            v = self.fused
        self._dropout = train and h.dropout_keep_prob > 0 and h.fusion != "state"
        if self._dropout:
            ops.dropout_fwd(v[:rows], self.dropped[:rows], self.drop_mask[:rows], h.dropout_keep_prob,
                            (self.step_count << 20) ^ 0x2545F4914F6CDD1D)
            v = self.dropped
        self._v = v
        if H != C:
            head = "fc_convert" if h.fusion == "state" else "output_fc"
            ops.gemm(v, P[sc + head + "_w"], self.logits, rows, C, H, bias=P[sc + head + "_b"])
        elif v is not self.logits:
            self.logits[:rows].copy_(v[:rows])
        self._rows = rows
        return rows

    def forward(self, frames_u8, words, mean_bgr=None, crop_y=None, crop_x=None, mirror=None, resize=None):
        """sess.run(model.logits, fdict) for the two-pipeline model.  words: device float32 [clips * fpc, in_dim]."""
        n, b1 = self.enc.feed_u8(frames_u8, mean_bgr, crop_y, crop_x, mirror, resize)
        self.enc._forward(n, b1, train=False)
        rows = self._head_forward(words, b1, train=False)
        return self.logits[:rows]

    # ---- backward ------------------------------------------------------------------------------------------------------------------
    def _head_backward(self, b1):
        h, P, G, sc, sw = self.h, self.P, self.G, self.scope2, self.small_ws
        b = self._b
        T, Ts, H, C, E = h.fpc, self.Ts, h.lstm_hidden, h.num_classes, self.E
        rows = self._rows
        d = self.dlogits
        if H != C:
            head = "fc_convert" if h.fusion == "state" else "output_fc"
            ops.gemm(self._v, self.dlogits, G[sc + head + "_w"], H, C, rows, transa=True)
            ops.colsum(self.dlogits, G[sc + head + "_b"], sw, rows, C)
            ops.gemm(self.dlogits, P[sc + head + "_w"], self.dpre, rows, H, C, transb=True)
            d = self.dpre
        if self._dropout:
            ops.dropout_bwd(d[:rows], self.drop_mask[:rows], self.dfused[:rows], h.dropout_keep_prob)
            d = self.dfused
        top = self.lstm[-1]
        if self.per_step:
            top["dout"][:rows].copy_(d[:rows])
        else:
            ops.temporal_fusion_bwd(d, top["dout"], b, Ts, H, "last" if h.fusion == "state" else h.fusion)
        has_state = self._s0 is not None
        for l in reversed(range(h.lstm_layers)):
            S = self.lstm[l]
            pre = sc + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
            K = P[pre + "kernel"]
            din = self.seq_dim if l == 0 else H
            xin = self._x if l == 0 else self.lstm[l - 1]["hseq"]
            ops.lstm_seq_bwd(S["dout"], K[din:], S["act"], S["cseq"], S["dz"], b, Ts, H, ws=self.lstm_ws, c0=self._s0,
                             dh0=S["dh0"] if has_state else None, dc0=S["dc0"] if has_state else None)
            ops.gemm(xin, S["dz"], G[pre + "kernel"], din, 4 * H, b * Ts, transa=True)
            ops.gemm(S["hprev"], S["dz"], G[pre + "kernel"][din:], H, 4 * H, b * Ts, transa=True)
            ops.colsum(S["dz"], G[pre + "bias"], sw, b * Ts, 4 * H)
            if has_state:                                               # c = h = state in EVERY layer: the gradients add up
                ops.eltwise2(S["dh0"], S["dc0"], S["dh0"], "add", count=b * H)
                if l == h.lstm_layers - 1:
                    self.dstate[:b].copy_(S["dh0"][:b])
                else:
                    ops.eltwise2(self.dstate, S["dh0"], self.dstate, "add", count=b * H)
            if l > 0:
                ops.gemm(S["dz"], K, self.lstm[l - 1]["dout"], b * Ts, H, 4 * H, transb=True, ldb=4 * H)
            elif self.dxseq is not None:
                ops.gemm(S["dz"], K, self.dxseq, b * Ts, din, 4 * H, transb=True, ldb=4 * H)
        # ---- into pipeline 1's output gradient, and the representation fc
        denc = self.enc.dlogits if self.rep is None else self.drep      # gradient of the (replicated) pipeline-1 vectors
        dx = None
        if h.input_fusion == "concat":
            # every word position of a clip carried a copy of its vector: sum the T column blocks
            ops.copy2d(self.dxseq, denc, b, self.enc_dim, src_ld=T * self.seq_dim, dst_ld=self.enc_dim)
            for t in range(1, T):
                ops.copy2d(self.dxseq[t:], self.tmp_enc, b, self.enc_dim, src_ld=T * self.seq_dim, dst_ld=self.enc_dim)
                ops.eltwise2(denc, self.tmp_enc, denc, "add", count=b * self.enc_dim)
            if self.dxfc is not None:
                ops.copy2d(self.dxseq[:, self.enc_dim:], self.dxfc, b * T, E, src_ld=self.seq_dim, dst_ld=E)
                dx = self.dxfc
        elif h.input_fusion == "ibias":
            ops.copy2d(self.dxseq, denc, b, E, src_ld=Ts * E, dst_ld=E)
            if self.dxfc is not None:
                ops.copy2d(self.dxseq[1:], self.dxfc, b, T * E, src_ld=Ts * E, dst_ld=T * E)
                dx = self.dxfc
        else:
            if self.state is not None:
                ops.gemm(self._enc_out, self.dstate, G[sc + "input_state_fc_w"], self.enc_dim, H, b, transa=True)
                ops.colsum(self.dstate, G[sc + "input_state_fc_b"], sw, b, H)
                ops.gemm(self.dstate, P[sc + "input_state_fc_w"], denc, b, self.enc_dim, H, transb=True)
            else:
                denc[:b].copy_(self.dstate[:b])
            if self.xfc is not None:
                dx = self.dxseq                                         # no fusion: d(sequence) is d(representation output)
        if self.xfc is not None:
            ops.gemm(self._words, dx, G[sc + "fc_convert_w"], h.in_dim, E, b * T, transa=True)
            ops.colsum(dx, G[sc + "fc_convert_b"], sw, b * T, E)
        if self.rep is not None:                                        # the tiles of the replicated batch add up
            ops.colsum(self.drep, self.enc.dlogits, sw, h.cpv_ratio, b1 * self.enc_dim)

    # ---- train step ------------------------------------------------------------------------------------------------------------------
    def train_step(self, frames_u8, words, onehot, lr, clip_norm=0.0, mean_bgr=None, crop_y=None, crop_x=None, mirror=None,
                   fetch=True, global_rows=None, resize=None):
        """sess.run([.., loss, .., optimizer], fdict): labels int32 one-hot [rows, classes], rows = clips (fusion avg | last |
        state) or clips * steps (fusion reshape: one row per time step, clip-major)."""
        if not self.training:
            raise VltfError("engine was built with training=False")
        n, b1 = self.enc.feed_u8(frames_u8, mean_bgr, crop_y, crop_x, mirror, resize)
        self.enc.step_count = self.step_count
        self.enc._forward(n, b1, train=True)
        self._words = words
        rows = self._head_forward(words, b1, train=True)
        if onehot.dtype != torch.int32 or tuple(onehot.shape) != (rows, self.h.num_classes):
            raise VltfError("labels must be int32 one-hot of shape (%d, %d)" % (rows, self.h.num_classes))
        world = self.dp.world if self.dp is not None else 1
        ops.fill(self.stats, 0.0)
        ops.softmax_xent(self.logits[:rows], onehot, self.dlogits, self.stats, 1.0 / (global_rows or rows * world), self.loss_rows)
        self._head_backward(b1)
        # the decoder's chunk is issued by _OffsetReduce together with the encoder's first one, i.e. AFTER the encoder's LSTM
        # backward: the cluster-form recurrence needs every CU and must not spin under an RCCL kernel that holds some
        self.enc.dp = _OffsetReduce(self.dp, self.g, self.enc.g, first=self.grad_chunks[0] if self.specs2 else None) \
            if self.dp is not None else None
        self.enc._backward(n, b1)
        self.enc.dp = None
        return self._finish_step(rows, lr, clip_norm, fetch)

    def train_step_empty(self, lr, clip_norm=0.0, fetch=True):
        """This rank's shard of the global batch is empty (fewer items than ranks in a short last batch): contribute zero gradients
        to the exchange and apply the same update as every other rank (LRCNEngine.train_step_empty)."""
        if self.dp is None:
            raise VltfError("train_step_empty is a data-parallel call")
        ops.fill(self.g, 0.0)
        ops.fill(self.stats, 0.0)
        for lo, cnt in self.grad_chunks:
            self.dp.reduce_async(self.g, lo, cnt)
        return self._finish_step(0, lr, clip_norm, fetch)

    def _finish_step(self, rows, lr, clip_norm, fetch):
        if self.dp is not None:
            self.dp.wait()
        ops.sumsq(self.g, self.ss, self.small_ws)
        self.step_count += 1
        if self.optimizer == "adam":
            ops.adam_apply(self.w, self.g, self.adam_m, self.adam_v, lr, self.step_count, clip_norm, self.ss, 1.0)
        else:
            ops.sgd_apply(self.w, self.g, lr, clip_norm, self.ss, 1.0)
        if not fetch:
            return None
        torch.cuda.synchronize(self.dev)
        self.check_status()
        st = self.stats.cpu().numpy()
        return {"loss": float(st[0]) / max(rows, 1), "accuracy": float(st[1]) / max(rows, 1),
                "grad_norm": math.sqrt(float(self.ss.item())), "rows": rows, "loss_sum": float(st[0]), "correct": float(st[1])}


class _OffsetReduce:
    """Lets pipeline 1's backward issue its gradient chunks on the shared flat buffer (its own `g` is a slice of it)."""

    def __init__(self, dp, whole, part, first=None):
        self.dp, self.world = dp, dp.world
        self.whole, self.base = whole, (part.data_ptr() - whole.data_ptr()) // 4
        self.first = first                      # (offset, count) of a chunk of `whole` to send ahead of the first chunk of `part`

    def reduce_async(self, flat, offset, count):
        if self.first is not None:
            self.dp.reduce_async(self.whole, *self.first)
            self.first = None
        self.dp.reduce_async(self.whole, self.base + offset, count)

    def wait(self):
        self.dp.wait()
