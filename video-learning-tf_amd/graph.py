"""GraphEngine: the reference's Model for ANY ordered list of pipelines (models/model.py:18-162).

Model.__init__ builds the pipelines in the order of the YAML list; an input of a pipeline is a dataset tag (a placeholder fed
from that dataset) or the output of an earlier pipeline; the LAST pipeline's output is the logits the loss is taken on
(model.py:157-162).  Model.build_pipeline (model.py:18-155) gives every pipeline the same five stages, each optional:

  inputs -> [input_fusion avg | maximum | concat | ibias  (apply_tensor_list_fusion, tf_util.py:136-179)]
         -> representation dcnn | nop | fc                (vectorizer.py: AlexNet tower / identity / convert_dim_fc "fc_convert")
         -> [early frame fusion                           (aggregate_clip_vectors, tf_util.py:126-133)]
         -> classifier None | fc | lstm                   (None: a feature pipeline, model.py:110-112; lstm: without input_fusion a
                                                           SECOND input is the LSTM's state vector, model.py:128-134)
         -> [late frame fusion]

Here a pipeline is a PipeNode: an optional AlexNet tower (an LRCNEngine built as a feature pipeline: conv stack + fc6..fc8 and
their backward, nothing else) and the generic stages around it, all on the C-ABI kernels of vltf_amd.ops.  Examples it builds:
the two-stream LRCN (dcnn on `main`, dcnn on `aux`, a third pipeline fusing them by avg | maximum | concat into an LSTM), the
encoder-decoder of BASELINE config 4 (vltf_amd.composed.ComposedEngine is this class behind its old two-pipeline interface),
feature pipelines consumed by several later ones (their gradients add up).

Variables: with more than one pipeline every variable is scoped "<pipeline>/<tf name>" (in the reference a second dcnn gets TF's
automatic "dcnn_1/" scope and a second LSTM / fc_convert cannot be created at all: tf.get_variable refuses the duplicate name).
ALL parameters live in one flat buffer ordered last pipeline first and, inside a pipeline, head before tower = the order backward
produces the gradients; global-norm clip, update and the data-parallel exchange run over the whole buffer as for one pipeline.
A pipeline the last one does not depend on is never evaluated by sess.run(logits) and is not built (a warning says so).

The oracle of this class is oracle.lrcn_oracle.model_forward / model_backward (cross-checked against torch autograd)."""
import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import ops
from ._ffi import VltfError
from .engine import FORGET_BIAS, LRCNEngine, NetConfig, param_specs


@dataclass
class PipelineSpec:
    """One entry of `network: pipelines:` (settings_.py:167-208)."""
    name: str
    input: List[str]                                # dataset tags and / or names of earlier pipelines
    representation: str = "nop"                     # dcnn | nop | fc
    frame_encoding_layer: Optional[str] = None      # dcnn
    fc_output_dim: Optional[int] = None             # fc
    classifier: Optional[str] = None                # None | fc | lstm
    lstm_params: Optional[Tuple[int, int, str]] = None   # (hidden, layers, fusion avg | last | reshape | state)
    frame_fusion: Optional[Tuple[str, str]] = None  # (early | late | none, avg | last | reshape)
    input_fusion: Optional[str] = None              # avg | maximum | concat | ibias


@dataclass
class DatasetInfo:
    """What the graph needs to know about the dataset behind a tag (dataset_.py: the .size sidecar decides mode, fpc, cpv)."""
    mode: str                                       # "video" (frames) | "vectors"
    fpc: int
    cpv: int
    max_clips: int                                  # most clips of this dataset in one batch
    image_shape: Optional[Tuple[int, int, int]] = None
    dim: Optional[int] = None                       # vectors


@dataclass
class _Src:
    kind: str               # "video" | "vectors" | "pipe"
    ref: object             # dataset tag, or the producing PipeNode
    dim: int
    cpv: int
    fpc: int
    max_rows: int
    rows: int = 0
    tensor: object = None


def _trunc_normal(rng, shape, sd):
    v = rng.standard_normal(shape)
    bad = np.abs(v) > 2.0
    while bad.any():
        v[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(v) > 2.0
    return (v * sd).astype(np.float32)


class PipeNode:
    """One pipeline of the graph (Model.build_pipeline, model.py:18-155)."""

    def __init__(self, g, spec: PipelineSpec, srcs, index):
        self.g, self.spec, self.srcs, self.index = g, spec, srcs, index
        self.scope = spec.name + "/" if g.scoped else ""
        s, C = spec, g.num_classes
        ftype, fmethod = s.frame_fusion if s.frame_fusion else (None, None)
        if ftype == "none":
            ftype = None
        if ftype is not None and fmethod not in ("avg", "last", "reshape"):
            raise VltfError("Undefined frame fusion type : %s" % fmethod)                 # apply_temporal_fusion, tf_util.py:28-29
        if s.classifier is None and ftype == "late":
            raise VltfError("Specified late fusion with no classifier selected")          # model.py:36-37
        if s.classifier not in (None, "fc", "lstm"):
            raise VltfError("Undefined classifier [%s]" % s.classifier)
        if s.representation not in ("dcnn", "nop", "fc"):
            raise VltfError("Undefined representation [%s]" % s.representation)
        if s.input_fusion not in (None, "avg", "maximum", "concat", "ibias"):
            raise VltfError("Unknown fusion method: [%s]" % s.input_fusion)               # tf_util.py:178-179
        self.cpv_out = srcs[-1].cpv            # model.py:44-59: `cpv` is the loop variable = the LAST input's, whatever the fusion does
        # ---- stage 0: the pipeline's main tensor after the optional input fusion: (dim, fpc, max rows) ----------------------------
        self.tower = None
        self.fusion = s.input_fusion
        self.state_src = None
        if s.representation == "dcnn":
            # the placeholder(s) hold frames; DCNN.build wants rank-3 items (vectorizer.py:47)
            seq = srcs if self.fusion else srcs[:1]
            if any(x.kind != "video" for x in seq):
                raise VltfError("The [dcnn] vectorizer required input tensors of rank [3], but pipeline [%s] feeds it vectors" % s.name)
            if self.fusion not in (None, "avg", "maximum"):
                raise VltfError("input_fusion [%s] of frame tensors is not built (avg | maximum of equally shaped frames only)" % self.fusion)
            shapes = {tuple(g.datasets[x.ref].image_shape) for x in seq}
            if len(shapes) != 1 or len({(x.fpc, x.max_rows) for x in seq}) != 1:
                raise VltfError("pipeline [%s]: the fused frame datasets differ in shape / frames per clip" % s.name)
            self.fpc = seq[0].fpc
            cfg = NetConfig(image_shape=shapes.pop(), num_classes=C, fpc=self.fpc, frame_encoding_layer=s.frame_encoding_layer,
                            classifier="none", optimizer=g.optimizer, conv_math=g.conv_math)
            self.tower_cfg = cfg
            self.x_dim, self.max_rows0 = cfg.encode_dim(), seq[0].max_rows
        elif self.fusion in ("avg", "maximum"):
            if len({(x.dim, x.max_rows) for x in srcs}) != 1:
                raise VltfError("pipeline [%s]: input_fusion %s needs equally shaped inputs (widths / rows %s)" %
                                (s.name, self.fusion, [(x.dim, x.max_rows) for x in srcs]))
            self.x_dim, self.fpc, self.max_rows0 = srcs[0].dim, srcs[0].fpc, srcs[0].max_rows
        elif self.fusion in ("concat", "ibias"):
            if len(srcs) != 2:
                raise VltfError("pipeline [%s]: input_fusion %s takes exactly two inputs (tf_util.py:137-140,184)" % (s.name, self.fusion))
            a, b = srcs
            self.ratio = int(a.cpv / b.cpv)
            if self.fusion == "concat" and self.ratio == 1:          # tf.concat(inputs, axis=1), tf_util.py:147-148
                if a.max_rows != b.max_rows:
                    raise VltfError("input_fusion concat at clips-per-video ratio 1 concatenates row by row: the inputs of pipeline "
                                    "[%s] have %d and %d rows per batch; it needs a ratio > 1 to tile a per-clip vector over a "
                                    "sequence" % (s.name, a.max_rows, b.max_rows))
                self.x_dim, self.fpc, self.max_rows0 = a.dim + b.dim, a.fpc, a.max_rows
            elif self.fusion == "concat":                            # vec_seq_concat, vector first (tf_util.py:99-124,150-152)
                self._check_aux_rows(a, b)
                self.x_dim, self.fpc, self.max_rows0 = a.dim + b.dim, a.fpc, a.max_rows
            else:                                                    # ibias: the vector becomes time step 0 (tf_util.py:154-176)
                if a.dim != b.dim:
                    raise VltfError("input_fusion ibias needs equal widths (the sequence %d, the vector %d)" % (a.dim, b.dim))
                self._check_aux_rows(a, b)
                self.x_dim, self.fpc, self.max_rows0 = a.dim, a.fpc + 1, a.max_rows // a.fpc * (a.fpc + 1)
        else:
            self.x_dim, self.fpc, self.max_rows0 = srcs[0].dim, srcs[0].fpc, srcs[0].max_rows
        if s.representation != "dcnn" and any(x.kind == "video" for x in (srcs if self.fusion else srcs[:1])):
            # a frame placeholder is rank 4: convert_dim_fc / the LSTM's reshape cannot take it (only DCNN.build can)
            raise VltfError("pipeline [%s]: representation %s cannot take the frame dataset [%s] (representation dcnn does)" %
                            (s.name, s.representation, [x.ref for x in srcs if x.kind == "video"][0]))
        if self.fusion is None and len(srcs) > 1:
            if s.classifier != "lstm":
                raise VltfError("pipeline [%s] has %d inputs but neither an input_fusion nor an LSTM classifier that would take the "
                                "second one as its state" % (s.name, len(srcs)))
            if len(srcs) != 2:
                raise VltfError("pipeline [%s]: an LSTM takes one state input (tf_util.py:184: too many values to unpack)" % s.name)
            self.state_src = srcs[1]
            if self.state_src.kind == "video":
                raise VltfError("pipeline [%s]: the LSTM's state input must be vectors or a pipeline output, [%s] holds frames" %
                                (s.name, self.state_src.ref))
            self.state_ratio = int(srcs[0].cpv / srcs[1].cpv)
            if self.state_ratio < 1 or srcs[1].max_rows * self.state_ratio * self.fpc != self.max_rows0:
                raise VltfError("pipeline [%s]: the state input has %d rows per batch, the sequence %d clips at clips-per-video ratio "
                                "%d" % (s.name, srcs[1].max_rows, self.max_rows0 // self.fpc, self.state_ratio))
        # ---- stage 1: representation ------------------------------------------------------------------------------------------------
        dim = self.x_dim
        self.rep_fc = s.representation == "fc" and s.fc_output_dim != dim
        if s.representation == "fc":
            if not s.fc_output_dim:
                raise VltfError("pipeline [%s]: representation fc needs fc_output_dim" % s.name)
            dim = int(s.fc_output_dim)
        self.feat_dim = dim
        # ---- stage 2: early fusion ---------------------------------------------------------------------------------------------------
        self.early = ftype == "early" and self.fpc > 1 and fmethod != "reshape"      # `reshape` of [B, T, d] back to [B*T, d]: identity
        self.fmethod = fmethod
        out_fpc = 1 if (ftype == "early" and self.fpc > 1) else self.fpc              # model.py:103-106
        rows = self.max_rows0 // self.fpc if self.early else self.max_rows0
        # ---- stage 3: classifier ----------------------------------------------------------------------------------------------------------
        self.cls = s.classifier
        self.late = False
        if self.cls is None:
            self.out_dim, self.fpc_out, self.max_rows = dim, out_fpc, rows
        else:
            if self.cls == "fc":
                self.cls_fc = dim != C
                if self.cls_fc and self.rep_fc:
                    raise VltfError("Variable %sfc_convert_w already exists (representation fc and classifier fc of one pipeline)" % self.scope)
            else:
                if self.fpc == 1:
                    raise VltfError("The LSTM classifier requires an fpc greater than 1")                     # model.py:121
                if ftype is not None:
                    raise VltfError("The LSTM classifier should be used only with [none] fusion, but it's [%s]" % ftype)
                self.H, self.L, self.lfusion = int(s.lstm_params[0]), int(s.lstm_params[1]), s.lstm_params[2]
                if self.lfusion not in ("avg", "last", "reshape", "state"):
                    raise VltfError("Undefined frame fusion type : %s" % self.lfusion)
                self.per_step = self.lfusion == "reshape"
                self.head_name = "fc_convert" if self.lfusion == "state" else "output_fc"
                if self.H != C and self.head_name == "fc_convert" and self.rep_fc:
                    raise VltfError("Variable %sfc_convert_w already exists (representation fc and LSTM fusion state)" % self.scope)
                rows = rows if self.per_step else rows // self.fpc
            self.late = ftype == "late" and self.fpc > 1 and fmethod != "reshape"
            if self.late:
                if self.cls == "lstm" or rows % self.fpc:
                    raise VltfError("pipeline [%s]: late fusion needs one row per frame" % s.name)
                self.pre_late_rows = rows
                rows //= self.fpc
            self.out_dim, self.fpc_out, self.max_rows = C, 1, rows                        # model.py:153

    @staticmethod
    def _check_aux_rows(a, b):
        ratio = int(a.cpv / b.cpv)
        if ratio < 1 or b.max_rows * ratio * a.fpc != a.max_rows:
            raise VltfError("the vector input has %d rows per batch but the sequence has %d clips at clips-per-video ratio %d" %
                            (b.max_rows, a.max_rows // a.fpc, ratio))

    # ---- variables ---------------------------------------------------------------------------------------------------------------------
    def head_specs(self):
        """[(name, shape)] of everything but the tower, in backward order."""
        sc, C, specs = self.scope, self.g.num_classes, []
        if self.cls == "fc" and self.cls_fc:
            specs += [(sc + "fc_convert_w", (self.feat_dim, C)), (sc + "fc_convert_b", (C,))]
        if self.cls == "lstm":
            H = self.H
            if H != C:
                specs += [(sc + self.head_name + "_w", (H, C)), (sc + self.head_name + "_b", (C,))]
            dims = [self.feat_dim] + [H] * (self.L - 1)
            for l in reversed(range(self.L)):
                pre = sc + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
                specs += [(pre + "kernel", (dims[l] + H, 4 * H)), (pre + "bias", (4 * H,))]
            if self.state_src is not None and self.state_src.dim != H:
                specs += [(sc + "input_state_fc_w", (self.state_src.dim, H)), (sc + "input_state_fc_b", (H,))]
        if self.rep_fc:
            specs += [(sc + "fc_convert_w", (self.x_dim, self.feat_dim)), (sc + "fc_convert_b", (self.feat_dim,))]
        return specs

    def tower_specs(self):
        return [(self.scope + n, s) for n, s in param_specs(self.tower_cfg)] if self.spec.representation == "dcnn" else []

    # ---- buffers ---------------------------------------------------------------------------------------------------------------------------
    def allocate(self, flat_w, flat_g, off):
        """Views of the flat parameter / gradient buffers and every activation buffer; returns the offset behind this pipeline."""
        g, dev, tr = self.g, self.g.dev, self.g.training

        def buf(*shape, dtype=torch.float32):
            return torch.empty(shape, dtype=dtype, device=dev)
        self.P, self.G = {}, {}
        self.head_off = off
        for name, shp in self.head_specs():
            n = int(np.prod(shp))
            self.P[name] = flat_w[off:off + n].view(shp)
            if tr:
                self.G[name] = flat_g[off:off + n].view(shp)
            off += n
        self.head_cnt = off - self.head_off
        if self.spec.representation == "dcnn":
            n1 = sum(int(np.prod(s)) for _, s in param_specs(self.tower_cfg))
            self.tower = LRCNEngine(self.tower_cfg, self.max_rows0 // self.fpc, str(dev), tr, dp=None,
                                    flat=(flat_w[off:off + n1], flat_g[off:off + n1] if tr else None))
            self.tower_off = off
            for n, _ in param_specs(self.tower_cfg):
                self.P[self.scope + n] = self.tower.P[n]
                if tr:
                    self.G[self.scope + n] = self.tower.G[n]
            if self.fusion:               # avg | maximum of several frame datasets: the other inputs' prepared frames (x0's layout)
                self.x0_others = [torch.zeros_like(self.tower.x0) for _ in self.srcs[1:]]
            off += n1
        R0, D0 = self.max_rows0, self.x_dim
        # is a gradient w.r.t. the stage-0 tensor wanted?  (a pipeline input other than the LSTM's state vector)
        self.primary_pipe = self.tower is None and any(x.kind == "pipe" for x in self.srcs if x is not self.state_src)
        self.wants_feat = self.tower is not None or self.rep_fc or self.primary_pipe     # ... w.r.t. the classifier's input?
        if self.tower is None and self.fusion:
            self.xf = buf(R0, D0)
        if self.rep_fc:
            self.xrep = buf(R0, self.feat_dim)
            if tr and self.primary_pipe:
                self.dx0 = buf(R0, D0)
        if self.early:
            self.xearly = buf(R0 // self.fpc, self.feat_dim)
            if tr:
                self.dfeat_full = buf(R0, self.feat_dim)
        rows_cls = R0 // self.fpc if self.early else R0
        C = g.num_classes
        sw = 64 * max(1024, C, self.feat_dim, D0)
        if self.cls == "fc":
            self.cls_out = buf(rows_cls, C) if self.cls_fc else None
            if tr and self.cls_fc:
                self.dcls_in = buf(rows_cls, self.feat_dim)
        elif self.cls == "lstm":
            H, T = self.H, self.fpc
            B = R0 // T
            self.lstm = []
            for l in range(self.L):
                S = dict(gx=buf(R0, 4 * H), act=buf(R0, 4 * H), cseq=buf(R0, H), hseq=buf(R0, H), hprev=buf(R0, H))
                if tr:
                    S.update(dz=buf(R0, 4 * H), dout=buf(R0, H), dh0=buf(B, H), dc0=buf(B, H))
                self.lstm.append(S)
            self.lstm_ws = ops.lstm_seq_ws(B, T, H, dev) if H <= 1024 else None
            if self.lstm_ws is None:
                raise VltfError("GraphEngine: LSTM hidden size above 1024 is not built")
            r = R0 if self.per_step else B
            self.fused = buf(r, H) if not self.per_step else None
            self.dropped = buf(r, H)
            self.drop_mask = buf(r, H, dtype=torch.uint8)
            self.cls_out = buf(r, C) if H != C else None
            if self.state_src is not None:
                sd = self.state_src.dim
                self.rep_state = buf(B, sd) if self.state_ratio > 1 else None
                self.state = buf(B, H) if sd != H else None
                if tr:
                    self.dstate = buf(B, H)
                    self.drep_state = buf(B, sd) if (sd != H and self.state_src.kind == "pipe") else None
            if tr:
                self.dpre, self.dfused = buf(r, H), buf(r, H)
                self.dxseq = buf(R0, self.feat_dim)
            sw = max(sw, 64 * 4 * H, 64 * H)
        if self.late:
            self.late_out = buf(self.max_rows, C)
            if tr:
                self.dlate = buf(self.pre_late_rows, C)
        if tr and self.fusion in ("concat", "ibias") and self.tower is None:
            a, b = self.srcs
            self.daux = buf(b.max_rows * max(1, getattr(self, "ratio", 1)), b.dim)       # gradient of the replicated vector input
            self.dtmp = buf(b.max_rows * max(1, getattr(self, "ratio", 1)), b.dim)
            sw = max(sw, 64 * b.max_rows * b.dim)
        if self.fusion in ("concat", "ibias") and self.tower is None and getattr(self, "ratio", 1) > 1:
            self.rep_aux = buf(self.srcs[1].max_rows * self.ratio, self.srcs[1].dim)
        if self.state_src is not None:
            sw = max(sw, 64 * self.state_src.max_rows * self.state_src.dim)
        self.small_ws = buf(sw)
        if tr:
            self.dout = buf(self.max_rows, self.out_dim)       # d(loss) / d(output): the loss (last pipeline) or the consumers write it
            self.din = [buf(x.max_rows, x.dim) if x.kind == "pipe" else None for x in self.srcs]
        return off

    # ---- forward ---------------------------------------------------------------------------------------------------------------------------
    def _src_tensor(self, x, feeds):
        if x.kind == "pipe":
            x.rows, x.tensor = x.ref.rows, x.ref.out
        else:
            t = feeds[x.ref]
            if x.kind == "vectors":
                if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != x.dim or t.shape[0] > x.max_rows or not t.is_contiguous():
                    raise VltfError("dataset [%s]: vectors must be contiguous float32 [rows <= %d, %d], got %s %s" %
                                    (x.ref, x.max_rows, x.dim, t.dtype, tuple(t.shape)))
                x.rows, x.tensor = t.shape[0], t
        return x

    def forward(self, feeds, train):
        g, s, P, sc, C = self.g, self.spec, self.P, self.scope, self.g.num_classes
        srcs = self.srcs
        # ---- stage 0 / 1 (dcnn): frames -> tower features
        if self.tower is not None:
            tw = self.tower
            seq = srcs if self.fusion else srcs[:1]
            if len(seq) > 1 and tw.c8:
                raise VltfError("input_fusion of frame datasets is built for the fp32 conv path only")
            n = b = None
            for i, x in enumerate(seq):        # each dataset's frames through input prep into x0; all but the last are set aside
                f = feeds[x.ref]
                ni, bi = tw.feed_u8(f["frames_u8"], f.get("mean_bgr"), f.get("crop_y"), f.get("crop_x"), f.get("mirror"), f.get("resize"))
                if n is not None and ni != n:
                    raise VltfError("pipeline [%s]: fused frame datasets gave %d and %d frames" % (s.name, n, ni))
                n, b = ni, bi
                if i + 1 < len(seq):
                    self.x0_others[i][:n].copy_(tw.x0[:n])
            if len(seq) > 1:                   # mean / maximum over the list (the zero halo stays zero under both)
                ops.fuse_n([t[:n] for t in self.x0_others] + [tw.x0[:n]], tw.x0[:n], self.fusion, count=tw.x0[:n].numel())
            tw.step_count = g.step_count
            tw._forward(n, b, train)
            self._nb = (n, b)
            x, rows = tw.logits, n
            for xs in srcs[1:]:
                self._src_tensor(xs, feeds)
        else:
            for xs in srcs:
                self._src_tensor(xs, feeds)
            a = srcs[0]
            if self.fusion in ("avg", "maximum"):
                rows = a.rows
                if any(xs.rows != rows for xs in srcs):
                    raise VltfError("pipeline [%s]: fused inputs have %s rows" % (s.name, [xs.rows for xs in srcs]))
                ops.fuse_n([xs.tensor for xs in srcs], self.xf, self.fusion, count=rows * self.x_dim)
                x = self.xf
            elif self.fusion == "concat" and self.ratio == 1:
                b_ = srcs[1]
                rows = a.rows
                if b_.rows != rows:
                    raise VltfError("pipeline [%s]: concat of %d and %d rows" % (s.name, a.rows, b_.rows))
                ops.copy2d(a.tensor, self.xf, rows, a.dim, src_ld=a.dim, dst_ld=self.x_dim)
                ops.copy2d(b_.tensor, self.xf[:, a.dim:], rows, b_.dim, src_ld=b_.dim, dst_ld=self.x_dim)
                x = self.xf
            elif self.fusion in ("concat", "ibias"):
                b_ = srcs[1]
                T = a.fpc
                clips = a.rows // T
                aux = self._replicate(b_, self.ratio, clips, getattr(self, "rep_aux", None))
                if self.fusion == "concat":        # every clip's vector in front of each of its T sequence rows (vecfirst)
                    for t in range(T):
                        ops.copy2d(aux, self.xf[t:], clips, b_.dim, src_ld=b_.dim, dst_ld=T * self.x_dim)
                    ops.copy2d(a.tensor, self.xf[:, b_.dim:], a.rows, a.dim, src_ld=a.dim, dst_ld=self.x_dim)
                    rows = a.rows
                else:                              # the vector is time step 0 of every clip
                    Ts = T + 1
                    ops.copy2d(aux, self.xf, clips, a.dim, src_ld=a.dim, dst_ld=Ts * a.dim)
                    ops.copy2d(a.tensor, self.xf[1:], clips, T * a.dim, src_ld=T * a.dim, dst_ld=Ts * a.dim)
                    rows = clips * Ts
                x = self.xf
            else:
                x, rows = a.tensor, a.rows
        if rows % self.fpc:
            raise VltfError("pipeline [%s]: %d rows are not whole clips of %d" % (s.name, rows, self.fpc))
        self._x0, self._rows0 = x, rows
        # ---- stage 1: representation fc
        if self.rep_fc:
            ops.gemm(x, P[sc + "fc_convert_w"], self.xrep, rows, self.feat_dim, self.x_dim, bias=P[sc + "fc_convert_b"])
            x = self.xrep
        # ---- stage 2: early fusion
        if self.early:
            ops.temporal_fusion_fwd(x, self.xearly, rows // self.fpc, self.fpc, self.feat_dim, self.fmethod)
            x, rows = self.xearly, rows // self.fpc
        self._xcls, self._rows_cls = x, rows
        # ---- stage 3: classifier
        if self.cls == "fc":
            if self.cls_fc:
                ops.gemm(x, P[sc + "fc_convert_w"], self.cls_out, rows, C, self.feat_dim, bias=P[sc + "fc_convert_b"])
                x = self.cls_out
        elif self.cls == "lstm":
            x, rows = self._lstm_forward(x, rows, train)
        # ---- stage 4: late fusion
        if self.late:
            ops.temporal_fusion_fwd(x, self.late_out, rows // self.fpc, self.fpc, C, self.fmethod)
            self._pre_late_rows = rows
            x, rows = self.late_out, rows // self.fpc
        self.out, self.rows = x, rows
        return x

    def _replicate(self, src, ratio, want_rows, scratch):
        """replicate_auxilliary_tensor (tf_util.py:182-192): the WHOLE batch of vectors repeated `ratio` times in sequence."""
        if src.rows * ratio != want_rows:
            raise VltfError("pipeline [%s]: %d vectors x clips-per-video ratio %d do not pair up with %d clips" %
                            (self.spec.name, src.rows, ratio, want_rows))
        if ratio <= 1:
            return src.tensor
        ops.copy2d(src.tensor, scratch, ratio, src.rows * src.dim, src_ld=0, dst_ld=src.rows * src.dim)
        return scratch

    def _lstm_forward(self, x, rows, train):
        g, P, sc, C, H, T = self.g, self.P, self.scope, self.g.num_classes, self.H, self.fpc
        b = rows // T
        s0 = None
        if self.state_src is not None:                  # model.py:128-134, lstm.py:34-42,74-77
            st = self._replicate(self.state_src, self.state_ratio, b, self.rep_state)
            self._state_in = st
            s0 = st
            if self.state is not None:
                ops.gemm(st, P[sc + "input_state_fc_w"], self.state, b, H, self.state_src.dim, bias=P[sc + "input_state_fc_b"])
                s0 = self.state
        self._s0, self._b = s0, b
        xin, d = x, self.feat_dim
        for l, S in enumerate(self.lstm):
            pre = sc + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
            K = P[pre + "kernel"]
            ops.gemm(xin, K, S["gx"], rows, 4 * H, d, bias=P[pre + "bias"], ws=g.ws)
            ops.lstm_seq_fwd(S["gx"], K[d:], S["act"], S["cseq"], S["hseq"], S["hprev"], b, T, H, FORGET_BIAS, ws=self.lstm_ws, h0=s0, c0=s0)
            xin, d = S["hseq"], H
        r = rows if self.per_step else b
        if self.per_step:
            v = xin                                                     # reshape fusion: every step's output (tf_util.py:26-27)
        else:
            ops.temporal_fusion_fwd(xin, self.fused, b, T, H, "last" if self.lfusion == "state" else self.lfusion)
            v = self.fused
        self._dropout = train and g.dropout_keep_prob > 0 and self.lfusion != "state"       # lstm.py:80-93
        if self._dropout:
            ops.dropout_fwd(v[:r], self.dropped[:r], self.drop_mask[:r], g.dropout_keep_prob,
                            (g.step_count << 20) ^ (0x2545F4914F6CDD1D + 0x9E3779B9 * self.index))
            v = self.dropped
        self._v = v
        if H != C:
            ops.gemm(v, P[sc + self.head_name + "_w"], self.cls_out, r, C, H, bias=P[sc + self.head_name + "_b"])
            v = self.cls_out
        return v, r

    # ---- backward --------------------------------------------------------------------------------------------------------------------------
    def backward(self):
        """Consumes self.dout[:rows]; writes the gradients of this pipeline's variables and adds its input gradients into the producers'
        dout."""
        g, P, G, sc, C, sw = self.g, self.P, self.G, self.scope, self.g.num_classes, self.small_ws
        d, rows = self.dout, self.rows
        if self.late:
            ops.temporal_fusion_bwd(d, self.dlate, rows, self.fpc, C, self.fmethod)
            d, rows = self.dlate, self._pre_late_rows
        if self.cls == "fc":
            if self.cls_fc:
                ops.gemm(self._xcls, d, G[sc + "fc_convert_w"], self.feat_dim, C, rows, transa=True)
                ops.colsum(d, G[sc + "fc_convert_b"], sw, rows, C)
                if self.wants_feat:
                    ops.gemm(d, P[sc + "fc_convert_w"], self.dcls_in, rows, self.feat_dim, C, transb=True)
                d = self.dcls_in
        elif self.cls == "lstm":
            d, rows = self._lstm_backward(d)
        if self.wants_feat:
            if self.early:
                ops.temporal_fusion_bwd(d, self.dfeat_full, rows, self.fpc, self.feat_dim, self.fmethod)
                d, rows = self.dfeat_full, rows * self.fpc
            if self.rep_fc:
                ops.gemm(self._x0, d, G[sc + "fc_convert_w"], self.x_dim, self.feat_dim, rows, transa=True)
                ops.colsum(d, G[sc + "fc_convert_b"], sw, rows, self.feat_dim)
                if self.primary_pipe:
                    ops.gemm(d, P[sc + "fc_convert_w"], self.dx0, rows, self.x_dim, self.feat_dim, transb=True)
                    d = self.dx0
        # Every gradient of the head chunk is queued now -- the classifier's, its LSTM launches, AND the `representation: fc` variables
        # (fc_convert_w / _b belong to the same chunk, head_specs): only here may the chunk's all-reduce be issued.  Round 3 called this
        # before the representation-fc block: under data parallelism the exchange then read those two gradients before they were written.
        g._head_done(self)
        if self.wants_feat:
            if self.tower is not None:
                n, b = self._nb
                tw = self.tower
                tw.dlogits[:n].copy_(d[:n])       # d(features); the tower applies the encode layer's ReluGrad itself
                tw.dp = g._tower_dp(self)
                tw._backward(n, b)
                tw.dp = None
            elif self.primary_pipe:
                self._input_grads(d, rows)
        if self.state_src is not None and self.state_src.kind == "pipe":
            self._route(1, self._dstate_out)

    def _lstm_backward(self, d):
        g, P, G, sc, C, H, T, sw = self.g, self.P, self.G, self.scope, self.g.num_classes, self.H, self.fpc, self.small_ws
        b = self._b
        rows = b * T
        r = rows if self.per_step else b
        if H != C:
            ops.gemm(self._v, d, G[sc + self.head_name + "_w"], H, C, r, transa=True)
            ops.colsum(d, G[sc + self.head_name + "_b"], sw, r, C)
            ops.gemm(d, P[sc + self.head_name + "_w"], self.dpre, r, H, C, transb=True)
            d = self.dpre
        if self._dropout:
            ops.dropout_bwd(d[:r], self.drop_mask[:r], self.dfused[:r], g.dropout_keep_prob)
            d = self.dfused
        top = self.lstm[-1]
        if self.per_step:
            top["dout"][:r].copy_(d[:r])
        else:
            ops.temporal_fusion_bwd(d, top["dout"], b, T, H, "last" if self.lfusion == "state" else self.lfusion)
        has_state = self._s0 is not None
        for l in reversed(range(self.L)):
            S = self.lstm[l]
            pre = sc + "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
            K = P[pre + "kernel"]
            din = self.feat_dim if l == 0 else H
            xin = self._xcls if l == 0 else self.lstm[l - 1]["hseq"]
            ops.lstm_seq_bwd(S["dout"], K[din:], S["act"], S["cseq"], S["dz"], b, T, H, ws=self.lstm_ws, c0=self._s0,
                             dh0=S["dh0"] if has_state else None, dc0=S["dc0"] if has_state else None)
            ops.gemm(xin, S["dz"], G[pre + "kernel"], din, 4 * H, rows, transa=True, ws=g.ws)
            ops.gemm(S["hprev"], S["dz"], G[pre + "kernel"][din:], H, 4 * H, rows, transa=True, ws=g.ws)
            ops.colsum(S["dz"], G[pre + "bias"], sw, rows, 4 * H)
            if has_state:                                               # c = h = state in EVERY layer: the gradients add up
                ops.eltwise2(S["dh0"], S["dc0"], S["dh0"], "add", count=b * H)
                if l == self.L - 1:
                    self.dstate[:b].copy_(S["dh0"][:b])
                else:
                    ops.eltwise2(self.dstate, S["dh0"], self.dstate, "add", count=b * H)
            if l > 0:
                ops.gemm(S["dz"], K, self.lstm[l - 1]["dout"], rows, H, 4 * H, transb=True, ldb=4 * H, ws=g.ws)
            elif self.wants_feat:
                ops.gemm(S["dz"], K, self.dxseq, rows, din, 4 * H, transb=True, ldb=4 * H, ws=g.ws)
        if has_state:
            ds, sd = self.dstate, self.state_src.dim
            to_pipe = self.state_src.kind == "pipe"
            if self.state is not None:                                  # input_state_fc (lstm.py:74-77)
                ops.gemm(self._state_in, self.dstate, G[sc + "input_state_fc_w"], sd, H, b, transa=True)
                ops.colsum(self.dstate, G[sc + "input_state_fc_b"], sw, b, H)
                if to_pipe:
                    ops.gemm(self.dstate, P[sc + "input_state_fc_w"], self.drep_state, b, sd, H, transb=True)
                    ds = self.drep_state
            if to_pipe:
                if self.state_ratio > 1:                                # the tiles of the replicated batch add up
                    ops.colsum(ds, self.din[1], sw, self.state_ratio, (b // self.state_ratio) * sd)
                    ds = self.din[1]
                self._dstate_out = ds
        return self.dxseq, rows

    def _input_grads(self, d, rows):
        """Gradient w.r.t. the fused stage-0 tensor -> the pipeline inputs' producers."""
        srcs = self.srcs
        a = srcs[0]
        if self.fusion in ("avg", "maximum"):
            dins = [self.din[i] if x.kind == "pipe" else None for i, x in enumerate(srcs)]
            ops.fuse_n_grad([x.tensor for x in srcs], d, dins, self.fusion, count=rows * self.x_dim)
            for i, x in enumerate(srcs):
                if x.kind == "pipe":
                    self._route(i, dins[i])
        elif self.fusion == "concat" and self.ratio == 1:
            b_ = srcs[1]
            if a.kind == "pipe":
                ops.copy2d(d, self.din[0], rows, a.dim, src_ld=self.x_dim, dst_ld=a.dim)
                self._route(0, self.din[0])
            if b_.kind == "pipe":
                ops.copy2d(d[:, a.dim:], self.din[1], rows, b_.dim, src_ld=self.x_dim, dst_ld=b_.dim)
                self._route(1, self.din[1])
        elif self.fusion in ("concat", "ibias"):
            b_ = srcs[1]
            T = a.fpc
            clips = a.rows // T
            if self.fusion == "concat":
                if b_.kind == "pipe":       # every sequence position of a clip carried a copy of its vector: sum the T column blocks
                    ops.copy2d(d, self.daux, clips, b_.dim, src_ld=T * self.x_dim, dst_ld=b_.dim)
                    for t in range(1, T):
                        ops.copy2d(d[t:], self.dtmp, clips, b_.dim, src_ld=T * self.x_dim, dst_ld=b_.dim)
                        ops.eltwise2(self.daux, self.dtmp, self.daux, "add", count=clips * b_.dim)
                if a.kind == "pipe":
                    ops.copy2d(d[:, b_.dim:], self.din[0], a.rows, a.dim, src_ld=self.x_dim, dst_ld=a.dim)
                    self._route(0, self.din[0])
            else:
                Ts = T + 1
                if b_.kind == "pipe":
                    ops.copy2d(d, self.daux, clips, a.dim, src_ld=Ts * a.dim, dst_ld=a.dim)
                if a.kind == "pipe":
                    ops.copy2d(d[1:], self.din[0], clips, T * a.dim, src_ld=Ts * a.dim, dst_ld=T * a.dim)
                    self._route(0, self.din[0])
            if b_.kind == "pipe":
                da = self.daux
                if self.ratio > 1:
                    ops.colsum(self.daux, self.din[1], self.small_ws, self.ratio, b_.rows * b_.dim)
                    da = self.din[1]
                self._route(1, da)
        elif a.kind == "pipe":
            self._route(0, d)

    def _route(self, i, d):
        """Adds (or, for the producer's only consumer edge, copies) d into the producer's dout."""
        x = self.srcs[i]
        prod = x.ref
        cnt = x.rows * x.dim
        if prod._dout_fresh:
            if d.data_ptr() != prod.dout.data_ptr():
                prod.dout.view(-1)[:cnt].copy_(d.reshape(-1)[:cnt])
            prod._dout_fresh = False
        else:
            ops.eltwise2(prod.dout, d, prod.dout, "add", count=cnt)


def init_params_for(specs, seed=0, stddev=0.05, well_scaled=False):
    """Reference initialisers for a variable list [(name, shape)] (alexnet.py:40-46, tf_util.py:44-45; BasicLSTMCell: glorot-uniform /
    zeros), drawn in list order from one generator."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in specs:
        if name.endswith("kernel"):
            lim = math.sqrt(6.0 / (shp[0] + shp[1]))
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = np.zeros(shp, np.float32)
        elif len(shp) > 1:
            out[name] = _trunc_normal(rng, shp, math.sqrt(2.0 / int(np.prod(shp[:-1]))) if well_scaled else stddev)
        else:
            out[name] = np.full(shp, 0.1, np.float32)
    return out


def model_specs(pipelines, datasets, num_classes):
    """[(name, shape)] of a model's variables in flat-buffer order -- the host-side plan of GraphEngine without a device (fixture
    generators, checkpoint tools)."""
    eng = object.__new__(GraphEngine)
    eng.dev, eng.training, eng.dp = torch.device("cpu"), False, None
    eng._plan(pipelines, datasets, num_classes)
    return list(eng.specs)


class GraphEngine:
    def __init__(self, pipelines: List[PipelineSpec], datasets: dict, num_classes: int, device="cuda:0", training=True, dp=None,
                 optimizer="sgd", dropout_keep_prob=0.0, conv_math="f32"):
        self.dev = torch.device(device)
        self._require_device()
        self.training, self.dp = training, dp
        self._plan(pipelines, datasets, num_classes, optimizer, dropout_keep_prob, conv_math)
        self._allocate()

    def _plan(self, pipelines, datasets, num_classes, optimizer="sgd", dropout_keep_prob=0.0, conv_math="f32"):
        """The graph and its variable list (self.nodes, self.specs): host logic only, no device (model_specs uses it alone)."""
        if not pipelines:
            raise VltfError("no pipeline defined")
        self.num_classes, self.optimizer, self.dropout_keep_prob, self.conv_math = int(num_classes), optimizer, float(dropout_keep_prob or 0.0), conv_math
        self.datasets = datasets
        self.scoped = len(pipelines) > 1
        self.step_count = 0
        names = [p.name for p in pipelines]
        if len(set(names)) != len(names) or any(n in datasets for n in names):
            raise VltfError("pipeline names must be unique and differ from the dataset tags: %s" % names)
        # the last pipeline defines the logits (model.py:161); only what it depends on is ever evaluated
        needed, stack = set(), [pipelines[-1].name]
        by_name = {p.name: p for p in pipelines}
        while stack:
            n = stack.pop()
            if n in needed:
                continue
            needed.add(n)
            stack += [i for i in by_name[n].input if i in by_name]
        self.skipped = [p.name for p in pipelines if p.name not in needed]
        self.nodes, self.by_name = [], {}
        for spec in pipelines:
            if spec.name not in needed:
                continue
            srcs = []
            for src in spec.input:
                if src in self.by_name:
                    nd = self.by_name[src]
                    srcs.append(_Src("pipe", nd, nd.out_dim, nd.cpv_out, nd.fpc_out, nd.max_rows))
                elif src in by_name:
                    raise VltfError("Input identifier [%s] of pipeline [%s] is a pipeline that has not been declared yet." % (src, spec.name))
                elif src in datasets:
                    ds = datasets[src]
                    dim = ds.dim if ds.mode == "vectors" else ds.image_shape[-1]
                    srcs.append(_Src(ds.mode, src, dim, ds.cpv, ds.fpc, ds.max_clips * ds.fpc))
                else:
                    raise VltfError("Could not find a dataset with the tag %s, required by the pipeline %s" % (src, spec.name))
            node = PipeNode(self, spec, srcs, len(self.nodes))
            self.nodes.append(node)
            self.by_name[spec.name] = node
        self.last = self.nodes[-1]
        if self.last.out_dim != self.num_classes:
            raise VltfError("the last pipeline [%s] ends with %d-wide rows, the loss needs num_classes = %d (a classifier is missing)" %
                            (self.last.spec.name, self.last.out_dim, self.num_classes))
        # ---- one flat parameter / gradient buffer, last pipeline first
        order = list(reversed(self.nodes))
        self.specs = []
        for nd in order:
            self.specs += nd.head_specs() + nd.tower_specs()
        dup = {n for n, _ in self.specs if [m for m, _ in self.specs].count(n) > 1}
        if dup:
            raise VltfError("Variable %s already exists" % sorted(dup)[0])

    def _allocate(self):
        order = list(reversed(self.nodes))
        training, optimizer = self.training, self.optimizer
        total = sum(int(np.prod(s)) for _, s in self.specs)
        dev = self.dev
        self.w = torch.zeros(total, device=dev)
        self.g = torch.zeros(total, device=dev) if training else None
        self.ws = torch.empty((64 << 20) // 4, device=dev)            # split-k slabs of the heads' GEMMs
        off = 0
        self.P, self.G = {}, {}
        for nd in order:
            off = nd.allocate(self.w, self.g, off)
            self.P.update(nd.P)
            self.G.update(nd.G)
        assert off == total
        # data-parallel chunks in the order backward completes them: per pipeline its head, then its tower's own chunk list
        self.grad_chunks = []
        for nd in order:
            if nd.head_cnt:
                self.grad_chunks.append((nd.head_off, nd.head_cnt))
            if nd.tower is not None:
                self.grad_chunks += [(nd.tower_off + lo, cnt) for lo, cnt in nd.tower.grad_chunks]
        if optimizer == "adam" and training:
            self.adam_m, self.adam_v = torch.zeros(total, device=dev), torch.zeros(total, device=dev)
        rows = self.last.max_rows
        self.stats = torch.zeros(2, device=dev)
        self.loss_rows = torch.zeros(2 * rows, device=dev)
        self.ss = torch.zeros(1, device=dev)
        self._skip = torch.zeros(1, dtype=torch.int32, device=dev)      # ops.step_guard: the optimizer launch's skip word
        self.small_ws = torch.empty(64 * 1024, device=dev)
        self._queued, self._pending_lstm = [], 0
        self.per_step = self.last.cls == "lstm" and self.last.per_step

    def _require_device(self):
        if self.dev.type != "cuda" or not torch.cuda.is_available():
            raise VltfError("GraphEngine needs a HIP device; there is no CPU fallback")
        torch.cuda.set_device(self.dev)

    def _sync(self):
        torch.cuda.synchronize(self.dev)

    # ---- parameters --------------------------------------------------------------------------------------------------------------------
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

    def init_params(self, seed=0, stddev=0.05, well_scaled=False):
        """Reference initialisers for every variable (alexnet.py:40-46, tf_util.py:44-45; BasicLSTMCell: glorot-uniform / zeros)."""
        return init_params_for(self.specs, seed, stddev, well_scaled)

    def get_params(self):
        self._sync()
        return {n: self.P[n].detach().cpu().numpy().copy() for n, _ in self.specs}

    def get_grads(self):
        self._sync()
        return {n: self.G[n].detach().cpu().numpy().copy() for n, _ in self.specs}

    OPT_PREFIX = LRCNEngine.OPT_PREFIX
    get_opt_state = LRCNEngine.get_opt_state
    load_opt_state = LRCNEngine.load_opt_state

    @property
    def cfg(self):
        """The optimizer-bearing view the checkpoint helpers of LRCNEngine read."""
        return self

    def check_status(self):
        ops.lstm_seq_check(*[nd.lstm_ws for nd in self.nodes if nd.cls == "lstm"])

    def logits_host(self):
        self._sync()
        self.check_status()
        return self.last.out[:self.last.rows].detach().cpu().numpy().copy()

    def pipeline_output_host(self, name):
        self._sync()
        nd = self.by_name[name]
        return nd.out[:nd.rows].detach().cpu().numpy().copy()

    # ---- the two executor calls ------------------------------------------------------------------------------------------------------
    def _forward(self, feeds, train):
        ops.set_conv_math(self.conv_math)
        for nd in self.nodes:
            nd.forward(feeds, train)
        return self.last.rows

    def forward(self, feeds):
        """sess.run(model.logits, fdict).  feeds: {dataset tag: dict(frames_u8=, mean_bgr=, crop_y=, crop_x=, mirror=, resize=) for a
        frame dataset | float32 device tensor [rows, dim] for a vectors dataset}.  Returns a device view [rows, classes]."""
        rows = self._forward(feeds, train=False)
        return self.last.out[:rows]

    def train_step(self, feeds, onehot, lr, clip_norm=0.0, fetch=True, global_rows=None):
        """sess.run([.., loss, .., optimizer], fdict): labels int32 one-hot [rows, classes] for the LAST pipeline's rows."""
        if not self.training:
            raise VltfError("engine was built with training=False")
        rows = self._forward(feeds, train=True)
        if onehot.dtype != torch.int32 or tuple(onehot.shape) != (rows, self.num_classes):
            raise VltfError("labels must be int32 one-hot of shape (%d, %d)" % (rows, self.num_classes))
        world = self.dp.world if self.dp is not None else 1
        ops.fill(self.stats, 0.0)
        last = self.last
        ops.softmax_xent(last.out[:rows], onehot, last.dout, self.stats, 1.0 / (global_rows or rows * world), self.loss_rows)
        # backward, last pipeline first.  RCCL chunks are held back while an LSTM backward launch is still to come: the cluster
        # form of the recurrence needs every CU and must not spin under an all-reduce kernel that holds some (vl_lstm_seq_status)
        self._pending_lstm = sum(1 for nd in self.nodes if nd.cls == "lstm")
        self._queued = []
        for nd in self.nodes:
            nd._dout_fresh = True
        for nd in reversed(self.nodes):
            nd.backward()
        self._flush()
        return self._finish_step(rows, lr, clip_norm, fetch)

    def _reduce(self, off, cnt):
        if self.dp is None or cnt == 0:
            return
        if self._pending_lstm > 0:
            self._queued.append((off, cnt))
        else:
            self.dp.reduce_async(self.g, off, cnt)

    def _flush(self):
        for off, cnt in self._queued:
            self.dp.reduce_async(self.g, off, cnt)
        self._queued = []

    def _head_done(self, nd):
        """The head of a pipeline has queued all its backward launches (its LSTM's included): its chunk may go."""
        if nd.cls == "lstm":
            self._pending_lstm -= 1
        if self.dp is None:
            return
        if nd.head_cnt:
            self._queued.append((nd.head_off, nd.head_cnt))
        if self._pending_lstm == 0:
            self._flush()

    def _tower_dp(self, nd):
        return _TowerReduce(self, nd.tower_off) if self.dp is not None else None

    def train_step_empty(self, lr, clip_norm=0.0, fetch=True):
        """This rank's shard of the global batch is empty: contribute zeros to the exchange, apply the same update as the others."""
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
        skip = ops.step_guard(self._skip, *[nd.lstm_ws for nd in self.nodes if nd.cls == "lstm"])   # LRCNEngine._finish_step
        if self.optimizer == "adam":
            ops.adam_apply(self.w, self.g, self.adam_m, self.adam_v, lr, self.step_count, clip_norm, self.ss, 1.0, skip=skip)
        else:
            ops.sgd_apply(self.w, self.g, lr, clip_norm, self.ss, 1.0, skip=skip)
        if not fetch:
            return None
        self._sync()
        self.check_status()
        st = self.stats.cpu().numpy()
        return {"loss": float(st[0]) / max(rows, 1), "accuracy": float(st[1]) / max(rows, 1),
                "grad_norm": math.sqrt(float(self.ss.item())), "rows": rows, "loss_sum": float(st[0]), "correct": float(st[1])}


class _TowerReduce:
    """Lets a tower's backward issue its gradient chunks on the shared flat buffer (its own `g` is a slice of it)."""

    def __init__(self, graph, base):
        self.graph, self.base, self.world = graph, base, graph.dp.world

    def reduce_async(self, flat, offset, count):
        self.graph._reduce(self.base + offset, count)

    def wait(self):
        self.graph.dp.wait()
