"""LRCNEngine: the two executor calls of the reference, on MI355X.

The reference assembles a TF graph once (models/model.py:18-155: dcnn representation ->
lstm | fc classifier -> logits; train.py:112-222: loss, clip, SGD) and then calls
  sess.run([summaries, loss, lr, global_step, optimizer], fdict)      run_task.py:44   -> train_step_*
  sess.run(model.logits, fdict)                                       run_task.py:95   -> forward_*
This class is that graph: a fixed plan of C-ABI kernel launches (vltf_amd.ops) over buffers
allocated once.  torch tensors are device memory only; no torch math runs on the data path.

Activations are NCHW; parameters keep the reference's layouts and TF variable names
(SURVEY.md section 5), stored in one flat fp32 buffer (and one flat gradient buffer) ordered
classifier -> fc -> conv5..conv1, i.e. the order backward produces gradients, so the
data-parallel all-reduce of the first (large: fc6 = 85 % of bytes) bucket overlaps the conv backward.
"""
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import contextlib
import os

import numpy as np
import torch

from . import ops
from ._ffi import VltfError

# name, kh, kw, cout, stride, groups, lrn, pool        (alexnet.py:60-211)
CONV_LAYERS = (
    ("conv1", 11, 11, 96, 4, 1, True, True),
    ("conv2", 5, 5, 256, 1, 2, True, True),
    ("conv3", 3, 3, 384, 1, 1, False, False),
    ("conv4", 3, 3, 384, 1, 2, False, False),
    ("conv5", 3, 3, 256, 1, 2, False, True),
)
FC_DIM = 4096
FORGET_BIAS = 1.0        # BasicLSTMCell default (lstm.py:17)
LRN = dict(radius=2, alpha=2e-5, beta=0.75, bias=1.0)     # alexnet.py:81-84


@dataclass
class NetConfig:
    """The subset of the `network:` / `train:` YAML keys (settings_.py:167-289) that shapes the graph."""
    image_shape: Tuple[int, int, int] = (227, 227, 3)
    num_classes: int = 101
    fpc: int = 16                               # frames per clip (from the .size file, dataset_.py:728)
    frame_encoding_layer: str = "fc6"           # alexnet.py:233-275: "fc6" | "fc7" | anything else -> fc8
    classifier: str = "lstm"                    # defs.classifier.{lstm, fc}; "none": a feature pipeline (classifier: None, model.py:110-112)
    lstm_hidden: int = 256
    lstm_layers: int = 1
    fusion: str = "avg"                         # lstm_params[2]: defs.fusion_method.{avg, last, reshape, state}
    frame_fusion: Optional[Tuple[str, str]] = None   # classifier fc: (early|late, avg|last) (model.py:103-106,149-151)
    dropout_keep_prob: float = 0.0              # <= 0 disables (lstm.py:52)
    optimizer: str = "sgd"                      # defs.optim.{sgd, adam}
    conv_math: str = "f32"                      # "f32" | "bf16x3" | "bf16x6" | "bf16" (ops.set_conv_math: opt-in bf16-MFMA conv products)

    def encode_dim(self):
        return FC_DIM if self.frame_encoding_layer in ("fc6", "fc7") else self.num_classes

    def out_dim(self):
        """Width of the pipeline's output rows: num_classes after a classifier, the encode width of a feature pipeline."""
        return self.encode_dim() if self.classifier == "none" else self.num_classes


def tf_same_out(n, s):
    return -(-n // s)


def param_specs(cfg: NetConfig):
    """[(tf_variable_name, shape)] in flat-buffer order (classifier, fc8..fc6, conv5..conv1)."""
    h, w, c = cfg.image_shape
    convs = []
    for name, kh, kw, co, s, g, _, pool in CONV_LAYERS:
        convs.append([("dcnn/%sW" % name, (kh, kw, c // g, co)), ("dcnn/%sb" % name, (co,))])
        h, w, c = tf_same_out(h, s), tf_same_out(w, s), co
        if pool:
            h, w = ops.pool_out(h), ops.pool_out(w)
    specs = []
    dim = cfg.encode_dim()
    if cfg.classifier == "lstm":
        if cfg.lstm_hidden != cfg.num_classes:
            # fusion `state`: logits = convert_dim_fc(final h) under its default name (model.py:137-141), else "output_fc" (lstm.py:90)
            head = "fc_convert" if cfg.fusion == "state" else "output_fc"
            specs += [(head + "_w", (cfg.lstm_hidden, cfg.num_classes)), (head + "_b", (cfg.num_classes,))]
        d = dim
        for l in range(cfg.lstm_layers):
            pre = "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
            specs += [(pre + "kernel", (d + cfg.lstm_hidden, 4 * cfg.lstm_hidden)), (pre + "bias", (4 * cfg.lstm_hidden,))]
            d = cfg.lstm_hidden
    elif cfg.classifier == "fc":
        if dim != cfg.num_classes:
            specs += [("fc_convert_w", (dim, cfg.num_classes)), ("fc_convert_b", (cfg.num_classes,))]
    elif cfg.classifier == "none":
        pass
    else:
        raise VltfError("Undefined classifier [%s]" % cfg.classifier)
    if cfg.frame_encoding_layer not in ("fc6", "fc7"):
        specs += [("dcnn/fc8W", (FC_DIM, cfg.num_classes)), ("dcnn/fc8b", (cfg.num_classes,))]
    if cfg.frame_encoding_layer != "fc6":
        specs += [("dcnn/fc7W", (FC_DIM, FC_DIM)), ("dcnn/fc7b", (FC_DIM,))]
    specs += [("dcnn/fc6W", (h * w * c, FC_DIM)), ("dcnn/fc6b", (FC_DIM,))]
    for pair in reversed(convs):
        specs += pair
    return specs


def init_params(cfg: NetConfig, seed=0, stddev=0.05, well_scaled=False):
    """Reference initialisers: W ~ truncated_normal(sigma=0.05) re-drawn beyond 2 sigma, b = 0.1
    (alexnet.py:40-46, tf_util.py:44-45); LSTM kernel glorot-uniform, bias 0 (TF defaults).
    well_scaled uses sigma = sqrt(2/fan_in) instead so activations stay O(1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in param_specs(cfg):
        if name.endswith("kernel"):
            lim = math.sqrt(6.0 / (shp[0] + shp[1]))
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = np.zeros(shp, np.float32)
        elif len(shp) > 1:
            sd = math.sqrt(2.0 / int(np.prod(shp[:-1]))) if well_scaled else stddev
            v = rng.standard_normal(shp)
            bad = np.abs(v) > 2.0
            while bad.any():
                v[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(v) > 2.0
            out[name] = (v * sd).astype(np.float32)
        else:
            out[name] = np.full(shp, 0.1, np.float32)
    return out


class LRCNEngine:
    FC6_CHUNKS = 4          # row blocks of the fc6 weight gradient = all-reduce chunks of the data-parallel exchange

    def __init__(self, cfg: NetConfig, max_clips: int, device="cuda:0", training=True, dp=None, flat=None):
        """flat: optional (w, g) slices of a larger flat parameter / gradient buffer to live in (vltf_amd.composed: several
        pipelines share one buffer so that the global-norm clip and the update run over all of them at once)."""
        if not torch.cuda.is_available():
            raise VltfError("LRCNEngine needs a HIP device; there is no CPU fallback")
        self.cfg, self.B, self.T = cfg, max_clips, cfg.fpc
        self.N = max_clips * cfg.fpc
        self.dev = torch.device(device)
        self.training = training
        self.dp = dp
        self.step_count = 0
        torch.cuda.set_device(self.dev)
        N, dev = self.N, self.dev

        def buf(*shape, dtype=torch.float32):
            return torch.empty(shape, dtype=dtype, device=dev)

        # ---- parameters: flat buffers + named views
        self.specs = param_specs(cfg)
        total = sum(int(np.prod(s)) for _, s in self.specs)
        if flat is not None:
            self.w, self.g = flat
            if self.w.numel() != total or (training and self.g.numel() != total):
                raise VltfError("flat parameter buffer has %d elements, the network needs %d" % (self.w.numel(), total))
        else:
            self.w = torch.zeros(total, device=dev)
            self.g = torch.zeros(total, device=dev) if training else None
        self.P, self.G, self.offsets = {}, {}, {}
        off = 0
        for name, shp in self.specs:
            n = int(np.prod(shp))
            self.P[name] = self.w[off:off + n].view(shp)
            if training:
                self.G[name] = self.g[off:off + n].view(shp)
            self.offsets[name] = (off, n)
            off += n
        # data-parallel exchange: chunks of the flat gradient in the order backward completes them (dp.py).  fc6W (85 % of the
        # bytes) goes in FC6_CHUNKS row blocks, each reduced as soon as the GEMM that produces it is queued.
        first_conv = self.offsets["dcnn/conv5W"][0]
        f6o, f6n = self.offsets["dcnn/fc6W"]
        rows6 = self.specs[[n for n, _ in self.specs].index("dcnn/fc6W")][1][0]
        nch = max(1, min(self.FC6_CHUNKS, rows6 // 128))
        edges = [(-(-rows6 * i // nch) + 127) // 128 * 128 if 0 < i < nch else (0 if i == 0 else rows6) for i in range(nch + 1)]
        self.fc6_row_blocks = [(edges[i], edges[i + 1]) for i in range(nch) if edges[i + 1] > edges[i]]
        self.grad_chunks = [(0, f6o)] if f6o > 0 else []
        for bi, (r0, r1) in enumerate(self.fc6_row_blocks):
            lo, hi = f6o + r0 * FC_DIM, f6o + r1 * FC_DIM
            if bi == len(self.fc6_row_blocks) - 1:
                hi = first_conv                                   # fc6b rides with the last block
            self.grad_chunks.append((lo, hi - lo))
        # conv5..conv3 (8.0 of the 9.3 MB of conv gradients) go out as soon as conv3's weight gradient is queued, while conv2 / conv1
        # backward (40 % of the conv backward) still runs; only conv2 + conv1 (1.4 MB) are left for the end of the step
        conv_lo = self.offsets["dcnn/conv2W"][0]
        self.grad_chunks.append((first_conv, conv_lo - first_conv))
        self.grad_chunks.append((conv_lo, total - conv_lo))
        assert sum(c for _, c in self.grad_chunks) == total and all(
            self.grad_chunks[i][0] + self.grad_chunks[i][1] == self.grad_chunks[i + 1][0] for i in range(len(self.grad_chunks) - 1))
        if cfg.optimizer == "adam" and training:
            self.adam_m, self.adam_v = torch.zeros(total, device=dev), torch.zeros(total, device=dev)

        # ---- conv stack plan.  Tensors a conv gathers from (its x, and the dy its dgrad reads) are stored
        # with a zero halo equal to the conv's SAME padding, so the im2col gather is test-free (vltf.h).
        def zbuf(n_, c_, h_, w_, halo, dtype=torch.float32):
            return torch.zeros((n_, c_, h_ + 2 * halo, w_ + 2 * halo), dtype=dtype, device=dev)

        h, w, c = cfg.image_shape
        convs = []
        for name, kh, kw, co, s, g, lrn, pool in CONV_LAYERS:
            conv = ops.Conv(c, h, w, co, kh, kw, s, g)
            convs.append(conv)
            h, w, c = conv.oh, conv.ow, co
            if pool:
                h, w = ops.pool_out(h), ops.pool_out(w)
        pads = [cv.same_pad() for cv in convs]
        h, w, c = cfg.image_shape
        self.x0_halo = pads[0]
        # conv1 is strided: its input is stored column-phase-split so that the taps of consecutive output columns are
        # consecutive addresses (vl_conv_set_x_phase_split); x0 is only ever read by conv1 (forward and wgrad)
        convs[0].set_halo(pads[0], 0, 0, 0)
        self.x0_phase = convs[0].set_x_phase_split(True) if pads[0] > 0 else 1
        self.x0 = torch.zeros(ops.phase_split_shape(N, c, h, w, self.x0_halo, self.x0_phase), device=dev)
        self.layers = []
        max_w = 0
        ws_bytes = 4
        for li, (name, kh, kw, co, s, g, lrn, pool) in enumerate(CONV_LAYERS):
            conv = convs[li]
            nxt = pads[li + 1] if li + 1 < len(convs) else 0
            L = dict(name=name, conv=conv, lrn=lrn, pool=pool, cin=c, h=h, w=w)
            L["x_halo"] = pads[li]
            L["y_halo"] = 0 if (lrn or pool) else nxt            # y feeds the next conv directly (conv3, conv4)
            L["dy_halo"] = pads[li] if li > 0 else 0             # dy is gathered by this layer's dgrad (not conv1)
            L["y"] = zbuf(N, co, conv.oh, conv.ow, L["y_halo"])
            if training:
                L["dy"] = zbuf(N, co, conv.oh, conv.ow, L["dy_halo"])
            out, out_halo = L["y"], L["y_halo"]
            h, w, c = conv.oh, conv.ow, co
            if pool:
                ph, pw = ops.pool_out(h), ops.pool_out(w)
                L["hwc"] = name == "conv5"       # pool5 writes the (h, w, c)-flat order fc6 reads (alexnet.py:228)
                L["p_halo"] = 0 if L["hwc"] else nxt
                L["p"] = buf(N, ph, pw, c) if L["hwc"] else zbuf(N, c, ph, pw, L["p_halo"])
                L["arg"] = torch.zeros(L["p"].shape, dtype=torch.uint8, device=dev)
                if training:
                    L["dp"] = buf(N, ph, pw, c) if L["hwc"] else zbuf(N, c, ph, pw, L["p_halo"])   # same layout as p / arg
                out, out_halo = L["p"], L["p_halo"]
                h, w = ph, pw
            L["out"], L["out_halo"] = out, out_halo
            max_w = max(max_w, kh * kw * (conv.cin // g) * co)
            self.layers.append(L)
        for li, L in enumerate(self.layers):
            prev = self.layers[li - 1] if li > 0 else None
            dx_halo = 0
            if prev is not None:                                 # dgrad writes into the previous layer's dp (pool) or dy
                dx_halo = prev["p_halo"] if prev["pool"] else prev["dy_halo"]
            L["conv"].set_halo(L["x_halo"], L["y_halo"], L["dy_halo"], dx_halo)
            if training:
                ws_bytes = max(ws_bytes, L["conv"].wgrad_ws_bytes(N))
        # conv_math "bf16" is a bf16 PATH for the stride-1 layers (csrc/conv_c8.hip): their operands live in memory as packed bf16
        # ("c8": 8 channels of a pixel per 16-byte chunk) next to the fp32 tensors the pool / LRN / bias-gradient kernels read.
        # xb: the layer's input; dyb: the gradient its dgrad / wgrad read; wb / wbt: the packed weights (rebuilt every step).
        self.c8 = cfg.conv_math == "bf16"
        self._xb_fed = False                          # feed_u8 wrote conv1's packed input directly
        if self.c8:
            def cbuf(c_, h_, w_, halo):
                return torch.zeros(ops.c8_shape(N, c_, h_, w_, halo), dtype=torch.bfloat16, device=dev)
            for L in self.layers[1:]:
                conv = L["conv"]
                L["xb"] = cbuf(conv.cin, conv.h, conv.w, L["x_halo"])
                L["wb"] = torch.zeros(conv.c8_w_bytes(False), dtype=torch.uint8, device=dev)
                if training:
                    L["dyb"] = cbuf(conv.cout, conv.oh, conv.ow, L["dy_halo"])
                    L["wbt"] = torch.zeros(conv.c8_w_bytes(True), dtype=torch.uint8, device=dev)
                    ws_bytes = max(ws_bytes, conv.c8_wgrad_ws_bytes(N))
            # the LRN layers' conv outputs are written packed only (no halo): lrn_pool_fwd_c8 / pool_lrn_bwd_c8 read them packed
            for L in self.layers:
                if L["lrn"] and L["pool"]:
                    L["yb"] = cbuf(L["conv"].cout, L["conv"].oh, L["conv"].ow, 0)
            # conv1 (strided, 3 channels) runs the same kernels as the equivalent stride-1 layer over its space-to-depth input
            L0 = self.layers[0]
            eq = L0["eq"] = L0["conv"].s2d_layer()
            L0["xb"] = cbuf(eq.cin, eq.h, eq.w, eq.x_halo)
            L0["ws2d"] = torch.zeros(eq.w_shape, device=dev)
            L0["wb"] = torch.zeros(eq.c8_w_bytes(False), dtype=torch.uint8, device=dev)
            if training:
                L0["dyb"] = cbuf(eq.cout, eq.oh, eq.ow, eq.dy_halo)
                L0["dws2d"] = torch.zeros(eq.w_shape, device=dev)
                ws_bytes = max(ws_bytes, eq.c8_wgrad_ws_bytes(N))
        self.flat_dim = h * w * c
        if self.c8 and N % 8 == 0 and os.environ.get("VLTF_FC6_KC8", "1") != "0":      # (0: A/B against the split-product GEMM)
            # fc6's three products on the wgrad kernel (ops.gemm_kc8: reduction-major packed operands), DESIGN 4.7
            F = self.flat_dim
            kb = lambda count: torch.zeros(count, dtype=torch.bfloat16, device=dev)
            self.k_act = kb(N * max(F, FC_DIM))            # the activation operand of the forward / input-gradient product
            self.k_w = kb(F * FC_DIM)                      # fc6W, reduction-major for the forward, then for the input gradient
            if training:
                self.k_p5, self.k_d6 = kb(N * F), kb(N * FC_DIM)
            H4_, D_ = 4 * cfg.lstm_hidden, cfg.encode_dim()
            for (gm, gn, gk) in ((N, FC_DIM, F), (N, F, FC_DIM), (F, FC_DIM, N), (N, H4_, D_), (D_, H4_, N), (N, D_, H4_)):
                ws_bytes = max(ws_bytes, ops.gemm_kc8_ws_bytes(gm, gn, gk))
        self.f6 = buf(N, FC_DIM)
        self.f7 = buf(N, FC_DIM) if cfg.frame_encoding_layer != "fc6" else None
        self.f8 = buf(N, cfg.num_classes) if cfg.frame_encoding_layer not in ("fc6", "fc7") else None
        self.feat = self.f8 if self.f8 is not None else (self.f7 if self.f7 is not None else self.f6)
        D, C, H, B, T = cfg.encode_dim(), cfg.num_classes, cfg.lstm_hidden, self.B, self.T
        if cfg.conv_math != "f32":                                # operand images of the split-product GEMMs (fc6 is the largest)
            for (gm, gn, gk) in ((N, FC_DIM, self.flat_dim), (N, self.flat_dim, FC_DIM), (self.flat_dim, FC_DIM, N), (N, 4 * H, D)):
                ws_bytes = max(ws_bytes, ops.gemm_split_ws_bytes(gm, gn, gk))
        self.ws = torch.empty(max(ws_bytes, 64 << 20) // 4, device=dev)       # wgrad slabs / split-K slabs / GEMM operand images
        self.small_ws = buf(64 * max(4 * H, FC_DIM, 1024, C))                     # colsum / bias / sumsq partials
        if training:
            self.wt = buf(max_w)
            self.df6 = buf(N, FC_DIM)
            self.df7 = buf(N, FC_DIM) if self.f7 is not None else None
            self.df8 = buf(N, C) if self.f8 is not None else None
            self.dfeat = self.df8 if self.df8 is not None else (self.df7 if self.df7 is not None else self.df6)
        # ---- classifier
        if cfg.classifier == "lstm":
            if cfg.fusion not in ops.FUSION_CODE and cfg.fusion not in ("state", "reshape"):
                raise VltfError("Undefined frame fusion type : %s" % cfg.fusion)          # tf_util.py:28-29
            # `state` = final h of the last layer = its output at t = T-1 (full-length sequences, lstm.py:136), no dropout;
            # `reshape` (tf_util.py:26-27) keeps every step: one logits row per frame, labels [clips * fpc, classes]
            self.lstm_fusion = "last" if cfg.fusion == "state" else cfg.fusion
            self.per_step = cfg.fusion == "reshape"
            self.head = "fc_convert" if cfg.fusion == "state" else "output_fc"
            self.lstm = []
            for l in range(cfg.lstm_layers):
                S = dict(gx=buf(N, 4 * H), act=buf(N, 4 * H), cseq=buf(N, H), hseq=buf(N, H), hprev=buf(N, H))
                if training:
                    S.update(dz=buf(N, 4 * H), dout=buf(N, H))
                self.lstm.append(S)
            self.gh = buf(B, 4 * H)
            self.lstm_ws = ops.lstm_seq_ws(B, T, H, dev) if H <= 1024 else None
            R = N if self.per_step else B                             # logits rows
            self.fused = buf(R, H)
            self.dropped = buf(R, H)
            self.drop_mask = buf(R, H, dtype=torch.uint8)
            self.logits = buf(R, C) if H != C else self.dropped
            if training:
                self.dh, self.dc = buf(B, H), buf(B, H)
                self.dfused, self.ddropped = buf(R, H), buf(R, H)
        else:
            ff = cfg.frame_fusion
            if cfg.classifier == "none" and ff and ff[0] == "late":
                raise VltfError("Specified late fusion with no classifier selected")       # model.py:36-37
            if ff and ff[0] in ("early", "late") and ff[1] not in ("avg", "last", "reshape"):
                raise VltfError("Undefined frame fusion type : %s" % ff[1])              # apply_temporal_fusion, tf_util.py:28-29
            if ff and ff[1] == "reshape":
                ff = None        # aggregate_clip_vectors with `reshape` (tf_util.py:126-133,26-27): [N, d] -> [B, T, d] -> [N, d], the identity
            self.early = bool(ff and ff[0] == "early" and T > 1)
            self.late = bool(ff and ff[0] == "late" and T > 1)
            self.ff_method = ff[1] if ff else None
            C = cfg.out_dim()                  # a feature pipeline (classifier none) ends at the encode width: no fc, D == C below
            rows = B if self.early else N
            self.fc_in = buf(B, D) if self.early else self.feat
            self.fc_out = buf(rows, C) if D != C else self.fc_in
            self.logits = buf(B, C) if self.late else self.fc_out
            if training:
                self.dfc_out = buf(rows, C) if self.late else None
                self.dfc_in = buf(B, D) if self.early else None
        self.rows_out = self.logits.shape[0]
        if training:
            self.dlogits = buf(*self.logits.shape)
        self.stats = torch.zeros(2, device=dev)
        self.loss_rows = torch.zeros(2 * self.rows_out, device=dev)     # per-row losses | hits (vl_softmax_xent workspace)
        self.ss = torch.zeros(1, device=dev)
        self._skip = torch.zeros(1, dtype=torch.int32, device=dev)      # ops.step_guard: the optimizer launch's skip word
        self.probe, self.probe_events = None, []
        self._resizers = {}
        self.mean_dev = torch.zeros(3, device=dev)

    # ---- live kernel timing (bench.py roofline): HIP events around labelled launches -------------
    def set_probe(self, labels):
        """labels: iterable of '<layer>.<fwd|dgrad|wgrad>' whose launches get bracketed by events
        recorded on the launch stream.  None disables."""
        self.probe = set(labels) if labels else None
        self.probe_events = []

    def _run(self, label, fn, *args, **kw):
        if self.probe is not None and label in self.probe:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*args, **kw)
            e1.record()
            self.probe_events.append((label, e0, e1))
        else:
            fn(*args, **kw)

    def probe_times_ms(self):
        torch.cuda.synchronize(self.dev)
        out = [(l, a.elapsed_time(b)) for l, a, b in self.probe_events]
        self.probe_events = []
        return out

    # ---- parameters ----------------------------------------------------------------------------
    def load_params(self, params: dict):
        """params: {tf variable name: numpy array in the reference layout}.  Unknown / missing names fail."""
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

    # ---- optimizer state (what tf.train.Saver() keeps besides the weights, feeder.py:201: Adam slots + beta powers) -------
    OPT_PREFIX = "__optimizer__/"

    def get_opt_state(self):
        """{reserved name: array} to store beside the weights; step_count drives Adam's bias correction and the dropout seed."""
        torch.cuda.synchronize(self.dev)
        st = {self.OPT_PREFIX + "step_count": np.array([self.step_count], np.int64)}
        if self.cfg.optimizer == "adam" and self.training:
            st[self.OPT_PREFIX + "adam_m"] = self.adam_m.detach().cpu().numpy().copy()
            st[self.OPT_PREFIX + "adam_v"] = self.adam_v.detach().cpu().numpy().copy()
        return st

    def load_opt_state(self, state, global_step=None):
        """Restores get_opt_state(); returns the names that were expected but absent (fresh optimizer for those)."""
        missing = []
        key = self.OPT_PREFIX + "step_count"
        if key in state:
            self.step_count = int(np.asarray(state[key]).ravel()[0])
        else:
            missing.append(key)
            if global_step is not None:
                self.step_count = int(global_step)
        if self.cfg.optimizer == "adam" and self.training:
            for name, t in (("adam_m", self.adam_m), ("adam_v", self.adam_v)):
                a = state.get(self.OPT_PREFIX + name)
                if a is None:
                    missing.append(self.OPT_PREFIX + name)
                    continue
                a = np.asarray(a, np.float32)
                if a.shape != (t.numel(),):
                    raise VltfError("optimizer state %s has shape %s, expected (%d,)" % (name, a.shape, t.numel()))
                t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
        return missing

    def check_status(self):
        """Raises when a cluster-form LSTM launch since the last check timed out (ops.lstm_seq_check: the flag is sticky over the
        launches of a step and reset here).  Synchronises; called wherever results are fetched to the host."""
        if getattr(self, "lstm_ws", None) is not None:
            ops.lstm_seq_check(self.lstm_ws)

    def logits_host(self, rows=None):
        torch.cuda.synchronize(self.dev)
        self.check_status()
        return self.logits[:rows if rows is not None else self._rows].detach().cpu().numpy().copy()

    # ---- input ---------------------------------------------------------------------------------
    def _check_frames(self, n):
        if n <= 0 or n % self.T or n > self.N:
            raise VltfError("got %d frames: need a positive multiple of fpc=%d, at most %d" % (n, self.T, self.N))
        return n // self.T

    def feed_u8(self, frames_u8, mean_bgr=None, crop_y=None, crop_x=None, mirror=None, resize=None):
        """frames uint8 [n, raw_h, raw_w, 3] on device (TFRecord image_raw bytes) -> x0 (dataset_.py:481-501).
        resize: [((h, w), (oh, ow)), ...] imresize steps applied first (imgproc raw_resize / resize: PIL bilinear on uint8)."""
        n = frames_u8.shape[0]
        b = self._check_frames(n)
        for src_hw, dst_hw in (resize or ()):
            key = (tuple(src_hw), tuple(dst_hw))
            if key not in self._resizers:
                self._resizers[key] = ops.Resize(src_hw[0], src_hw[1], dst_hw[0], dst_hw[1])
            frames_u8 = self._resizers[key](frames_u8)
        mean = None
        if mean_bgr is not None:
            self.mean_dev.copy_(torch.as_tensor(np.asarray(mean_bgr, np.float32)), non_blocking=True)
            mean = self.mean_dev
        if self.c8:     # bf16 path: the frames go straight into conv1's packed (space-to-depth) input; x0 is not written
            self.layers[0]["conv"].input_prep_u8_s2d(frames_u8, self.layers[0]["xb"][:n], crop_y, crop_x, mirror, mean)
            self._xb_fed = True
            return n, b
        ops.input_prep_u8(frames_u8, self.x0[:n], crop_y, crop_x, mirror, mean, halo=self.x0_halo, phase=self.x0_phase,
                          out_hw=self.cfg.image_shape[:2])
        return n, b

    def feed_f32_nhwc(self, frames):
        """The reference's placeholder format: float32 NHWC, already cropped / mean-subtracted (model.py:54)."""
        n = frames.shape[0]
        b = self._check_frames(n)
        ops.nhwc_to_nchw(frames, self.x0[:n], halo=self.x0_halo, phase=self.x0_phase)
        self._xb_fed = False
        return n, b

    def _fc6_kc8(self, n):
        """fc6 runs on the packed-operand product kernel: bf16 path, whole 8-frame blocks."""
        return self.c8 and hasattr(self, "k_w") and n % 8 == 0

    def _lstm_kc8(self, n, l):
        """The first LSTM layer's three whole-sequence products (input projection, kernel gradient, input gradient) run on the
        packed-operand kernel too (bf16 path): the same arithmetic as the split-product GEMM in mode 1 -- operands rounded to bf16, fp32
        accumulation -- without its operand images.  They reuse fc6's operand buffers (the launches are serial on one stream)."""
        D, H4 = self.cfg.encode_dim(), 4 * self.cfg.lstm_hidden
        return (l == 0 and self._fc6_kc8(n) and os.environ.get("VLTF_LSTM_KC8", "1") != "0" and D % 8 == 0 and H4 % 8 == 0
                and D * H4 <= self.k_w.numel() and n * max(D, H4) <= self.k_act.numel() and (not self.training or n * H4 <= self.k_d6.numel()))

    # ---- forward -------------------------------------------------------------------------------
    def _forward(self, n, b, train):
        P, cfg = self.P, self.cfg
        ops.set_conv_math(cfg.conv_math)             # process-wide switch: set per call so that engines of both kinds can coexist
        x = self.x0[:n]
        for li, L in enumerate(self.layers):
            name = L["name"]
            nxt = self.layers[li + 1] if li + 1 < len(self.layers) else None
            if self.c8 and li == 0:
                conv = L["conv"]
                if not self._xb_fed:                      # fed as fp32 frames (feed_f32_nhwc): pack x0
                    conv.s2d_c8_from_x0(x, L["xb"][:n])
                conv.s2d_weights(P["dcnn/%sW" % name], L["ws2d"])
                L["eq"].c8_pack_w(L["ws2d"], L["wb"], False)
                self._run(name + ".fwd", L["eq"].c8_fwd, L["xb"][:n], L["wb"], P["dcnn/%sb" % name], yb=L["yb"][:n], relu=True)
            elif self.c8:
                # bf16 path: packed operands; a conv that feeds the next conv directly also writes that conv's packed input
                L["conv"].c8_pack_w(P["dcnn/%sW" % name], L["wb"], False)
                yb = nxt["xb"][:n] if (nxt is not None and not L["pool"]) else (L["yb"][:n] if "yb" in L else None)
                y = L["y"][:n] if yb is None else None       # fp32 output only where the plain max-pool reads it (conv5)
                self._run(name + ".fwd", L["conv"].c8_fwd, L["xb"][:n], L["wb"], P["dcnn/%sb" % name], y=y, yb=yb, relu=True)
            else:
                self._run(name + ".fwd", L["conv"].fwd, x, P["dcnn/%sW" % name], P["dcnn/%sb" % name], L["y"][:n], relu=True)
            x = L["y"][:n]
            if L["lrn"] and L["pool"] and self.c8:
                # bf16 path: the pooled output is only ever the next conv's packed operand -- written packed, fp32 p stays unused
                ops.lrn_pool_fwd_c8(L["yb"][:n], nxt["xb"][:n], L["arg"][:n], p_halo=L["p_halo"], channels=L["conv"].cout, **LRN)
                x = None
            elif L["lrn"] and L["pool"]:
                # LRN + pool in one pass: the LRN output is only ever the pool's input and is never stored
                ops.lrn_pool_fwd(x, L["p"][:n], L["arg"][:n], p_halo=L["p_halo"], **LRN)
                x = L["p"][:n]
            elif L["pool"]:
                ops.maxpool_fwd(x, L["p"][:n], L["arg"][:n], hwc=L["hwc"], y_halo=L["p_halo"])
                x = L["p"][:n]
            if self.c8 and L["pool"] and not L["lrn"] and nxt is not None:
                ops.pack_c8(x, nxt["xb"][:n], L["p_halo"], nxt["x_halo"])
        if self._fc6_kc8(n):
            F = self.flat_dim
            a = self.k_act[:n * F].view(ops.kc8_shape(F, n))
            w = self.k_w.view(ops.kc8_shape(F, FC_DIM))
            ops.pack_kc8(x, a, F, n, 1, F)                                    # (position f, channel frame) = pool5[frame][f]
            ops.pack_kc8(P["dcnn/fc6W"], w, F, FC_DIM, FC_DIM, 1)
            ops.gemm_kc8(a, w, self.f6, n, FC_DIM, F, bias=P["dcnn/fc6b"], relu=True, ws=self.ws)
        else:
            ops.gemm(x, P["dcnn/fc6W"], self.f6, n, FC_DIM, self.flat_dim, bias=P["dcnn/fc6b"], relu=True, ws=self.ws)
        if self.f7 is not None:
            ops.gemm(self.f6, P["dcnn/fc7W"], self.f7, n, FC_DIM, FC_DIM, bias=P["dcnn/fc7b"], relu=True, ws=self.ws)
        if self.f8 is not None:
            ops.gemm(self.f7, P["dcnn/fc8W"], self.f8, n, cfg.num_classes, FC_DIM, bias=P["dcnn/fc8b"], ws=self.ws)
        D, C, H, T = cfg.encode_dim(), cfg.num_classes, cfg.lstm_hidden, self.T
        if cfg.classifier == "lstm":
            xin, d = self.feat, D
            for l, S in enumerate(self.lstm):
                pre = "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
                K = P[pre + "kernel"]
                # hoisted input projection for all (clip, t) rows, then the serial recurrence
                if self._lstm_kc8(n, l):
                    a = self.k_act[:n * d].view(ops.kc8_shape(d, n))
                    w = self.k_w[:d * 4 * H].view(ops.kc8_shape(d, 4 * H))
                    ops.pack_kc8(xin, a, d, n, 1, d)                                  # (position j, channel frame) = x[frame][j]
                    ops.pack_kc8(K, w, d, 4 * H, 4 * H, 1)                            # (position j, channel column) = K[j][column]
                    ops.gemm_kc8(a, w, S["gx"], n, 4 * H, d, bias=P[pre + "bias"], ws=self.ws)
                else:
                    ops.gemm(xin, K, S["gx"], n, 4 * H, d, bias=P[pre + "bias"], ws=self.ws)
                if H <= 1024:
                    ops.lstm_seq_fwd(S["gx"], K[d:], S["act"], S["cseq"], S["hseq"], S["hprev"], b, T, H, FORGET_BIAS, ws=self.lstm_ws)
                else:
                    for t in range(T):
                        if t > 0:
                            ops.gemm(S["hseq"][t - 1:], K[d:], self.gh, b, 4 * H, H, lda=T * H)
                        ops.lstm_step_fwd(S["gx"], self.gh if t > 0 else None, S["act"], S["cseq"], S["hseq"], S["hprev"], b, T,
                                          t, H, FORGET_BIAS)
                xin, d = S["hseq"], H
            r = n if self.per_step else b
            if self.per_step:
                v = xin                                                  # every step's output of the last layer
            else:
                ops.temporal_fusion_fwd(xin, self.fused, b, T, H, self.lstm_fusion)
                v = self.fused
            self._dropout = train and cfg.dropout_keep_prob > 0 and cfg.fusion != "state"
            if self._dropout:
                ops.dropout_fwd(v[:r], self.dropped[:r], self.drop_mask[:r], cfg.dropout_keep_prob,
                                (self.step_count << 20) ^ 0x5DEECE66D)
                v = self.dropped
            self._v = v
            if H != C:
                ops.gemm(v, P[self.head + "_w"], self.logits, r, C, H, bias=P[self.head + "_b"])
            elif v is not self.logits:
                self.logits[:r].copy_(v[:r])
            self._rows = r
        else:
            C = cfg.out_dim()
            v, rows = self.feat, n
            if self.early:
                ops.temporal_fusion_fwd(self.feat, self.fc_in, b, T, D, self.ff_method)
                v, rows = self.fc_in, b
            if D != C:
                ops.gemm(v, P["fc_convert_w"], self.fc_out, rows, C, D, bias=P["fc_convert_b"])
            if self.late:
                ops.temporal_fusion_fwd(self.fc_out, self.logits, b, T, C, self.ff_method)
                rows = b
            self._rows = rows
        return self._rows

    def forward_u8(self, frames_u8, mean_bgr=None, crop_y=None, crop_x=None, mirror=None, resize=None):
        """sess.run(model.logits, fdict) (run_task.py:95).  Returns a device view [rows, classes]."""
        n, b = self.feed_u8(frames_u8, mean_bgr, crop_y, crop_x, mirror, resize)
        rows = self._forward(n, b, train=False)
        return self.logits[:rows]

    def forward_f32(self, frames_nhwc):
        n, b = self.feed_f32_nhwc(frames_nhwc)
        rows = self._forward(n, b, train=False)
        return self.logits[:rows]

    # ---- backward ------------------------------------------------------------------------------
    def _backward(self, n, b):
        P, G, cfg = self.P, self.G, self.cfg
        ops.set_conv_math(cfg.conv_math)             # process-wide switch (see _forward): another engine may have run in between
        self._frames_now = n
        D, C, H, T = cfg.encode_dim(), cfg.num_classes, cfg.lstm_hidden, self.T
        sw = self.small_ws
        side = self._side_stream()
        if side is not None:
            # the flipped / transposed weights every dgrad reads, all layers now, on the second stream (idle until fc6): a transpose
            # in front of its dgrad sits on the backward's critical chain while the weight gradient on the other stream takes the CUs
            # (round 3, same box: 8 clips 5.71 -> 5.42 ms, 16 clips 9.90 -> 9.81, 64 clips on two streams 35.25 -> 35.05; which of
            # dgrad / wgrad the host issues first made no difference)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for L in self.layers[1:]:
                    L["conv"].wt_transpose(P["dcnn/%sW" % L["name"]], L["wt"])
                self._wt_ready.record(side)          # the first dgrad waits for this (below), nothing else on the launch stream does
        main = torch.cuda.current_stream(self.dev)

        def param_grads(launch):
            """The chain of INPUT gradients is the backward's critical path and stays on the launch stream; a parameter gradient only has
            to be done by the end of the pass: with a second stream it runs there (from where the launch stream stands now, on the
            second stream's own scratch), beside the chain.  Head / LSTM / fc6 parameter gradients there instead of on the chain
            (round 3, same box): 8 clips 5.43 -> 5.32 ms, 64 clips 35.0 -> 34.75."""
            if side is None:
                launch(self.ws, sw)
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    launch(self.ws_side, self.small_ws_side)

        if cfg.classifier == "lstm":
            d, r = self.dlogits, self._rows
            if H != C:
                def head_grads(ws, sws, r=r):
                    ops.gemm(self._v, self.dlogits, G[self.head + "_w"], H, C, r, transa=True)
                    ops.colsum(self.dlogits, G[self.head + "_b"], sws, r, C)
                param_grads(head_grads)
                ops.gemm(self.dlogits, P[self.head + "_w"], self.ddropped, r, H, C, transb=True)
                d = self.ddropped
            if self._dropout:
                ops.dropout_bwd(d[:r], self.drop_mask[:r], self.dfused[:r], cfg.dropout_keep_prob)
                d = self.dfused
            top = self.lstm[-1]
            if self.per_step:
                top["dout"][:r].copy_(d[:r])
            else:
                ops.temporal_fusion_bwd(d, top["dout"], b, T, H, self.lstm_fusion)
            for l in reversed(range(cfg.lstm_layers)):
                S = self.lstm[l]
                pre = "rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % l
                K = P[pre + "kernel"]
                din = D if l == 0 else H
                xin = self.feat if l == 0 else self.lstm[l - 1]["hseq"]
                if H <= 1024:
                    ops.lstm_seq_bwd(S["dout"], K[din:], S["act"], S["cseq"], S["dz"], b, T, H, ws=self.lstm_ws)
                else:
                    ops.fill(self.dc, 0.0)
                    for t in reversed(range(T)):
                        ops.lstm_step_bwd(S["dout"], self.dh if t < T - 1 else None, S["act"], S["cseq"], self.dc, S["dz"], b, T, t, H)
                        if t > 0:
                            ops.gemm(S["dz"][t:], K[din:], self.dh, b, H, 4 * H, transb=True, lda=T * 4 * H)
                lk = self._lstm_kc8(n, l)

                def lstm_grads(ws, sws, S=S, pre=pre, din=din, xin=xin, lk=lk):
                    if lk:                                                                # (bf16 path: one stream)
                        a = self.k_act[:n * din].view(ops.kc8_shape(n, din))
                        zb = self.k_d6[:n * 4 * H].view(ops.kc8_shape(n, 4 * H))
                        ops.pack_kc8(xin, a, n, din, din, 1)                              # (position frame, channel j)
                        ops.pack_kc8(S["dz"], zb, n, 4 * H, 4 * H, 1)                     # (position frame, channel column)
                        ops.gemm_kc8(a, zb, G[pre + "kernel"], din, 4 * H, n, ws=ws)
                    else:
                        ops.gemm(xin, S["dz"], G[pre + "kernel"], din, 4 * H, n, transa=True, ws=ws)
                    ops.gemm(S["hprev"], S["dz"], G[pre + "kernel"][din:], H, 4 * H, n, transa=True, ws=ws)
                    ops.colsum(S["dz"], G[pre + "bias"], sws, n, 4 * H)
                param_grads(lstm_grads)
                if l == 0 and lk:
                    a = self.k_act[:n * 4 * H].view(ops.kc8_shape(4 * H, n))
                    w = self.k_w[:D * 4 * H].view(ops.kc8_shape(4 * H, D))
                    ops.pack_kc8(S["dz"], a, 4 * H, n, 1, 4 * H)                      # (position column, channel frame) = dz[frame][column]
                    ops.pack_kc8(K, w, 4 * H, D, 1, 4 * H)                            # (position column, channel j) = K[j][column]
                    ops.gemm_kc8(a, w, self.dfeat, n, D, 4 * H, ws=self.ws)
                    if self.f8 is None:
                        ops.relu_grad(self.dfeat, self.feat, n * D)                   # ReluGrad of fc6 / fc7
                elif l == 0:
                    relu_mask = self.feat if self.f8 is None else None       # ReluGrad of fc6 / fc7 fused here
                    ops.gemm(S["dz"], K, self.dfeat, n, D, 4 * H, transb=True, ldb=4 * H, relu_mask=relu_mask, ws=self.ws)
                else:
                    ops.gemm(S["dz"], K, self.lstm[l - 1]["dout"], n, H, 4 * H, transb=True, ldb=4 * H, ws=self.ws)
        else:
            Co = cfg.out_dim()                # width of the pipeline output (= D for a feature pipeline)
            d, rows = self.dlogits, self._rows
            if self.late:
                ops.temporal_fusion_bwd(self.dlogits, self.dfc_out, b, T, Co, self.ff_method)
                d, rows = self.dfc_out, b * T
            relu_mask = self.feat if (self.f8 is None and not self.early) else None
            target = self.dfc_in if self.early else self.dfeat
            if D != Co:
                ops.gemm(self.fc_in, d, G["fc_convert_w"], D, Co, rows, transa=True)
                ops.colsum(d, G["fc_convert_b"], sw, rows, Co)
                ops.gemm(d, P["fc_convert_w"], target, rows, D, Co, transb=True, relu_mask=relu_mask)
            else:
                target[:rows].copy_(d[:rows])
                if relu_mask is not None:        # no fc in between (a feature pipeline): the encode layer's ReluGrad applies here
                    ops.relu_grad(target, relu_mask, rows * D)
            if self.early:
                # ReluGrad of the encode layer applies per frame after un-fusing
                ops.temporal_fusion_bwd(self.dfc_in, self.dfeat, b, T, D, self.ff_method)
                if self.f8 is None:
                    ops.relu_grad(self.dfeat, self.feat, n * D)
        # ---- fc8 / fc7 / fc6 (dfeat already carries the ReluGrad of the encode layer)
        d = self.dfeat
        if self.f8 is not None:
            ops.gemm(self.f7, d, G["dcnn/fc8W"], FC_DIM, C, n, transa=True, ws=self.ws)
            ops.colsum(d, G["dcnn/fc8b"], sw, n, C)
            ops.gemm(d, P["dcnn/fc8W"], self.df7, n, FC_DIM, C, transb=True, relu_mask=self.f7, ws=self.ws)
            d = self.df7
        if self.f7 is not None:
            ops.gemm(self.f6, d, G["dcnn/fc7W"], FC_DIM, FC_DIM, n, transa=True, ws=self.ws)
            ops.colsum(d, G["dcnn/fc7b"], sw, n, FC_DIM)
            ops.gemm(d, P["dcnn/fc7W"], self.df6, n, FC_DIM, FC_DIM, transb=True, relu_mask=self.f6, ws=self.ws)
            d = self.df6
        L5 = self.layers[-1]
        kc8 = self._fc6_kc8(n)
        if kc8:
            # bf16 path (one stream): weight gradient in one pass on the packed-operand kernel (the exchange chunks follow it)
            ops.colsum(d, G["dcnn/fc6b"], sw, n, FC_DIM)
            F = self.flat_dim
            a = self.k_p5[:n * F].view(ops.kc8_shape(n, F))
            b_ = self.k_d6[:n * FC_DIM].view(ops.kc8_shape(n, FC_DIM))
            ops.pack_kc8(L5["p"], a, n, F, F, 1)                              # (position frame, channel f)
            ops.pack_kc8(d, b_, n, FC_DIM, FC_DIM, 1)
            if self.dp is None:
                ops.gemm_kc8(a, b_, G["dcnn/fc6W"], F, FC_DIM, n, ws=self.ws)
            else:
                # the exchange starts here, as on the fp32 path: a row block of fc6W = a range of the packed operand's 8-row
                # blocks (block edges are multiples of 128), each block's all-reduce issued right behind its product
                chunks = iter(self.grad_chunks)
                if self.offsets["dcnn/fc6W"][0] > 0:
                    self.dp.reduce_async(self.g, *next(chunks))
                for r0, r1 in self.fc6_row_blocks:
                    ops.gemm_kc8(a[r0 // 8:r1 // 8], b_, G["dcnn/fc6W"][r0:r1], r1 - r0, FC_DIM, n, ws=self.ws)
                    self.dp.reduce_async(self.g, *next(chunks))
            a = self.k_act[:n * FC_DIM].view(ops.kc8_shape(FC_DIM, n))
            w = self.k_w.view(ops.kc8_shape(FC_DIM, F))
            ops.pack_kc8(d, a, FC_DIM, n, 1, FC_DIM)                          # (position j, channel frame) = dfc6[frame][j]
            ops.pack_kc8(P["dcnn/fc6W"], w, FC_DIM, F, 1, FC_DIM)             # (position j, channel f) = W[f][j]
            ops.gemm_kc8(a, w, L5["dp"], n, F, FC_DIM, ws=self.ws)
        else:
            def fc6_grads(ws, sws, d=d):
                ops.colsum(d, G["dcnn/fc6b"], sws, n, FC_DIM)
                if self.dp is None:
                    ops.gemm(L5["p"], d, G["dcnn/fc6W"], self.flat_dim, FC_DIM, n, transa=True, ws=ws)
                    return
                # the exchange starts here: everything produced so far, then fc6W block by block -- block i is on the wire
                # (RCCL's stream, which waits for the stream these launches are on) while block i+1 is computed, and the whole 85 % of
                # the bytes before the conv backward is far along
                chunks = iter(self.grad_chunks)
                if self.offsets["dcnn/fc6W"][0] > 0:
                    self.dp.reduce_async(self.g, *next(chunks))
                flat_p = L5["p"].view(self.N, self.flat_dim)
                for r0, r1 in self.fc6_row_blocks:
                    ops.gemm(flat_p[:, r0:], d, G["dcnn/fc6W"][r0:r1], r1 - r0, FC_DIM, n, transa=True, lda=self.flat_dim)
                    self.dp.reduce_async(self.g, *next(chunks))
            param_grads(fc6_grads)
            ops.gemm(d, P["dcnn/fc6W"], L5["dp"], n, self.flat_dim, FC_DIM, transb=True, ws=self.ws)     # the chain: into pool5's gradient
        # ---- conv stack, last to first
        if side is not None:
            main.wait_event(self._wt_ready)

        def on_side(launch):
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                launch()

        for li in reversed(range(len(self.layers))):
            L = self.layers[li]
            name, conv = L["name"], L["conv"]
            x_in = self.layers[li - 1]["out"][:n] if li > 0 else self.x0[:n]
            dy = L["dy"][:n]
            if L["pool"] and L["lrn"] and self.c8:
                # bf16 path: the gradient is only ever read packed (wgrad, dgrad, bias gradient) -- written packed, fp32 dy stays unused
                ops.pool_lrn_bwd_c8(L["yb"][:n], L["dp"][:n], L["arg"][:n], L["dyb"][:n], p_halo=L["p_halo"],
                                    dxb_halo=(L["eq"].dy_halo if li == 0 else L["dy_halo"]), relu_fused=True, **LRN)
            elif L["pool"] and L["lrn"]:
                # pool -> LRN -> ReLU backward in one pass: d(lrn out) is never written
                ops.pool_lrn_bwd(L["y"][:n], L["dp"][:n], L["arg"][:n], dy, p_halo=L["p_halo"], dx_halo=L["dy_halo"],
                                 relu_fused=True, **LRN)
            elif L["pool"]:
                self._pool_bwd(L, n, dy, L["y"][:n], L["dy_halo"])
            # else: dy was written (ReluGrad fused) by the next layer's dgrad epilogue
            if self.c8 and li == 0:
                eq = L["eq"]
                self._run(name + ".wgrad", eq.c8_wgrad, L["xb"][:n], L["dyb"][:n], L["dws2d"], self.ws)
                conv.s2d_weights(L["dws2d"], G["dcnn/%sW" % name], grad=True)
                ops.bias_grad_c8(L["dyb"][:n], G["dcnn/%sb" % name], sw, conv.cout, eq.dy_halo)
                continue
            if self.c8:
                if L["pool"] and not L["lrn"]:
                    ops.pack_c8(dy, L["dyb"][:n], L["dy_halo"], L["dy_halo"])
                self._run(name + ".wgrad", conv.c8_wgrad, L["xb"][:n], L["dyb"][:n], G["dcnn/%sW" % name], self.ws)
                ops.bias_grad_c8(L["dyb"][:n], G["dcnn/%sb" % name], sw, conv.cout, L["dy_halo"])
                if self.dp is not None and name == "conv3":
                    self.dp.reduce_async(self.g, *self.grad_chunks[-2])
                prev = self.layers[li - 1]
                conv.c8_pack_w(P["dcnn/%sW" % name], L["wbt"], True)
                if prev["pool"]:
                    self._run(name + ".dgrad", conv.c8_dgrad, L["dyb"][:n], L["wbt"], dx=prev["dp"][:n])
                else:      # straight into the previous conv's packed gradient; its ReluGrad reads this layer's packed input
                    self._run(name + ".dgrad", conv.c8_dgrad, L["dyb"][:n], L["wbt"], dxb=prev["dyb"][:n], relu_mask_c8=L["xb"][:n])
                continue
            # the side stream's OWN scratch: nothing the main stream launches meanwhile can touch it, whatever takes a workspace there later
            # conv1's weight gradient is the last launch of the pass and nothing on the chain follows it: on the launch stream it runs
            # beside what the second stream still holds instead of queueing behind it (64 clips 34.76 -> 34.71 ms, 8 clips 5.33 -> 5.29)
            last_on_main = side is not None and li == 0
            wws, wsw = (self.ws_side, self.small_ws_side) if (side is not None and not last_on_main) else (self.ws, sw)

            def wgrad(name=name, conv=conv, x_in=x_in, dy=dy, wws=wws, wsw=wsw):
                if conv.fuses_bias():      # bias gradient comes out of the same pass over dy
                    self._run(name + ".wgrad", conv.wgrad, x_in, dy, G["dcnn/%sW" % name], wws, db=G["dcnn/%sb" % name])
                else:
                    self._run(name + ".wgrad", conv.wgrad, x_in, dy, G["dcnn/%sW" % name], wws)
                    ops.bias_grad_nchw(dy, G["dcnn/%sb" % name], wsw)
                if self.dp is not None and name == "conv3":
                    # issued from the stream the weight gradients ran on: RCCL's stream waits for that stream only
                    self.dp.reduce_async(self.g, *self.grad_chunks[-2])

            if side is None or last_on_main:
                wgrad()                               # (conv1's: the chain ends here -- beside what the second stream still holds)
            else:
                on_side(wgrad)                        # beside this layer's dgrad / the next pool backward
            if li > 0:
                prev = self.layers[li - 1]
                wt = self.wt
                if side is not None:
                    wt = L["wt"]                      # transposed at the start of the backward pass
                else:
                    conv.wt_transpose(P["dcnn/%sW" % name], wt)
                if prev["pool"]:
                    self._run(name + ".dgrad", conv.dgrad, dy, wt, prev["dp"][:n])     # into the pool output gradient
                else:
                    self._run(name + ".dgrad", conv.dgrad, dy, wt, prev["dy"][:n], relu_mask=prev["y"][:n])
        if side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(side)
        if self.dp is not None:
            self.dp.reduce_async(self.g, *self.grad_chunks[-1])

    def _side_stream(self):
        """Second HIP stream of the backward pass, or None (VLTF_WGRAD_STREAM=0; the bf16 path).  Independent launches -- the dgrad
        weight transposes at the start of the pass, fc6's input gradient beside its weight-gradient blocks, a layer's weight gradient
        beside its input gradient and the next pool / LRN backward -- fill the CUs the tail of a launch leaves idle: same box, one
        stream -> two: 8 clips 5.71 -> 5.42 ms, 16 clips 10.1 -> 9.81, 32 clips 18.5 -> 18.2, 64 clips 35.64 -> 35.05 (round 3, once the
        transposes had left the critical chain; round 2 kept the full batch on one stream for a gain of 0.4 ms).  A bracket or a
        rocprof average of a launch that has a neighbour times both: bench.py takes its per-launch table from extra steps with
        VLTF_WGRAD_STREAM=0 and the dominant symbol, live, from its forward launches, which always run alone.
        Other schedules measured at 64 clips and dropped (35.75 ms on one stream then): conv5..conv3's weight gradients held back
        until conv2's pool / LRN backward, to hide that HBM-bound kernel under MFMA-bound ones -- 35.8 ms, next to a 135 KB-of-LDS
        wgrad workgroup a CU holds ONE pool / LRN workgroup, which then crawls (1.33 ms instead of 0.65) while the chain of input
        gradients waits for it; conv3's two launches each alone with the others paired -- 35.85, the joins cost what the pairs
        win; only conv2's weight gradient on the second stream, behind conv2's input gradient, so that it runs beside conv1's pool /
        LRN backward (off the critical chain) -- 36.03: an HBM-bound kernel beside an MFMA-bound one costs the latter more than it
        hides; which of dgrad / wgrad the host issues first: no difference.
        fp32 and the split-bf16 arithmetics (bf16x3 22.3 -> 21.5 ms).  Round 3 kept the latter on one stream because conv2's
        `pool_lrn_bwd` came out different from run to run beside conv3's split-product weight gradient.  Round 4 found the cause
        (DESIGN 6): not a race -- hipcc's SLP vectoriser had formed `v_pk_add_f32 ... op_sel:[0,1]` in that kernel, a packed-fp32 form
        that on MI355X returns src0.lo + 0 in lanes 48..63 while a kernel mixing v_cvt_pk / v_pk_add and bf16 MFMAs shares the CU
        (tools/ubench/pk_opsel_raw.hip reproduces it in 100 lines).  The library is built without SLP vectorisation and
        tests/test_isa_lint.py refuses the form in the built code objects.  The packed-bf16 PATH stays on one stream because it is
        slower on two (8.3 vs 8.2 ms)."""
        if os.environ.get("VLTF_WGRAD_STREAM", "") == "0" or self.cfg.conv_math == "bf16":
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
            # Invariant of the two-stream backward: a launch on the side stream reads tensors the main stream has finished (it waits
            # for the main stream at the start of every layer), writes only its own gradient tensors, and takes its scratch from
            # THESE buffers, never from self.ws / self.small_ws -- so no main-stream launch between the fork and the join can race it.
            self.ws_side = torch.empty_like(self.ws)              # wgrad slabs / split-k slabs of what runs on the side stream
            self.small_ws_side = torch.empty_like(self.small_ws)
            for L in self.layers[1:]:
                L["wt"] = torch.empty(L["conv"].w_shape, device=self.dev).view(-1)
            self._wt_ready = torch.cuda.Event()
        return self._side

    def _pool_bwd(self, L, n, dx, relu_mask, dx_halo):
        ops.maxpool_bwd(L["dp"][:n], L["arg"][:n], dx, relu_mask=relu_mask, hwc=L["hwc"], dy_halo=L["p_halo"], dx_halo=dx_halo)

    # ---- train step ----------------------------------------------------------------------------
    def _train(self, n, b, onehot, lr, clip_norm, fetch, global_rows=None):
        if not self.training:
            raise VltfError("engine was built with training=False")
        if self.cfg.classifier == "none":
            raise VltfError("a feature pipeline (classifier none) has no loss of its own: it trains inside a GraphEngine")
        if onehot.dtype != torch.int32 or tuple(onehot.shape) != (self._rows_for(b, n), self.cfg.num_classes):
            raise VltfError("labels must be int32 one-hot of shape (%d, %d)" % (self._rows_for(b, n), self.cfg.num_classes))
        rows = self._forward(n, b, train=True)
        world = self.dp.world if self.dp is not None else 1
        ops.fill(self.stats, 0.0)
        # mean over the GLOBAL batch (train.py:123): each rank scales its rows by 1/global_rows and the all-reduce sums.
        # global_rows defaults to rows*world (equal shards); a workflow with ragged shards passes the true count.
        ops.softmax_xent(self.logits[:rows], onehot, self.dlogits, self.stats, 1.0 / (global_rows or rows * world), self.loss_rows)
        self._backward(n, b)
        return self._finish_step(rows, lr, clip_norm, fetch)

    def train_step_empty(self, lr, clip_norm=0.0, fetch=True):
        """This rank's shard of the global batch is empty (fewer videos than ranks in a short last batch): contribute zero
        gradients to the exchange and apply the same update as every other rank."""
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
        # a step whose LSTM cluster launch timed out must not reach the weights -- also with fetch=False, where the host reads the
        # status only later: the optimizer launch drops the update on the device (ops.step_guard), check_status raises at the next fetch
        skip = ops.step_guard(self._skip, getattr(self, "lstm_ws", None))
        if self.cfg.optimizer == "adam":
            ops.adam_apply(self.w, self.g, self.adam_m, self.adam_v, lr, self.step_count, clip_norm, self.ss, 1.0, skip=skip)
        else:
            ops.sgd_apply(self.w, self.g, lr, clip_norm, self.ss, 1.0, skip=skip)
        if not fetch:
            return None
        torch.cuda.synchronize(self.dev)
        self.check_status()
        st = self.stats.cpu().numpy()
        return {"loss": float(st[0]) / max(rows, 1), "accuracy": float(st[1]) / max(rows, 1), "grad_norm": math.sqrt(float(self.ss.item())),
                "rows": rows, "loss_sum": float(st[0]), "correct": float(st[1])}

    def _rows_for(self, b, n):
        if self.cfg.classifier == "lstm":
            return n if self.per_step else b
        return b if (self.early or self.late) else n

    def train_step_u8(self, frames_u8, onehot, lr, clip_norm=0.0, mean_bgr=None, crop_y=None, crop_x=None, mirror=None,
                      fetch=True, global_rows=None, resize=None):
        """sess.run([summaries, loss, lr, global_step, optimizer], fdict) (run_task.py:44)."""
        n, b = self.feed_u8(frames_u8, mean_bgr, crop_y, crop_x, mirror, resize)
        return self._train(n, b, onehot, lr, clip_norm, fetch, global_rows)

    def train_step_f32(self, frames_nhwc, onehot, lr, clip_norm=0.0, fetch=True):
        n, b = self.feed_f32_nhwc(frames_nhwc)
        return self._train(n, b, onehot, lr, clip_norm, fetch)
