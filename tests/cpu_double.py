"""TEST INFRASTRUCTURE: a torch-CPU stand-in for vltf_amd.ops and for the AlexNet tower, so that the HOST LOGIC of
vltf_amd.graph.GraphEngine (which stage runs when, which buffer a gradient lands in, how pipelines are wired, chunk order of the
data-parallel exchange) can be exercised in the build container, which has no GPU.  Nothing here is product code and the product
never imports it: `install(monkeypatch)` swaps graph.ops / graph.LRCNEngine for these stand-ins for the duration of one test.
The device kernels themselves are tested on the GPU (tests/test_graph_gpu.py, -m gpu) against the same oracle."""
import math

import numpy as np
import torch

from oracle import torch_cpu as TC


def _mat(t, rows, cols, ld):
    return torch.as_strided(t, (rows, cols), (ld, 1))


class CpuOps:
    """The subset of vltf_amd.ops that graph.py calls, with the same signatures, on CPU float32 tensors."""
    FUSION_CODE = {"avg": 0, "last": 1}

    @staticmethod
    def gemm(a, b, c, m, n, k, transa=False, transb=False, lda=None, ldb=None, ldc=None, bias=None, relu=False, relu_mask=None, ws=None):
        lda = lda if lda is not None else (m if transa else k)
        ldb = ldb if ldb is not None else (k if transb else n)
        ldc = ldc if ldc is not None else n
        A = _mat(a, k, m, lda).t() if transa else _mat(a, m, k, lda)
        B = _mat(b, n, k, ldb).t() if transb else _mat(b, k, n, ldb)
        out = A.double() @ B.double()
        if bias is not None:
            out = out + bias[:n].double()
        if relu:
            out = out.clamp_min(0)
        if relu_mask is not None:
            out = out * (_mat(relu_mask, m, n, ldc) > 0)
        _mat(c, m, n, ldc).copy_(out.float())

    @staticmethod
    def colsum(a, out, ws, m, n, lda=None):
        out.view(-1)[:n].copy_(_mat(a, m, n, lda if lda is not None else n).double().sum(0).float())

    @staticmethod
    def copy2d(src, dst, rows, cols, src_ld=None, dst_ld=None):
        s = torch.as_strided(src, (rows, cols), (cols if src_ld is None else src_ld, 1))
        _mat(dst, rows, cols, cols if dst_ld is None else dst_ld).copy_(s.clone())

    @staticmethod
    def eltwise2(a, b, out, op, count=None):
        n = a.numel() if count is None else count
        x, y = a.reshape(-1)[:n], b.reshape(-1)[:n]
        out.view(-1)[:n].copy_({"add": x + y, "avg": (x + y) * 0.5, "maximum": torch.maximum(x, y)}[op])

    @staticmethod
    def fuse_n(ins, out, method, count=None):
        n = ins[0].numel() if count is None else count
        st = torch.stack([t.reshape(-1)[:n] for t in ins])
        out.view(-1)[:n].copy_(st.mean(0) if method == "avg" else st.amax(0))

    @staticmethod
    def fuse_n_grad(ins, d, dins, method, count=None):
        n = d.numel() if count is None else count
        g = d.reshape(-1)[:n]
        if method == "avg":
            parts = [g / len(ins)] * len(ins)
        else:
            st = torch.stack([t.reshape(-1)[:n] for t in ins])
            hit = st == st.amax(0)
            parts = [g * hit[i] / hit.sum(0) for i in range(len(ins))]
        for t, p in zip(dins, parts):
            if t is not None:
                t.view(-1)[:n].copy_(p)

    @staticmethod
    def temporal_fusion_fwd(x, y, batch, T, H, method):
        x3 = x.reshape(-1)[:batch * T * H].view(batch, T, H)
        y.view(-1)[:batch * H].view(batch, H).copy_(x3.mean(1) if method == "avg" else x3[:, -1])

    @staticmethod
    def temporal_fusion_bwd(dy, dx, batch, T, H, method):
        d = dy.reshape(-1)[:batch * H].view(batch, 1, H)
        out = dx.view(-1)[:batch * T * H].view(batch, T, H)
        if method == "avg":
            out.copy_((d / T).expand(batch, T, H))
        else:
            out.zero_()
            out[:, -1] = d[:, 0]

    @staticmethod
    def lstm_seq_ws(batch, T, H, device):
        return torch.zeros(64)

    @staticmethod
    def lstm_seq_check(*ws):
        pass

    @staticmethod
    def lstm_seq_fwd(gx, kh, act, cseq, hseq, hprev, batch, T, H, forget_bias=1.0, ws=None, h0=None, c0=None):
        gx3 = gx.reshape(-1)[:batch * T * 4 * H].view(batch, T, 4 * H).double()
        Kh = _mat(kh, H, 4 * H, 4 * H).double()
        h = torch.zeros(batch, H, dtype=torch.float64) if h0 is None else h0[:batch].double()
        c = torch.zeros(batch, H, dtype=torch.float64) if c0 is None else c0[:batch].double()
        A, Cs, Hs, Hp = (t.view(-1)[:batch * T * w].view(batch, T, w) for t, w in ((act, 4 * H), (cseq, H), (hseq, H), (hprev, H)))
        for t in range(T):
            z = gx3[:, t] + h @ Kh
            i, j, f, o = z.chunk(4, 1)
            gi, gj, gf, go = torch.sigmoid(i), torch.tanh(j), torch.sigmoid(f + forget_bias), torch.sigmoid(o)
            Hp[:, t] = h.float()
            c = c * gf + gi * gj
            h = torch.tanh(c) * go
            A[:, t] = torch.cat([gi, gj, gf, go], 1).float()
            Cs[:, t] = c.float()
            Hs[:, t] = h.float()

    @staticmethod
    def lstm_seq_bwd(dout, kh, act, cseq, dz, batch, T, H, ws=None, c0=None, dh0=None, dc0=None):
        Kh = _mat(kh, H, 4 * H, 4 * H).double()
        D, A, Cs, Z = (t.view(-1)[:batch * T * w].view(batch, T, w) for t, w in ((dout, H), (act, 4 * H), (cseq, H), (dz, 4 * H)))
        dh = torch.zeros(batch, H, dtype=torch.float64)
        dc = torch.zeros(batch, H, dtype=torch.float64)
        for t in reversed(range(T)):
            gi, gj, gf, go = A[:, t].double().chunk(4, 1)
            c = Cs[:, t].double()
            cp = Cs[:, t - 1].double() if t > 0 else (torch.zeros_like(c) if c0 is None else c0[:batch].double())
            din = D[:, t].double() + dh
            tc = torch.tanh(c)
            dcv = dc + din * go * (1 - tc * tc)
            z = torch.cat([dcv * gj * gi * (1 - gi), dcv * gi * (1 - gj * gj), dcv * cp * gf * (1 - gf), din * tc * go * (1 - go)], 1)
            dc = dcv * gf
            Z[:, t] = z.float()
            dh = z @ Kh.t()
        if dh0 is not None:
            dh0[:batch].copy_(dh.float())
        if dc0 is not None:
            dc0[:batch].copy_(dc.float())

    @staticmethod
    def dropout_fwd(x, y, mask, keep, seed):
        raise AssertionError("the CPU double runs with dropout off")

    dropout_bwd = dropout_fwd

    @staticmethod
    def softmax_xent(logits, labels, dlogits, stats, grad_scale, rows=None):
        b = logits.shape[0]
        z = logits.double()
        logp = torch.log_softmax(z, 1)
        y = labels.double()
        stats[0] += float(-(y * logp).sum())
        stats[1] += float((z.argmax(1) == y.argmax(1)).sum())
        dlogits.view(-1)[:z.numel()].view_as(z).copy_(((logp.exp() - y) * grad_scale).float())

    @staticmethod
    def sumsq(g, out, ws, accumulate=False):
        out[0] = float((g.double() ** 2).sum()) + (float(out[0]) if accumulate else 0.0)

    @staticmethod
    def step_guard(skip, *lstm_workspaces):
        skip.zero_()
        return skip

    @staticmethod
    def sgd_apply(w, g, lr, clip_norm=0.0, sumsq_t=None, gscale=1.0, skip=None):
        if skip is not None and int(skip[0]):
            return
        scale = gscale
        if clip_norm > 0:
            gn = math.sqrt(float(sumsq_t[0]))
            scale *= clip_norm / max(gn, clip_norm)
        w -= lr * scale * g

    @staticmethod
    def adam_apply(*a, **k):
        raise AssertionError("not used by the CPU double")

    @staticmethod
    def fill(t, value):
        t.fill_(value)

    @staticmethod
    def set_conv_math(name):
        pass


class CpuTower:
    """Stands in for LRCNEngine(classifier "none"): frames -> AlexNet features (oracle.torch_cpu.dcnn_features + autograd), with the
    attributes graph.PipeNode touches: P / G views of the flat buffers, grad_chunks, x0, c8, logits, dlogits, dp, feed_u8,
    _forward, _backward."""

    def __init__(self, cfg, max_clips, device="cpu", training=True, dp=None, flat=None):
        from vltf_amd.engine import param_specs
        self.cfg, self.T, self.N = cfg, cfg.fpc, max_clips * cfg.fpc
        self.w, self.g = flat
        self.dp, self.c8, self.step_count = dp, False, 0
        self.specs = param_specs(cfg)
        self.P, self.G, off = {}, {}, 0
        for name, shp in self.specs:
            n = int(np.prod(shp))
            self.P[name] = self.w[off:off + n].view(shp)
            if training:
                self.G[name] = self.g[off:off + n].view(shp)
            off += n
        half = off // 2
        self.grad_chunks = [(0, half), (half, off - half)]
        h, w, c = cfg.image_shape
        self.x0 = torch.zeros((self.N, h, w, c))
        D = cfg.encode_dim()
        self.logits = torch.zeros((self.N, D))
        self.dlogits = torch.zeros((self.N, D))

    def feed_u8(self, frames_u8, mean_bgr=None, crop_y=None, crop_x=None, mirror=None, resize=None):
        n = frames_u8.shape[0]
        x = frames_u8.float()
        if mean_bgr is not None:
            x = x - torch.as_tensor(np.asarray(mean_bgr, np.float32))
        self.x0[:n].copy_(x)
        return n, n // self.T

    def _forward(self, n, b, train):
        self._leaves = {k: v.detach().clone().double().requires_grad_() for k, v in self.P.items()}
        self._feat = TC.dcnn_features(self._leaves, "", self.x0[:n].double(), self.cfg.frame_encoding_layer)
        self.logits[:n].copy_(self._feat.detach().float())
        return n

    def _backward(self, n, b):
        self._feat.backward(self.dlogits[:n].double())
        for k, v in self._leaves.items():
            self.G[k].copy_((v.grad if v.grad is not None else torch.zeros_like(v)).float())
        if self.dp is not None:
            for lo, cnt in self.grad_chunks:
                self.dp.reduce_async(self.g, lo, cnt)


def install(monkeypatch):
    """graph.ops / graph.LRCNEngine -> the CPU stand-ins; returns a GraphEngine subclass that accepts device "cpu"."""
    from vltf_amd import graph
    monkeypatch.setattr(graph, "ops", CpuOps)
    monkeypatch.setattr(graph, "LRCNEngine", CpuTower)

    class CpuGraphEngine(graph.GraphEngine):
        def _require_device(self):
            assert self.dev.type == "cpu"

        def _sync(self):
            pass

    return CpuGraphEngine
