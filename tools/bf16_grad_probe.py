#!/usr/bin/env python3
"""Where does the bf16 conv path's first-step gradient differ from fp32's?  (BENCH_r02 bf16.first_step_check reported a
gradient norm of 45.6 against 32.7 in fp32 on the benchmark's sigma = 0.05 initialisation.)

CPU part (no GPU needed): the fp64 oracle on rank 0's 8-clip shard of the benchmark job (fixture case c3shard_ref's inputs), once
exact and once with the bf16 path's operand roundings (oracle q = bf16_round: every tensor that path stores as packed bf16), per-tensor
gradient norms side by side -- is the difference a property of the ARITHMETIC or of the kernels?
GPU part (--gpu): the engine with conv_math="bf16" on the same inputs; its gradients against the rounding-aware oracle evaluated
with the device's own ReLU / arg-max decisions (tests/test_full_workload_gpu.py:device_gates), relative L2 per tensor.

  python tools/bf16_grad_probe.py [--clips 8] [--ws] [--gpu]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lrcn_oracle as O  # noqa: E402

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--fpc", type=int, default=16)
    ap.add_argument("--ws", action="store_true", help="well-scaled weights instead of the reference's sigma = 0.05")
    ap.add_argument("--gpu", action="store_true")
    args = ap.parse_args()
    from vltf_amd.engine import NetConfig, init_params
    cfg = NetConfig(image_shape=(227, 227, 3), num_classes=101, fpc=args.fpc, lstm_hidden=256)
    p = init_params(cfg, seed=2, well_scaled=args.ws)
    frames = np.random.default_rng(0).integers(0, 256, (args.clips * args.fpc, 227, 227, 3), dtype=np.uint8)
    onehot = O.labels_to_one_hot([[l] for l in np.random.default_rng(1000).integers(0, 101, args.clips)], 101)
    x = frames.astype(np.float32) - MEAN
    res = {}
    gates = None
    if args.gpu:
        import dataclasses
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_full_workload_gpu import device_gates_bf16
        from vltf_amd.engine import LRCNEngine
        eng = LRCNEngine(dataclasses.replace(cfg, conv_math="bf16"), max_clips=args.clips, device="cuda:0")
        eng.load_params(p)
        out = eng.train_step_u8(torch.from_numpy(frames).to("cuda:0"), torch.from_numpy(onehot).to("cuda:0"), lr=0.0, clip_norm=0.0, mean_bgr=MEAN)
        res["device bf16"] = (out["loss"], out["grad_norm"], eng.get_grads(), eng.logits_host())
        gates = device_gates_bf16(eng, args.clips * args.fpc)
        acts = device_activations_bf16(eng, args.clips * args.fpc)
    for tag, q in (("oracle exact", None), ("oracle bf16-rounded", O.bf16_round)):
        logits, cache = O.lrcn_forward(p, x, args.fpc, keep=True, chunk=32, q=q)
        loss, dlogits = O.softmax_xent_mean(logits, onehot)
        g = O.lrcn_backward(p, cache, dlogits, args.fpc, q=q)
        _, gn = O.clip_by_global_norm(g, 0.0)
        res[tag] = (loss, gn, g, logits)
        if args.gpu:
            print("forward activations, device bf16 path vs %s (relative L2):" % tag)
            want = {k: np.concatenate([c[k] for c in cache["cnn"]], axis=0) for k in ("conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "conv5", "pool5", "fc6")}
            want["lstm_out"] = np.stack([np.tanh(cs) * gt[3] for cs, gt in zip(cache["lstm"][0]["cs"], cache["lstm"][0]["gates"])], 1)
            want["fused"] = cache["fused"]
            for k, v in acts.items():
                w = want[k].reshape(v.shape)
                print("   %-10s %.3e" % (k, np.linalg.norm((v - w).ravel()) / (np.linalg.norm(w.ravel()) + 1e-300)))
        if q is not None and gates is not None:
            gg = O.lrcn_backward(p, cache, dlogits, args.fpc, q=q, gates=gates)
            _, gn2 = O.clip_by_global_norm(gg, 0.0)
            res["oracle bf16-rounded, device gates"] = (loss, gn2, gg, logits)
        del cache
    tags = list(res)
    print("%-44s" % "" + "".join("%24s" % t[:23] for t in tags))
    print("%-44s" % "loss" + "".join("%24.6f" % res[t][0] for t in tags))
    print("%-44s" % "global gradient norm" + "".join("%24.4f" % res[t][1] for t in tags))
    for k in sorted(p):
        print("%-44s" % k + "".join("%24.5e" % np.linalg.norm(res[t][2][k].astype(np.float64)) for t in tags))
    if args.gpu:
        print("\nrelative L2 error of the device's bf16-path gradients per tensor, against each oracle form:")
        for k in sorted(p):
            d = res["device bf16"][2][k].astype(np.float64)
            print("%-44s" % k + "".join("%24.3e" % (np.linalg.norm((d - res[t][2][k]).ravel()) / (np.linalg.norm(res[t][2][k].ravel()) + 1e-300))
                                         for t in tags if t != "device bf16"))
        print("max |dlogit| device vs oracle bf16-rounded: %.3e, vs exact: %.3e" % (
            np.abs(res["device bf16"][3] - res["oracle bf16-rounded"][3]).max(), np.abs(res["device bf16"][3] - res["oracle exact"][3]).max()))


def device_activations_bf16(eng, n):
    """Forward tensors of the engine's last step in the oracle's layouts (NHWC / row-major), unpacked from wherever the bf16 path keeps them."""
    import torch
    torch.cuda.synchronize()

    def unpack(t, c, halo):
        a = t.float()
        nn, cb, hp, wp, _ = a.shape
        full = a.permute(0, 2, 3, 1, 4).reshape(nn, hp, wp, cb * 8)
        return full[:, halo:hp - halo, halo:wp - halo, :c].cpu().numpy().astype(np.float64)
    L = eng.layers
    out = {"conv1": unpack(L[0]["yb"][:n], 96, 0), "pool1": unpack(L[1]["xb"][:n], 96, L[1]["x_halo"]),
           "conv2": unpack(L[1]["yb"][:n], 256, 0), "pool2": unpack(L[2]["xb"][:n], 256, L[2]["x_halo"]),
           "conv3": unpack(L[3]["xb"][:n], 384, L[3]["x_halo"]), "conv4": unpack(L[4]["xb"][:n], 384, L[4]["x_halo"]),
           "conv5": L[4]["y"][:n].permute(0, 2, 3, 1).cpu().numpy().astype(np.float64),
           "pool5": L[4]["p"][:n].cpu().numpy().astype(np.float64),
           "fc6": eng.f6[:n].cpu().numpy().astype(np.float64),
           "lstm_out": eng.lstm[0]["hseq"][:n].cpu().numpy().astype(np.float64),
           "fused": eng.fused[:n // eng.T].cpu().numpy().astype(np.float64)}
    return out


if __name__ == "__main__":
    main()
