#!/usr/bin/env python3
"""How far the conv arithmetic moves the gradients of ONE train step, next to what a different fp32 summation order does.

  math_noise.py run <f32|bf16x3|bf16> <out.npz> [clips] [init]   one step (lr 0) of the benchmark network, gradients -> out.npz
  math_noise.py cmp <ref.npz> <other.npz> [...]                  per-tensor relative L2 difference against ref

init = "scaled" (well-scaled weights, the parity tests' regime) or "reference" (the reference's sigma-0.05 initialiser, the
benchmark's regime, where pre-activations are ~1e3 and a rounding-level change flips ReLU gates / pool arg-max by the thousand).
Run the fp32 case twice, once with VL_CONV_STAGED=1 (the register-staged kernels: same arithmetic, other summation order), to get
the yardstick: a conv arithmetic is "fp32-equivalent" for training if it moves the gradients no more than that does.
GPU; used for DESIGN.md 4.6."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)


def run(math, out, clips=16, init="scaled"):
    import torch
    from oracle import lrcn_oracle as O
    from vltf_amd.engine import LRCNEngine, NetConfig, init_params
    cfg = NetConfig(image_shape=(227, 227, 3), num_classes=101, fpc=16, conv_math=math)
    eng = LRCNEngine(cfg, max_clips=clips, device="cuda:0")
    if init == "reference":
        eng.load_params(init_params(cfg, seed=2))
    else:
        eng.load_params(O.init_params(np.random.default_rng(2), 101, "fc6", 256, 1, (227, 227, 3), well_scaled=True))
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, (clips * 16, 227, 227, 3), dtype=np.uint8)).to("cuda:0")
    onehot = torch.zeros((clips, 101), dtype=torch.int32)
    onehot[torch.arange(clips), torch.from_numpy(rng.integers(0, 101, clips))] = 1
    o = eng.train_step_u8(frames, onehot.to("cuda:0"), lr=0.0, clip_norm=0.0, mean_bgr=MEAN)
    g = eng.get_grads()
    np.savez(out, loss=np.float64(o["loss"]), grad_norm=np.float64(o["grad_norm"]), logits=eng.logits_host(), **{k.replace("/", "__"): v for k, v in g.items()})
    print("%s %s: loss %.6f grad_norm %.5f -> %s" % (math, init, o["loss"], o["grad_norm"], out))


def cmp(ref, others):
    r = np.load(ref)
    keys = [k for k in r.files if k not in ("loss", "grad_norm")]
    print("%-28s" % "tensor" + "".join("%22s" % os.path.basename(o)[:21] for o in others))
    loaded = [np.load(o) for o in others]
    print("%-28s" % "loss / grad_norm (ref %.6f / %.5f)" % (float(r["loss"]), float(r["grad_norm"])))
    print("%-28s" % "" + "".join("%22s" % ("%.6f / %.5f" % (float(l["loss"]), float(l["grad_norm"]))) for l in loaded))
    for k in keys:
        den = np.linalg.norm(r[k].ravel()) + 1e-30
        print("%-28s" % k.replace("__", "/") + "".join("%22.3e" % (np.linalg.norm((l[k] - r[k]).ravel()) / den) for l in loaded))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 16, sys.argv[5] if len(sys.argv) > 5 else "scaled")
    else:
        cmp(sys.argv[2], sys.argv[3:])
