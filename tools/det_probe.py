"""Run-to-run determinism of one train step: the same step from the same parameters `reps` times, every gradient (and a few
intermediate buffers) compared bitwise with the first run.  usage: det_probe.py <f32|bf16x3|bf16x6|bf16> <clips> <reps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vltf_amd.engine import LRCNEngine, NetConfig, init_params
math, clips, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
DEV = "cuda:0"
MEAN = np.array([99.197148, 105.293620, 109.503945], np.float32)
cfg = NetConfig(image_shape=(227, 227, 3), num_classes=101, fpc=16, lstm_hidden=256, conv_math=math)
eng = LRCNEngine(cfg, max_clips=clips, device=DEV)
p = init_params(cfg, seed=2)
rng = np.random.default_rng(0)
frames = torch.from_numpy(rng.integers(0, 256, (clips * 16, 227, 227, 3), dtype=np.uint8)).to(DEV)
lab = rng.integers(0, 101, clips)
onehot = torch.zeros((clips, 101), dtype=torch.int32); onehot[torch.arange(clips), torch.from_numpy(lab)] = 1
onehot = onehot.to(DEV)
ref = None
for r in range(reps):
    eng.load_params(p)
    out = eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN)
    g = eng.get_grads()
    n_ = clips * 16
    for nm, t in (("dp2", eng.layers[1]["dp"][:n_]), ("dy2", eng.layers[1]["dy"][:n_]), ("dy3", eng.layers[2]["dy"][:n_]), ("dp1", eng.layers[0]["dp"][:n_]),
                  ("wt3", eng.layers[2].get("wt")), ("wt2", eng.layers[1].get("wt"))):
        if t is not None:
            g["buf/" + nm] = t.detach().float().cpu().numpy().copy()
    if ref is None:
        ref = g; print("run 0 loss %.6f gn %.6f streams %s" % (out["loss"], out["grad_norm"], "two" if eng._side_stream() is not None else "one"), flush=True)
        continue
    diff = [(k, int((g[k] != ref[k]).sum()), g[k].size) for k in g if not np.array_equal(g[k], ref[k])]
    print("run", r, "loss %.6f gn %.6f" % (out["loss"], out["grad_norm"]), "differs:", diff, flush=True)
    if "buf/dy2" in g and not np.array_equal(g["buf/dy2"], ref["buf/dy2"]):
        idx = np.argwhere(g["buf/dy2"] != ref["buf/dy2"])
        print("  dy2 shape", g["buf/dy2"].shape, "first differing (n, c, h, w): ", idx[:12].tolist())
        print("  values now / ref:", [(float(g["buf/dy2"][tuple(i)]), float(ref["buf/dy2"][tuple(i)])) for i in idx[:6]])
        print("  distinct n:", sorted(set(idx[:, 0].tolist()))[:20], " distinct c:", sorted(set(idx[:, 1].tolist()))[:40])
        print("  distinct h:", sorted(set(idx[:, 2].tolist())), " distinct w:", sorted(set(idx[:, 3].tolist())))
