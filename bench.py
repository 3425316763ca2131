#!/usr/bin/env python3
"""bench.py -- clips/s of one full LRCN train step (input-prep -> forward -> loss -> backward ->
[all-reduce] -> global-norm clip + SGD) on synthetic 16-frame 227x227x3 clips (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its N ranks itself, dp.self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1 under a launcher)

Workload (configs[1] of BASELINE.json): AlexNet encode at fc6 -> LSTM(256, 1 layer, avg) -> 101 classes, T = 16, the
reference's batch_size = 64 clips (= 1024 frames) per step, fp32, dropout keep 0.5.  N = 1: the whole batch on one
GPU.  N > 1 (configs[2]): STRONG scaling -- the same global batch of 64 clips split over the ranks (8 clips per GPU at N = 8),
RCCL all-reduce of the 178 MB gradient in chunks that overlap the backward pass; the weak-scaling form (64 clips on every rank,
global batch 64*N) is timed in the same run and reported beside it (`weak_scaling`; `--weak` makes it the headline instead).
Inputs are resident in HBM before timing starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# forward MACs per frame; dgrad and wgrad of a layer have the same count
CONV_MACS = {"conv1": 113221152, "conv2": 240844800, "conv3": 149520384, "conv4": 112140288, "conv5": 74760192}
FC6_MACS = 9216 * 4096                     # per frame
LSTM_STEP_MACS = (4096 + 256) * 1024       # per clip and step (kernel [D+H, 4H])


def clip_flops(fpc, classes=101):
    """(forward, train) FLOP per clip (FLOP = 2 MAC; train = 3 x forward minus conv1's absent dgrad).
    fpc = 16, 101 classes: 2 x 11,723,097,856 and 2 x 33,357,755,136."""
    fwd = fpc * (sum(CONV_MACS.values()) + FC6_MACS + LSTM_STEP_MACS) + 256 * classes
    return 2 * fwd, 2 * (3 * fwd - fpc * CONV_MACS["conv1"])


PEAK_FP32_MFMA_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0             # MI355X_MICROARCH.md: bf16 dense (v_mfma_f32_32x32x16_bf16); only used with --conv-math
# launches that share the single-kernel symbol conv_dma_kernel<32> (output-channel tile 128, LDS-DMA operand rows:
# conv2/3/5 forward and conv3 dgrad, whose output channel count is 256)
DOMINANT = ("conv2.fwd", "conv3.fwd", "conv5.fwd", "conv3.dgrad")
DOMINANT_SYMBOL = "conv_dma_kernel<32>"
# algorithmic HBM bytes of those launches per frame: input + weights (per launch) + output, fp32, interiors only
DOMINANT_ELEMS_PER_FRAME = {"conv2.fwd": 96 * 28 * 28 + 256 * 28 * 28, "conv3.fwd": 256 * 169 + 384 * 169,
                            "conv5.fwd": 384 * 169 + 256 * 169, "conv3.dgrad": 384 * 169 + 256 * 169}
DOMINANT_WEIGHT_ELEMS = {"conv2.fwd": 5 * 5 * 48 * 256, "conv3.fwd": 3 * 3 * 256 * 384, "conv5.fwd": 3 * 3 * 192 * 256,
                         "conv3.dgrad": 3 * 3 * 256 * 384}
MEAN_BGR = np.array([99.197148, 105.293620, 109.503945], np.float32)


def committed_traffic(symbol_prefix):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile_round.sh (FETCH_SIZE and WRITE_SIZE
    cannot be collected inside this process): profiles/r*_bench_n1_traffic.json, newest round first.  None if absent or
    taken on a different kernel symbol."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        if str(rec.get("kernel", "")).startswith(symbol_prefix):
            return rec, os.path.relpath(f, ROOT)
    return None, None


def host_cores():
    """Cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants 16 cores per GPU; 256 torch threads on that quota ran the baseline 25x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(clips, fpc, num_classes):
    """The CPU restatement of the reference's op graph (kind 'port': TensorFlow itself is not installable here) timed on this
    box's host cores on a bounded sample of the same workload: oracle/torch_cpu.py (torch functional ops + autograd, fp32,
    oneDNN/MKL on every core) -- the multi-threaded form of the oracle, pinned against the numpy oracle in tests/test_oracle.py.
    One untimed step, then whole steps until ~12 s have passed."""
    from oracle import lrcn_oracle as O
    from oracle import torch_cpu as TC
    torch.set_num_threads(host_cores())
    rng = np.random.default_rng(2)
    p = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True)
         for k, v in O.init_params(rng, num_classes, "fc6", 256, 1, (227, 227, 3)).items()}
    frames = np.random.default_rng(0).integers(0, 256, (clips * fpc, 227, 227, 3), dtype=np.uint8)
    x = torch.from_numpy(frames.astype(np.float32) - MEAN_BGR)
    lab = torch.from_numpy(np.random.default_rng(1).integers(0, num_classes, clips))
    TC.train_step(p, x, lab, fpc, lr=1e-3, clip_norm=10.0)
    steps, t0 = 0, time.time()
    while steps < 1 or time.time() - t0 < 12.0:
        TC.train_step(p, x, lab, fpc, lr=1e-3, clip_norm=10.0)
        steps += 1
    dt = time.time() - t0
    return {"value": clips * steps / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d train steps (fwd+bwd+clip+SGD) of %d clips x %d frames 227x227, torch-CPU fp32 restatement of the "
                      "reference's TF op graph, %.1f s" % (steps, clips, fpc, dt)}


def oracle_first_step(eng, frames, onehot, clips, fpc, classes):
    """First step from init_params(seed=2) against the CPU oracle's committed answer for exactly these inputs
    (tests/golden/lrcn_full.npz case cfg2_ref, made by tests/golden/make_golden_full.py; a data file, not the oracle).
    Dropout is switched off for this one step (the oracle has no TF RNG) and the parameters are re-loaded afterwards."""
    path = os.path.join(ROOT, "tests", "golden", "lrcn_full.npz")
    if (clips, fpc, classes) != (64, 16, 101) or not os.path.exists(path):
        return None
    gold = np.load(path)
    if "cfg2_ref/logits" not in gold.files:
        return None
    keep = eng.cfg.dropout_keep_prob
    eng.cfg.dropout_keep_prob = 0.0
    out = eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=True)
    logits = eng.logits_host()
    eng.cfg.dropout_keep_prob = keep
    loss, gn, _ = gold["cfg2_ref/loss_gn_acc"]
    return {"source": "tests/golden/lrcn_full.npz:cfg2_ref (fp64 CPU oracle on these inputs)",
            "loss": [round(out["loss"], 6), round(float(loss), 6)], "grad_norm": [round(out["grad_norm"], 5), round(float(gn), 5)],
            "max_abs_dlogit": float(np.abs(logits - gold["cfg2_ref/logits"]).max()),
            "ok": bool(abs(out["loss"] - loss) < 1e-4 * max(1.0, abs(loss)) and abs(out["grad_norm"] - gn) < 2e-3 * gn
                       and np.abs(logits - gold["cfg2_ref/logits"]).max() < 1e-3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=64, help="the reference's batch_size: clips per step over ALL ranks (strong scaling)")
    ap.add_argument("--clips-per-gpu", type=int, default=0, help="N = 1: run this many clips instead of --global-batch (e.g. 8 = one rank's "
                                                                 "shard of the 8-GPU job); with --weak: clips on every rank")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling (every rank runs a whole batch) is the headline, strong the side field")
    ap.add_argument("--fpc", type=int, default=16)
    ap.add_argument("--classes", type=int, default=101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=16)
    ap.add_argument("--dropout", type=float, default=0.5, help="dropout keep probability of the timed steps (SURVEY 8d: 0.5; 0 disables)")
    ap.add_argument("--no-side", action="store_true", help="N > 1: skip the other scaling form's measurement")
    ap.add_argument("--conv-math", default="f32", choices=["f32", "bf16x3", "bf16x6", "bf16"],
                    help="arithmetic of the MAIN measurement (default f32 = the headline; anything else is labelled in `dtype`)")
    ap.add_argument("--no-split-math", action="store_true", help="N = 1: skip the extra bf16x3 / bf16x6 / bf16 (opt-in conv arithmetic) measurements")
    args = ap.parse_args()

    from vltf_amd import dp as dpmod
    # `python bench.py --gpus N` as the driver starts it (no launcher): become the launcher -- N ranks under torch.distributed.run --
    # before anything here touches the GPU; their one JSON line (rank 0's) passes through, their exit code is ours
    # ... and, first of all, send a tiny GPU process ahead: the first GPU process on a freshly acquired box runs short kernels slower
    # for its whole life (dp.pretouch_gpu; the ranks inherit the marker and do not repeat it)
    pretouched = False if os.environ.get("VLTF_BENCH_RENDEZVOUS_ONLY") == "1" else dpmod.pretouch_gpu(args.gpus)
    rc = dpmod.self_launch(args.gpus)
    if rc is not None:
        raise SystemExit(rc)
    from vltf_amd.engine import LRCNEngine, NetConfig, init_params

    if os.environ.get("VLTF_BENCH_RENDEZVOUS_ONLY") == "1":
        # launch-path test (tests/test_launch.py, no GPU needed): the ranks meet over the backend of VLTF_DIST_BACKEND, sum their ranks,
        # rank 0 reports and everybody leaves -- nothing is measured
        rank, world, _ = dpmod.init_from_env()
        t = torch.tensor([float(rank)])
        if world > 1:
            torch.distributed.all_reduce(t)
            torch.distributed.destroy_process_group()
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "world": world, "gpus": args.gpus, "rank_sum": float(t.item())}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    rank, world, local = dpmod.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus disagree" % (args.gpus, world))
    local = local % torch.cuda.device_count()      # ranks beyond the visible GPUs share them (gloo rehearsal on a 1-GPU box only)
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)
    gar = dpmod.GradAllReduce() if world > 1 else None
    cfg = NetConfig(image_shape=(227, 227, 3), num_classes=args.classes, fpc=args.fpc, frame_encoding_layer="fc6",
                    classifier="lstm", lstm_hidden=256, lstm_layers=1, fusion="avg", dropout_keep_prob=args.dropout,
                    conv_math=args.conv_math)
    fwd_flop, train_flop = clip_flops(args.fpc, args.classes)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def job(mode):
        """(clips on this rank, global clips) of a scaling form.  strong: the global batch split over the ranks; weak: a whole
        batch on every rank."""
        if world == 1 or mode == "weak":
            c = args.clips_per_gpu or args.global_batch
            return c, c * world
        lo, hi = dpmod.shard_range(args.global_batch, rank, world)
        return hi - lo, args.global_batch

    def inputs(clips):
        n = clips * args.fpc
        # rank r's clips: its own stream of synthetic frames (rank 0 = the inputs of tests/golden/lrcn_full.npz)
        fr = torch.from_numpy(np.random.default_rng(rank).integers(0, 256, (max(n, 1), 227, 227, 3), dtype=np.uint8)).to(dev)[:n]
        lab = np.random.default_rng(1000 + rank).integers(0, args.classes, clips)
        oh = torch.zeros((clips, args.classes), dtype=torch.int32)
        oh[torch.arange(clips), torch.from_numpy(lab)] = 1
        return fr, oh.to(dev)

    def build(clips):
        eng = LRCNEngine(cfg, max_clips=max(clips, 1), device=dev, dp=gar)
        eng.load_params(init_params(cfg, seed=2))          # reference initialisers; same seed on every rank
        if gar is not None:
            gar.broadcast_params(eng.w)
        return eng

    host_issue = [0.0]

    def timed(eng, frames, onehot, clips, global_clips, probe):
        """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        def step(fetch=False):
            if clips == 0:                                 # more ranks than clips: this rank only joins the exchange
                return eng.train_step_empty(1e-3, 10.0, fetch=fetch)
            return eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=fetch,
                                     global_rows=global_clips if world > 1 else None)
        for _ in range(args.warmup):
            step()
        if probe:
            eng.set_probe([l + p for l in CONV_MACS for p in (".fwd", ".dgrad", ".wgrad")])
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        host_issue[0] = (time.perf_counter() - t0) / args.steps * 1e3      # ms per step the host needed to ENQUEUE (no device wait in it
        torch.cuda.synchronize()                                             # unless the queue fills): close to ms_per_step = host-bound
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        times = eng.probe_times_ms() if probe else []
        eng.set_probe(None)
        return elapsed, times, step(fetch=True)

    scaling = "weak" if (args.weak and world > 1) else "strong"
    clips, total_clips = job(scaling)
    eng = build(clips)
    frames, onehot = inputs(clips)
    n = clips * args.fpc
    oracle_check = None
    if world == 1 and args.conv_math == "f32":
        oracle_check = oracle_first_step(eng, frames, onehot, clips, args.fpc, args.classes)
        eng.load_params(init_params(cfg, seed=2))
        eng.step_count = 0
    eng_chunks = list(eng.grad_chunks)
    elapsed, times, out = timed(eng, frames, onehot, clips, total_clips, probe=True)
    host_issue_main = host_issue[0]
    overlapped = clips > 0 and eng._side_stream() is not None      # the backward ran its independent launches on two streams
    serial_times = None
    if overlapped and world == 1:
        # per-launch times of the conv stack need launches that run alone: the same steps once more with the second stream off
        # (after the timed region, not part of `value`)
        keep = os.environ.get("VLTF_WGRAD_STREAM")
        os.environ["VLTF_WGRAD_STREAM"] = "0"
        try:
            for _ in range(2):
                eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=False)
            eng.set_probe([l + p for l in CONV_MACS for p in (".fwd", ".dgrad", ".wgrad")])
            for _ in range(args.steps):
                eng.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=False)
            torch.cuda.synchronize()
            serial_times = eng.probe_times_ms()
            eng.set_probe(None)
        finally:
            if keep is None:
                del os.environ["VLTF_WGRAD_STREAM"]
            else:
                os.environ["VLTF_WGRAD_STREAM"] = keep
    # forward only (sess.run(model.logits), run_task.py:95; SURVEY 8d asks for it beside the train step): per rank, untimed for `value`
    fwd_ms = None
    if clips > 0:
        for _ in range(2):
            eng.forward_u8(frames, mean_bgr=MEAN_BGR)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(args.steps):
            eng.forward_u8(frames, mean_bgr=MEAN_BGR)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - tf0) / args.steps * 1e3

    # Opt-in arithmetics (NetConfig.conv_math = "bf16x3": the three conv contractions as split bf16 products; "bf16": plain bf16
    # products, BASELINE config 5's reduced-precision conv path; everything else unchanged):
    # the same job timed the same way, reported BESIDE `value` (which stays the fp32 path), with its own loss / gradient norm
    # after the same number of steps from the same initial parameters as evidence of what the arithmetic changes.
    split, split6, plain = None, None, None
    if world == 1 and not args.no_split_math and args.conv_math == "f32":
        import dataclasses

        def first_step(e):
            keep, e.cfg.dropout_keep_prob = e.cfg.dropout_keep_prob, 0.0
            r = e.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=True)
            e.cfg.dropout_keep_prob = keep
            return r

        def side_run(math, first_of):
            eng_x = LRCNEngine(dataclasses.replace(cfg, conv_math=math), max_clips=clips, device=dev)
            eng_x.load_params(init_params(cfg, seed=2))
            # first step from the SAME parameters in both arithmetics, dropout off (later steps of this chaotic sigma-0.05
            # initialisation drift apart under any rounding difference, so only the first one is a like-for-like comparison)
            first = dict(first_of)
            first[math] = first_step(eng_x)
            for _ in range(args.warmup):
                eng_x.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=False)
            eng_x.set_probe([l + p for l in CONV_MACS for p in (".fwd", ".dgrad", ".wgrad")])
            torch.cuda.synchronize()
            tx0 = time.perf_counter()
            for _ in range(args.steps):
                eng_x.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=False)
            torch.cuda.synchronize()
            tx = time.perf_counter() - tx0
            per_x = {}
            for label, ms in eng_x.probe_times_ms():
                per_x.setdefault(label, []).append(ms)
            eng_x.set_probe(None)
            out_x = eng_x.train_step_u8(frames, onehot, lr=1e-3, clip_norm=10.0, mean_bgr=MEAN_BGR, fetch=True)
            del eng_x
            what = {"bf16x3": "split bf16 products, 2 pieces / 3 products", "bf16x6": "split bf16 products, 3 pieces / 6 products",
                    "bf16": "the bf16 path: packed-bf16 operands in memory (csrc/conv_c8.hip), v_mfma_f32_32x32x16_bf16, reduced precision"}[math]
            return {"value": round(clips * args.steps / tx, 2), "unit": "clips/s", "ms_per_step": round(tx / args.steps * 1e3, 3),
                    "dtype": "%s (conv fwd/dgrad/wgrad: %s, fp32 accumulate) + f32 (dense GEMMs, LSTM, pointwise)" % (math, what),
                    "per_launch_ms": {k: round(sum(v) / len(v), 3) for k, v in sorted(per_x.items())},
                    "check": {"loss": round(out_x["loss"], 4), "grad_norm": round(out_x["grad_norm"], 3)},
                    "first_step_check": {k: {"loss": round(v["loss"], 5), "grad_norm": round(v["grad_norm"], 4)} for k, v in first.items()},
                    "note": "opt-in NetConfig.conv_math; never `value`"}

        eng.load_params(init_params(cfg, seed=2))
        f32_first = {"f32": first_step(eng)}
        split = side_run("bf16x3", f32_first)
        split6 = side_run("bf16x6", f32_first)
        plain = side_run("bf16", f32_first)

    # N > 1: the other scaling form of the same job, timed the same way, reported beside `value`
    side = None
    if world > 1 and not args.no_side:
        other = "weak" if scaling == "strong" else "strong"
        sc, stotal = job(other)
        del eng
        eng_s = build(sc)
        f_s, o_s = inputs(sc)
        es, _, _ = timed(eng_s, f_s, o_s, sc, stotal, probe=False)
        side = {"scaling": other, "global_batch": stotal, "clips_per_gpu": sc, "value": round(stotal * args.steps / es, 2),
                "unit": "clips/s", "ms_per_step": round(es / args.steps * 1e3, 3)}
        del eng_s
    if rank != 0:
        return

    # ---- roofline of the dominant kernel symbol, from the live HIP-event brackets --------------------
    per = {}
    for label, ms in times:
        per.setdefault(label, []).append(ms)
    avg = {k: sum(v) / len(v) for k, v in per.items()}
    # The fp32 backward runs its independent launches on two streams (engine._side_stream): a bracket around a launch that has a
    # neighbour times both.  The dominant symbol is measured live, in the timed region, on the launches that run alone there (the
    # forward ones: 3 of its 4 launches per step); the conv-stack table comes from the serial pass above.
    dominant = tuple(l for l in DOMINANT if not overlapped or l.endswith(".fwd"))
    dom_flop = sum(2.0 * CONV_MACS[l.split(".")[0]] * n for l in dominant) / len(dominant)      # per launch
    dom_ms = sum(avg[l] for l in dominant) / len(dominant)                                       # per launch
    achieved = dom_flop / (dom_ms * 1e-3) / 1e12
    timed_avg = avg
    if serial_times is not None:
        per_s = {}
        for label, ms in serial_times:
            per_s.setdefault(label, []).append(ms)
        avg = {k: sum(v) / len(v) for k, v in per_s.items()}
    stack_flop = sum(2.0 * CONV_MACS[l.split(".")[0]] * n for l in avg)
    stack_ms = sum(avg.values())
    traffic_rec, traffic_src = committed_traffic(DOMINANT_SYMBOL)
    alg_bytes = sum(4.0 * (DOMINANT_ELEMS_PER_FRAME[l] * n + DOMINANT_WEIGHT_ELEMS[l]) for l in dominant) / len(dominant)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_clips * args.steps / elapsed
    # --conv-math other than f32: the same launches run the split-product kernels; their bound is the bf16 matrix pipe at one
    # MFMA per product (bf16x3: 3, bf16x6: 6), so the peak of an fp32-equivalent FLOP is the dense bf16 peak / products
    f32_main = args.conv_math == "f32"
    peak = PEAK_FP32_MFMA_TFLOPS if f32_main else PEAK_BF16_MFMA_TFLOPS / {"bf16x3": 3, "bf16x6": 6, "bf16": 1}[args.conv_math]
    rec = {
        "metric": "clips/sec (16-frame 227x227) LRCN train step" if args.fpc == 16 else
                  "clips/sec (%d-frame 227x227) LRCN train step" % args.fpc,
        "value": round(value, 2), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "host_issue_ms_per_step": round(host_issue_main, 3),
        "gpu_pretouch_process": bool(pretouched or os.environ.get("VLTF_GPU_PRETOUCHED") == "1"),   # dp.pretouch_gpu ran before this process      # this rank's host time to enqueue a step (<< ms_per_step: not launch-bound)
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32" if args.conv_math == "f32" else "%s conv / GEMM products (opt-in, --conv-math), fp32 accumulate and elsewhere" % args.conv_math,
        "data": "synthetic",
        "config": {"workload": "LRCN AlexNet(fc6)+LSTM(256) %d-frame 227x227x3 clips, %d classes, UCF-101-shaped synthetic, "
                               "full train step" % (args.fpc, args.classes),
                   "global_batch": total_clips, "clips_per_gpu": clips, "frames_per_clip": args.fpc,
                   "parallelism": "dp%d" % world, "optimizer": "sgd+clip_by_global_norm(10)", "init": "reference (sigma 0.05)",
                   "dropout_keep_prob": args.dropout,
                   # what torch.distributed itself reports (so that a multi-GPU record shows the collective saw N ranks over RCCL)
                   "dist_world_size": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                   "dist_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                   "grad_exchange": None if gar is None else {"all_reduces_issued": gar.issued, "chunks_per_step": len(eng_chunks)}},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4),
                     "traffic": traffic_rec["bytes_per_launch"] if (traffic_rec and n == 1024 and f32_main) else None,
                     "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                     "traffic_source": traffic_src if n == 1024 else None,
                     "algorithmic_bytes_per_launch": alg_bytes if f32_main else None,
                     "kernel": DOMINANT_SYMBOL if f32_main else
                               ("conv_c8_kernel<4, 2, 2, 2> (packed-bf16 operands, csrc/conv_c8.hip)" if args.conv_math == "bf16" else
                                "conv_ring4_kernel / conv_ring_kernel (%s)" % args.conv_math),
                     "launches": list(dominant),
                     "backward_overlap": None if not overlapped else
                                         ("conv weight gradients (and fc6's input gradient) run on a second stream beside the input "
                                          "gradients; forward launches run alone"
                                          + ("; conv_stack below = the same steps with the second stream off (%d extra steps after "
                                             "the timed region)" % args.steps if serial_times is not None else
                                             "; conv_stack is not a serial sum")),
                     "per_launch_ms_in_timed_region": {k: round(v, 3) for k, v in sorted(timed_avg.items())} if overlapped else None,
                     "flop_per_launch": dom_flop, "ms_per_launch": round(dom_ms, 4),
                     "conv_stack": {"tflops": round(stack_flop / (stack_ms * 1e-3) / 1e12, 2),
                                    "frac": round(stack_flop / (stack_ms * 1e-3) / 1e12 / peak, 4),
                                    "ms_per_step": round(stack_ms, 3),
                                    "per_launch_ms": {k: round(v, 3) for k, v in sorted(avg.items())}},
                     "flop_per_clip_train": train_flop,
                     "step_frac_of_mfma_roofline": round(value / world * train_flop / 1e12 / peak, 4)},
        "forward_only": None if fwd_ms is None else
                        {"value": round(clips / (fwd_ms * 1e-3), 2), "unit": "clips/s per GPU", "ms_per_batch": round(fwd_ms, 3),
                         "frac_of_mfma_roofline": round(clips / (fwd_ms * 1e-3) * fwd_flop / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
        "oracle_check": oracle_check,
        ("weak_scaling" if scaling == "strong" else "strong_scaling"): side,
        "bf16x3": split,
        "bf16x6": split6,
        "bf16": plain,
        "check": {"loss": round(out["loss"], 4), "grad_norm": round(out["grad_norm"], 3)},
    }
    if world == 1 and not args.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(args.cpu_clips, args.fpc, args.classes)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
