"""Multi-pipeline models at AlexNet's REAL layer shapes against committed fp64 oracle answers (tests/golden/graph_full.npz, written
by tests/golden/make_golden_graph.py from oracle.lrcn_oracle.model_forward / model_backward; cases in tests/graph_cases.py):

  c4_ws  BASELINE config 4 at its own shapes -- 8 clips x 16 frames of 227x227, AlexNet(fc6) + LSTM(256, state) => the state of a
         256-unit LSTM over 21 word vectors (300-d), per-step logits over 1000 words (models/model.py:128-141, lstm.py:34-42)
  ts_ws  the two-stream LRCN -- two towers x 4 clips x 16 frames, fc6 features averaged into LSTM(256)

One clipped-SGD step through vltf_amd.graph.GraphEngine: logits 1e-3, loss 1e-4, global and per-tensor gradient norms 2e-3, strided
samples and heads of every gradient (direction, 2e-2), the applied update -- the tolerances of the benchmark job's own fixture
(tests/test_full_workload_gpu.py).  The small-geometry forms of the same graphs are checked element-wise in tests/test_graph_gpu.py."""
import os

import numpy as np
import pytest
import torch

from tests import graph_cases as GC
from tests.test_full_workload_gpu import CLIP, LR, check_step_against_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_full.npz")


@pytest.mark.parametrize("name", sorted(GC.FULL_CASES))
def test_full_geometry_graph_step_matches_fixture(name):
    from vltf_amd.graph import GraphEngine, init_params_for, model_specs
    gold = np.load(GOLD)
    if name + "/logits" not in gold.files:
        pytest.fail("fixture case %s missing from graph_full.npz (run tests/golden/make_golden_graph.py)" % name)
    case = GC.FULL_CASES[name]()
    pipes, ds = GC.specs_and_datasets(case)
    p = init_params_for(model_specs(pipes, ds, case["V"]), seed=case["seed"], well_scaled=True)
    eng = GraphEngine(pipes, ds, case["V"], device=DEV)
    assert [n for n, _ in eng.specs] == [n for n, _ in model_specs(pipes, ds, case["V"])]
    eng.load_params(p)
    raw, _ = GC.inputs(case)
    fd = {t: (dict(frames_u8=torch.from_numpy(v).to(DEV), mean_bgr=GC.MEAN) if v.dtype == np.uint8 else torch.from_numpy(v).to(DEV))
          for t, v in raw.items()}
    logits = eng.forward(fd).cpu().numpy()
    rows = logits.shape[0]
    assert logits.shape == gold[name + "/logits"].shape
    # labels: graph_cases.expect's stream
    lab = np.random.default_rng(0).integers(0, case["V"], rows)
    onehot = np.zeros((rows, case["V"]), np.int32)
    onehot[np.arange(rows), lab] = 1
    out = eng.train_step(fd, torch.from_numpy(onehot).to(DEV), lr=LR, clip_norm=CLIP)
    eng.check_status()
    check_step_against_fixture(gold, name, p, rows, out, logits, eng.get_grads(), eng.get_params())
    towers = [k for k in p if k.endswith("dcnn/conv1W")]
    assert towers and all(float(gold["%s/gradnorm/%s" % (name, k)][0]) > 0 for k in towers)
