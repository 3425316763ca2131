"""Golden fixtures (tests/golden/lrcn_small.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them.  GPU: the HIP engine reproduces them through the C-ABI."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import lrcn_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
G = np.load(os.path.join(HERE, "golden", "lrcn_small.npz"))


@pytest.mark.parametrize("name", list(mg.CASES))
def test_oracle_reproduces_golden(name):
    shape, ncls, fpc, b, layer, hid, layers, fusion, seed = mg.CASES[name]
    p, frames, onehot = mg.case_inputs(name)
    x = frames.astype(np.float32) - mg.MEAN
    newp, loss, gn, acc, logits, grads = O.lrcn_train_step(p, x, onehot, fpc, lr=0.01, clip_norm=0.5, final_layer=layer,
                                                          lstm_layers=layers, fusion=fusion)
    np.testing.assert_allclose(logits, G[name + "/logits"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose([loss, gn, acc], G[name + "/loss_gn_acc"], rtol=1e-12)
    for k in p:
        np.testing.assert_allclose(grads[k].ravel()[:16], G[name + "/gradhead/" + k], rtol=1e-10, atol=1e-14)


def test_op_known_answers():
    np.testing.assert_allclose(O.lrn(G["lrn/x"])[0], G["lrn/y"], rtol=1e-14)
    y, arg = O.max_pool_valid(G["lrn/x"])
    np.testing.assert_array_equal(y, G["pool/y"])
    np.testing.assert_array_equal(arg, G["pool/arg"])
    assert [tuple(r) for r in G["same_pad"]] == [(57, 4, 4), (56, 3, 4), (28, 2, 2), (13, 1, 1)]
    np.testing.assert_allclose(O.precompute_learning_rates(0.05, ["exp", "drops", 4, 0.5], 5, 2), G["lr_table"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mg.CASES))
def test_engine_reproduces_golden(name):
    import torch
    from vltf_amd.engine import LRCNEngine, NetConfig
    shape, ncls, fpc, b, layer, hid, layers, fusion, seed = mg.CASES[name]
    p, frames, onehot = mg.case_inputs(name)
    eng = LRCNEngine(NetConfig(image_shape=shape, num_classes=ncls, fpc=fpc, frame_encoding_layer=layer, lstm_hidden=hid,
                               lstm_layers=layers, fusion=fusion), max_clips=b)
    eng.load_params(p)
    out = eng.train_step_u8(torch.tensor(frames, device="cuda:0"), torch.tensor(onehot, device="cuda:0"), lr=0.01,
                            clip_norm=0.5, mean_bgr=mg.MEAN)
    np.testing.assert_allclose(eng.logits_host(), G[name + "/logits"], rtol=1e-3, atol=1e-3)     # north_star tolerance
    loss, gn, acc = G[name + "/loss_gn_acc"]
    assert abs(out["loss"] - loss) < 1e-4 * max(1, abs(loss)) and abs(out["grad_norm"] - gn) < 1e-3 * gn
    g, w = eng.get_grads(), eng.get_params()
    for k in p:
        want = float(G[name + "/gradnorm/" + k][0])
        assert abs(np.linalg.norm(g[k].ravel().astype(np.float64)) - want) < 2e-3 * want + 1e-9, k
        np.testing.assert_allclose(w[k].ravel()[:16], G[name + "/newhead/" + k], rtol=1e-4, atol=1e-5, err_msg=k)
