"""Validation metrics / checkpoint files (gfe_hip/validate.py; reference classify_mamba.py:119-173).  The metric formulas are
checked against scikit-learn (an independent implementation; torchmetrics, which the reference uses, is not in the image)."""
import numpy as np
import pytest
import torch


def _counts(prob, y, nbatch):
    from gfe_hip.validate import ValidationCounts
    c = ValidationCounts(prob.device)
    for p, t in zip(prob.chunk(nbatch), y.chunk(nbatch)):
        c.update(p, t)
    return c.compute()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_metrics_match_sklearn_with_the_reference_argument_order(seed):
    from sklearn.metrics import accuracy_score, f1_score, matthews_corrcoef, recall_score
    g = np.random.default_rng(seed)
    prob = torch.from_numpy(g.random(96).astype(np.float32))
    y = torch.from_numpy((g.random(96) < 0.4).astype(np.int64))
    m = _counts(prob.unsqueeze(1), y, 6)
    lab = prob.round().numpy().astype(int)
    # classify_mamba.py:143-146: metric.update(preds=labels, target=rounded predictions)
    assert abs(m["recall"] - recall_score(y_true=lab, y_pred=y.numpy())) < 1e-12
    assert abs(m["precision_ref"] - recall_score(y_true=y.numpy(), y_pred=lab)) < 1e-12
    assert abs(m["f1"] - f1_score(lab, y.numpy())) < 1e-12
    assert abs(m["accuracy"] - 100 * accuracy_score(lab, y.numpy())) < 1e-9
    assert abs(m["mcc"] - matthews_corrcoef(y.numpy(), lab)) < 1e-9
    # :151 sum of batch-mean losses / samples
    want = sum(float(torch.nn.functional.binary_cross_entropy(p, t.float())) for p, t in zip(prob.chunk(6), y.chunk(6))) / 96
    assert abs(m["validation_loss"] - want) < 1e-6 and m["total"] == 96 and m["correct"] == int((lab == y.numpy()).sum())


def test_degenerate_batches():
    m = _counts(torch.zeros(4), torch.zeros(4, dtype=torch.int64), 1)              # nothing positive anywhere: zero_division = 0
    assert m["recall"] == 0.0 and m["f1"] == 0.0 and m["accuracy"] == 100.0 and m["mcc"] == 0.0


def test_best_tracker_order():
    from gfe_hip.validate import BestTracker
    b = BestTracker()
    assert b.is_best(dict(accuracy=50.0, validation_loss=0.7))
    assert not b.is_best(dict(accuracy=50.0, validation_loss=0.7))
    assert b.is_best(dict(accuracy=50.0, validation_loss=0.6))                     # equal accuracy, lower loss (:155)
    assert not b.is_best(dict(accuracy=49.0, validation_loss=0.1))
    assert b.is_best(dict(accuracy=51.0, validation_loss=0.9))


def _small_step(device, seed=3):
    from gfe_hip.step import ClassifyStep, build_models
    gen, head, ft = build_models(vol=(32, 32, 32), f_maps=(8, 16, 32), dim=64, depth=2, heads=8,
                                 vit_kwargs=dict(dim=64, depth=2, heads=2, dim_head=16, mlp_dim=128), seed=seed, device=device)
    return ClassifyStep(gen, head, ft)


def test_checkpoint_files_round_trip(tmp_path):
    """File names and key layout of classify_mamba.py:157-158, 168-169; loading restores the flat buffers in place."""
    import os
    from gfe_hip.validate import load_checkpoint, save_checkpoint
    st = _small_step("cpu")
    paths = save_checkpoint(st, str(tmp_path), "best") + save_checkpoint(st, str(tmp_path), "current")
    assert [os.path.relpath(p, tmp_path) for p in paths] == ["model_best/best_model.pth", "model_best/best_ft_model.pth",
                                                             "model_current/model_current.pth", "model_current/ft_model_current.pth"]
    sd = torch.load(paths[3], map_location="cpu")
    assert list(sd.keys()) == list(st.ft.state_dict().keys())
    assert all(v.untyped_storage().nbytes() == v.numel() * v.element_size() for v in sd.values())     # clones, not views of the flat buffer
    before = st.opt.flat_p.clone()
    ptrs = [p.data_ptr() for p in st.all_params]
    with torch.no_grad():
        for p in st.all_params:
            p.add_(1.0)
    load_checkpoint(st, str(tmp_path), "current")
    assert torch.equal(st.opt.flat_p, before) and ptrs == [p.data_ptr() for p in st.all_params]
    assert torch.equal(st.opt.flat_p16.float(), before.to(torch.bfloat16).float())


@pytest.mark.gpu
def test_validate_epoch_on_device():
    import gfe_hip.det_init as det
    from gfe_hip.validate import validate
    st = _small_step("cuda")
    batches = [[t.cuda() for t in det.det_inputs(2, (32, 32, 32), seed=40 + i)] for i in range(3)]
    m = validate(st, batches)
    probs = torch.cat([st.eval_step(*b[:3]).reshape(-1) for b in batches]).cpu()
    ys = torch.cat([b[3] for b in batches]).cpu()
    lab = probs.round()
    assert m["total"] == 6 and m["correct"] == int((lab == ys).sum())
    tp, fp = float((lab * ys).sum()), float((lab * (1 - ys)).sum())
    assert abs(m["recall"] - (tp / (tp + fp) if tp + fp else 0.0)) < 1e-12
    want = sum(float(torch.nn.functional.binary_cross_entropy(st.eval_step(*b[:3]).reshape(-1), b[3].float())) for b in batches) / 6
    assert abs(m["validation_loss"] - want) < 1e-3        # two forward passes: split-K f32 atomics make them equal to ~1e-4 only
