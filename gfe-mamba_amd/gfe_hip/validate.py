"""Validation pass and checkpoint files of classify_mamba.py (reference: classify_mamba.py:119-173), device-side.

The reference syncs the host three times per validation batch (`.item()` on the loss and on the correct count) and hands the
concatenated labels to torchmetrics at the end.  Here a validation batch adds to one small device buffer (confusion counts + loss sum)
and the host reads it ONCE per epoch.  torchmetrics is an un-vendored dependency of the reference (absent from this image), so
the three metrics are restated from their definitions for the `task='binary'` case -- parity unpinned for this file, the formulas are
three lines each.

Reference quirks kept on purpose (a maintainer switching over must see the same numbers in the same log line):
  * classify_mamba.py:143-150 passes the LABELS as torchmetrics' `preds` and the rounded predictions as `target`, so the logged
    "Recall" is TP / (TP + FP) in the usual convention (i.e. precision).  Accuracy and F1 are symmetric.  `recall`/`f1`/`accuracy`
    below follow the reference; `precision_ref` is the other one (= the conventional recall) for whoever wants both.
  * validation_loss (:151) is the sum of per-batch MEAN losses divided by the number of SAMPLES.
  * accuracy is in percent (:148).
"""
import os

import torch
import torch.nn.functional as F


class ValidationCounts:
    """tp, fp, tn, fn (prediction vs label, usual convention), sum of batch-mean losses, samples -- f64 on the device."""

    def __init__(self, device):
        self.buf = torch.zeros(6, dtype=torch.float64, device=device)

    def update(self, prob, y):
        """prob: sigmoid outputs (B,) or (B,1); y: labels (B,).  No host sync."""
        prob = prob.reshape(-1).float()
        yf = y.reshape(-1).float()
        lab = prob.round()                                                         # :139 (round-half-even like torch.round)
        loss = F.binary_cross_entropy(prob, yf)                                    # :138
        tp = (lab * yf).sum()
        self.buf += torch.stack([tp, lab.sum() - tp, ((1 - lab) * (1 - yf)).sum(), yf.sum() - tp, loss,
                                 torch.tensor(float(yf.numel()), device=prob.device)]).double()

    def compute(self):
        tp, fp, tn, fn, loss_sum, n = self.buf.tolist()                            # the epoch's only device->host read
        div = lambda a, b: a / b if b else 0.0                                     # torchmetrics' zero_division=0
        mcc_den = ((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn)) ** 0.5
        return dict(accuracy=100.0 * div(tp + tn, n),                              # :148
                    recall=div(tp, tp + fp),                                       # :149 with the swapped arguments of :146
                    f1=div(2 * tp, 2 * tp + fp + fn),                              # :150
                    validation_loss=div(loss_sum, n),                              # :151
                    precision_ref=div(tp, tp + fn), mcc=div(tp * tn - fp * fn, mcc_den), total=int(n), correct=int(tp + tn))


def validate(step, batches):
    """One validation epoch (classify_mamba.py:129-151).  `step`: ClassifyStep; `batches`: iterable of (x, x_cat, x_num, y)
    already on the device.  Returns the dict of ValidationCounts.compute()."""
    counts = None
    for x, x_cat, x_num, y in batches:
        prob = step.eval_step(x, x_cat, x_num)                                     # :134-137
        counts = counts or ValidationCounts(prob.device)
        counts.update(prob, y)
    return counts.compute() if counts is not None else ValidationCounts("cpu").compute()


class BestTracker:
    """classify_mamba.py:154: a new best is a higher accuracy, or an equal accuracy with a lower validation loss."""

    def __init__(self):
        self.best_accuracy, self.best_losses = 0.0, float("inf")      # :80-81

    def is_best(self, m):
        if m["accuracy"] > self.best_accuracy or (m["accuracy"] == self.best_accuracy and m["validation_loss"] < self.best_losses):
            self.best_accuracy, self.best_losses = m["accuracy"], m["validation_loss"]
            return True
        return False


_FILES = dict(best=("model_best/best_model.pth", "model_best/best_ft_model.pth"),                 # :157-158
              current=("model_current/model_current.pth", "model_current/ft_model_current.pth"))  # :168-169


def _host_state(module):
    # parameters are views into FlatAdam's flat buffer: clone, or torch.save would write the whole buffer once per file
    return {k: v.detach().to("cpu", copy=True).contiguous() for k, v in module.state_dict().items()}


def save_checkpoint(step, run_dir, kind="current"):
    """Writes the head's and the classifier's state dicts under the reference's file names and key layout, so that the reference's
    own `load_state_dict` (plain `torch.load` + `load_state_dict`) accepts them.  `step.opt.wait_updated()` first: an overlapped
    update may still be running."""
    step.join()
    step.opt.wait_updated()
    torch.cuda.current_stream().synchronize() if torch.cuda.is_available() else None
    paths = [os.path.join(run_dir, f) for f in _FILES[kind]]
    for path, module in zip(paths, (step.head, step.ft)):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp"
        torch.save(_host_state(module), tmp)
        os.replace(tmp, path)                                                      # never a half-written checkpoint
    return paths


def load_checkpoint(step, run_dir, kind="current"):
    """Inverse of save_checkpoint; parameters are copied INTO the flat buffers (the views stay valid) and the bf16 shadows follow.
    Ordered against a head step still running on the pipeline's head stream and against an overlapped update (as save_checkpoint is);
    the head stream in turn waits for the load before its next step."""
    step.join()
    step.opt.wait_updated()
    for f, module in zip(_FILES[kind], (step.head, step.ft)):
        sd = torch.load(os.path.join(run_dir, f), map_location="cpu")
        missing, unexpected = module.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
    step.opt.flat_p16.copy_(step.opt.flat_p)
    step.opt._register_shadows()                    # load_state_dict bumped the parameters' versions
    hs = getattr(step, "_head_stream", None)
    if hs is not None:
        hs.wait_stream(torch.cuda.current_stream())
