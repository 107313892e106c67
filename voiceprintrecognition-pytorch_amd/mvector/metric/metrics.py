"""Verification metrics with the reference's signatures (mvector/metric/metrics.py:5-49).

Host-side numpy: one sort of the trial scores and two cumulative sums -- not part of the accelerated path, but part of
``MVectorTrainer.evaluate`` (trainer.py:463-468), whose GPU-side work (features, embeddings, score matrix) ends here.
"""
import numpy as np
import torch


def compute_fnr_fpr(scores, labels, weights=None):
    """Miss / false-alarm rates at every threshold = sorted score (ascending).

    fnr[i]: weighted fraction of target trials with score <= thresholds[i];
    fpr[i]: weighted fraction of impostor trials with score > thresholds[i].
    """
    scores = np.asarray(scores)
    labels = np.asarray(labels)
    order = np.argsort(scores)
    thresholds = scores[order]
    labels = labels[order]
    w = np.ones(labels.shape, dtype='f8') if weights is None else np.asarray(weights)[order]
    tgt = w * (labels == 1)
    imp = w * (labels == 0)
    fnr = np.cumsum(tgt) / np.sum(tgt)
    fpr = 1 - np.cumsum(imp) / np.sum(imp)
    return fnr, fpr, thresholds


def compute_eer(fnr, fpr, scores=None):
    """Equal error rate by linear interpolation between the two operating points around fnr == fpr; with ``scores``
    also the score at the first point where fnr >= fpr (the decision threshold the reference reports)."""
    gap = fnr - fpr
    i1 = np.flatnonzero(gap >= 0)[0]
    i2 = np.flatnonzero(gap < 0)[-1]
    a = (fnr[i1] - fpr[i1]) / (fpr[i2] - fpr[i1] - (fnr[i2] - fnr[i1]))
    eer = fnr[i1] + a * (fnr[i2] - fnr[i1])
    if scores is not None:
        return eer, np.sort(scores)[i1]
    return eer


def compute_dcf(fnr, fpr, p_target=0.01, c_miss=1, c_fa=1):
    """Minimum normalised detection cost."""
    c_det = np.min(c_miss * fnr * p_target + c_fa * fpr * (1 - p_target))
    c_def = min(c_miss * p_target, c_fa * (1 - p_target))
    return c_det / c_def


def accuracy(output, label):
    """Top-1 accuracy of classifier logits (training-time metric; kept for API completeness)."""
    pred = torch.argmax(torch.nn.functional.softmax(output, dim=-1), dim=1).cpu().numpy()
    return np.mean((pred == label.data.cpu().numpy()).astype(int))
