"""Oracle restatement of the reference's evaluation path (test infrastructure only; never imported by the product).

Follows, step by step:
  * mvector/data_utils/reader.py:84-106   per utterance: (dB normalise,) crop from the start, featurise ALONE
  * mvector/data_utils/collate_fn.py:4-24 zero-pad the features of a batch to its longest item
  * mvector/trainer.py:427-451            model forward per padded batch, embeddings concatenated
  * mvector/trainer.py:452-461            per trial: cosine against every enrolment embedding, label = same speaker
  * mvector/metric/metrics.py:5-40        sort once, cumulative sums -> fnr / fpr; EER by interpolation; minDCF
Pinned: the metric functions are checked against tests/golden/metrics.npz, which oracle/make_golden.py produced by running
the reference's own ``mvector/metric/metrics.py``.
"""
import numpy as np
import torch

from oracle import frontend, models, scoring


def fnr_fpr(scores, labels):
    order = np.argsort(scores)
    thresholds = scores[order]
    lab = labels[order]
    tgt = (lab == 1).astype('f8')
    imp = (lab == 0).astype('f8')
    return np.cumsum(tgt) / np.sum(tgt), 1 - np.cumsum(imp) / np.sum(imp), thresholds


def eer(fnr, fpr, scores):
    d = fnr - fpr
    x1 = np.flatnonzero(d >= 0)[0]
    x2 = np.flatnonzero(d < 0)[-1]
    a = (fnr[x1] - fpr[x1]) / (fpr[x2] - fpr[x1] - (fnr[x2] - fnr[x1]))
    return fnr[x1] + a * (fnr[x2] - fnr[x1]), np.sort(scores)[x1]


def min_dcf(fnr, fpr, p_target=0.01, c_miss=1, c_fa=1):
    return min(c_miss * fnr * p_target + c_fa * fpr * (1 - p_target)) / min(c_miss * p_target, c_fa * (1 - p_target))


def embed_list(state, model_name, waveforms, batch_size, method, method_args):
    """waveforms: list of 1-D float32 tensors (already cropped / normalised), in list order."""
    embs = []
    for i in range(0, len(waveforms), batch_size):
        feats = [frontend.audio_featurizer(w.unsqueeze(0), None, method, method_args)[0] for w in waveforms[i:i + batch_size]]
        t_max = max(f.shape[0] for f in feats)
        batch = torch.zeros(len(feats), t_max, feats[0].shape[1])
        for j, f in enumerate(feats):
            batch[j, :f.shape[0]] = f
        with torch.no_grad():
            embs.append(models.FORWARDS[model_name](state, batch))
    return torch.cat(embs).numpy()


def evaluate(state, model_name, enroll, enroll_labels, trials, trials_labels, batch_size, method='Fbank', method_args=None):
    """-> (eer, min_dcf, threshold, scores [n_trials, n_enroll])"""
    method_args = method_args or {}
    e = embed_list(state, model_name, enroll, batch_size, method, method_args)
    t = embed_list(state, model_name, trials, batch_size, method, method_args)
    all_score, all_labels = [], []
    for i in range(len(t)):
        all_score.extend(scoring.cosine_similarity(t[i:i + 1], e).astype(np.float32).tolist()[0])
        all_labels.extend((np.asarray(enroll_labels) == trials_labels[i]).astype(np.int32).tolist())
    sc = np.array(all_score, dtype=np.float32)
    lb = np.array(all_labels, dtype=np.int32)
    fnr, fpr, _ = fnr_fpr(sc, lb)
    e_, thr = eer(fnr, fpr, sc)
    return float(e_), float(min_dcf(fnr, fpr)), float(thr), sc.reshape(len(t), len(e))
