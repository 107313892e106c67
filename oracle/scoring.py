"""Oracle cosine scoring (test infrastructure only).

Restates mvector/predict.py:165-183 (``normalize_features`` + sklearn ``cosine_similarity`` +
argmax/threshold), predict.py:275-279 (``contrast``) and the per-trial loop of
mvector/trainer.py:454-461, in numpy fp32.
"""
import numpy as np


def cosine_similarity(x, y):
    """sklearn.metrics.pairwise.cosine_similarity: row-L2-normalise both, X @ Y.T."""
    x = np.asarray(x, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    xn = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), np.finfo(np.float32).tiny)
    yn = y / np.maximum(np.linalg.norm(y, axis=1, keepdims=True), np.finfo(np.float32).tiny)
    return xn @ yn.T


def contrast(f1, f2):
    return float(np.dot(f1, f2) / (np.linalg.norm(f1) * np.linalg.norm(f2)))


def retrieval(features, gallery_means, names, threshold):
    """predict.py:169-183: best gallery row per query, thresholded, rounded to 5 places."""
    sims = cosine_similarity(features, gallery_means)
    labels = []
    for sim in sims:
        idx = int(np.argmax(sim))
        s = sim[idx]
        labels.append([names[idx], round(float(s), 5)] if s >= threshold else [None, None])
    return labels
