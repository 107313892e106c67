"""Batch assembly (reference: mvector/data_utils/collate_fn.py:4-24)."""
import torch


def collate_fn(batch):
    """[(feature [T_i, F], label)] -> (features [B, T_max, F] zero padded, labels int64 [B], input_lens int64 [B]);
    the batch order is kept."""
    freq = batch[0][0].size(1)
    t_max = max(item[0].size(0) for item in batch)
    features = torch.zeros((len(batch), t_max, freq), dtype=torch.float32)
    lens, labels = [], []
    for i, (feat, label) in enumerate(batch):
        n = feat.size(0)
        features[i, :n] = feat
        lens.append(n)
        labels.append(int(label))
    return features, torch.tensor(labels, dtype=torch.int64), torch.tensor(lens, dtype=torch.int64)


def collate_waveforms(batch):
    """[(samples [L_i], label)] -> (waveforms [B, L_max] zero padded, labels, num_samples).  Feeds the variable-length
    GPU front-end (``AudioFeaturizer.forward_varlen``), which featurises every row on its own length -- the same
    features ``collate_fn`` would have padded, without a per-utterance CPU pass."""
    l_max = max(item[0].numel() for item in batch)
    wav = torch.zeros((len(batch), l_max), dtype=torch.float32)
    lens, labels = [], []
    for i, (samples, label) in enumerate(batch):
        n = samples.numel()
        wav[i, :n] = samples
        lens.append(n)
        labels.append(int(label))
    return wav, torch.tensor(labels, dtype=torch.int64), torch.tensor(lens, dtype=torch.int64)
