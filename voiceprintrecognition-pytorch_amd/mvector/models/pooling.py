"""Pooling layers with the reference's names and parameters (mvector/models/pooling.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from mvector.models.utils import TDNNBlock, Conv1d, length_to_mask


class TemporalAveragePooling(nn.Module):
    """TAP: mean over time."""

    def forward(self, x):
        return torch.mean(x, dim=2).flatten(start_dim=1)


class TemporalStatisticsPooling(nn.Module):
    """TSP: mean and (unbiased) variance over time."""

    def forward(self, x):
        return torch.cat((torch.mean(x, dim=2), torch.var(x, dim=2)), dim=1)


class SelfAttentivePooling(nn.Module):
    """SAP: softmax(V tanh(W x)) weighted mean."""

    def __init__(self, in_dim, bottleneck_dim=128):
        super().__init__()
        self.linear1 = nn.Conv1d(in_dim, bottleneck_dim, kernel_size=1)
        self.linear2 = nn.Conv1d(bottleneck_dim, in_dim, kernel_size=1)

    def forward(self, x):
        alpha = torch.softmax(self.linear2(torch.tanh(self.linear1(x))), dim=2)
        return torch.sum(alpha * x, dim=2)


class AttentiveStatisticsPooling(nn.Module):
    """ASP: per-channel attention over time, returns cat(weighted mean, weighted std)."""

    def __init__(self, channels, attention_channels=128, global_context=True):
        super().__init__()
        self.eps = 1e-12
        self.global_context = global_context
        self.tdnn = TDNNBlock(channels * 3 if global_context else channels, attention_channels, 1, 1)
        self.tanh = nn.Tanh()
        self.conv = Conv1d(in_channels=attention_channels, out_channels=channels, kernel_size=1)

    def _stats(self, x, w):
        mean = (w * x).sum(2)
        std = torch.sqrt((w * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(self.eps))
        return mean, std

    def forward(self, x, lengths=None):
        L = x.shape[-1]
        if lengths is None:
            lengths = torch.ones(x.shape[0], device=x.device)
        mask = length_to_mask(lengths * L, max_len=L, device=x.device).unsqueeze(1)
        attn = x
        if self.global_context:
            total = mask.sum(dim=2, keepdim=True).float()
            mean, std = self._stats(x, mask / total)
            attn = torch.cat([x, mean.unsqueeze(2).expand(-1, -1, L), std.unsqueeze(2).expand(-1, -1, L)], dim=1)
        attn = self.conv(self.tanh(self.tdnn(attn)))
        attn = F.softmax(attn.masked_fill(mask == 0, float('-inf')), dim=2)
        return torch.cat(self._stats(x, attn), dim=1)


class TemporalStatsPool(nn.Module):
    """TSTP: mean and sqrt(var + 1e-8) over the last axis, flattened."""

    def forward(self, x):
        mean = x.mean(dim=-1).flatten(start_dim=1)
        std = torch.sqrt(torch.var(x, dim=-1) + 1e-8).flatten(start_dim=1)
        return torch.cat((mean, std), 1)
