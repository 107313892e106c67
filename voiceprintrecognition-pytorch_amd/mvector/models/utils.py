"""Building blocks shared by the TDNN-family backbones; parameter names follow the reference
(mvector/models/utils.py) so checkpoints load: ``Conv1d.conv``, ``BatchNorm1d.norm``, ``TDNNBlock.{conv,norm}``.
The torch forwards here serve CPU tensors and training; eval-mode CUDA forwards of whole backbones run in HIP."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def length_to_mask(length, max_len=None, dtype=None, device=None):
    assert len(length.shape) == 1
    if max_len is None:
        max_len = length.max().long().item()
    steps = torch.arange(max_len, device=length.device, dtype=length.dtype)
    mask = steps.expand(len(length), max_len) < length.unsqueeze(1)
    return torch.as_tensor(mask, dtype=length.dtype if dtype is None else dtype,
                           device=length.device if device is None else device)


def get_padding_elem(L_in, stride, kernel_size, dilation):
    """[left, right] padding that keeps the length ('same'), as the reference computes it (models/utils.py:28-36): half the kernel on a strided
    conv, half of what the dilated kernel takes away otherwise.  ``Conv1d.forward`` below applies the same rule."""
    if stride > 1:
        p = kernel_size // 2
    else:
        p = (L_in - ((L_in - dilation * (kernel_size - 1) - 1) // stride + 1)) // 2
    return [p, p]


class Conv1d(nn.Module):
    """nn.Conv1d with 'same' (reflect by default) / 'causal' / 'valid' padding done outside the conv."""

    def __init__(self, out_channels, kernel_size, in_channels, stride=1, dilation=1, padding='same', groups=1,
                 bias=True, padding_mode='reflect'):
        super().__init__()
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.padding, self.padding_mode = padding, padding_mode
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, padding=0,
                              groups=groups, bias=bias)

    def forward(self, x):
        if self.padding == 'same':
            x = F.pad(x, tuple(get_padding_elem(x.shape[-1], self.stride, self.kernel_size, self.dilation)), mode=self.padding_mode)
        elif self.padding == 'causal':
            x = F.pad(x, ((self.kernel_size - 1) * self.dilation, 0))
        elif self.padding != 'valid':
            raise ValueError(f"Padding must be 'same', 'valid' or 'causal'. Got {self.padding}")
        return self.conv(x)


class BatchNorm1d(nn.Module):
    def __init__(self, input_size, eps=1e-05, momentum=0.1):
        super().__init__()
        self.norm = nn.BatchNorm1d(input_size, eps=eps, momentum=momentum)

    def forward(self, x):
        return self.norm(x)


class TDNNBlock(nn.Module):
    """conv -> activation -> BatchNorm (in that order)."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation, activation=nn.ReLU, groups=1):
        super().__init__()
        self.conv = Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                           dilation=dilation, groups=groups)
        self.activation = activation()
        self.norm = BatchNorm1d(input_size=out_channels)

    def forward(self, x):
        return self.norm(self.activation(self.conv(x)))
