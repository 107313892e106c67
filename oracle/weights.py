"""Seeded weights keyed by state_dict name (oracle/test infrastructure only).

No trained checkpoint of the reference is obtainable offline (README.md:74-85), so every
parity check runs on synthetic weights.  They are drawn per tensor from a numpy PCG64
stream seeded by crc32(name) ^ seed, i.e. independent of module construction order, so the
reference modules (make_golden.py), the oracle and the HIP-backed modules all see identical
values from the key -> shape manifest alone.  BatchNorm statistics are randomised
(SURVEY.md section 8(d)): default-initialised BN is the identity and would hide BN bugs.
"""
import zlib

import numpy as np
import torch


def make_state_dict(shapes, seed=0, bn_gain=1.0, stress=False):
    """shapes: {key: tuple}.  Returns {key: torch fp32 tensor (int64 for num_batches_tracked)}.
    bn_gain scales every BatchNorm weight: deep residual stacks with unit-gain random weights are chaotic (the reference's own
    fp32 forward then differs from fp64 by >10 %), a gain < 1 makes such a case well conditioned.
    stress: the fp16 stress recipe -- every conv / linear output channel is scaled by 10^U(-1.5, 0.5) (activation variances
    spread over 1e-3 .. 10, as the running_var of a trained network can be) and BatchNorm gains are U(0.5, 3).  The running
    statistics of such a model must be CALIBRATED ones (make_golden.py::save_stress_golden stores them next to the vectors:
    with independent random statistics the reference's own fp32 forward overflows), so the values drawn here for
    running_mean / running_var are placeholders the fixture overrides."""
    out = {}
    keys = set(shapes)
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        rng = np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))
        stem, _, leaf = name.rpartition('.')
        is_bn = (stem + '.running_mean') in keys
        if leaf == 'num_batches_tracked':
            out[name] = torch.tensor(100, dtype=torch.int64)
            continue
        if leaf == 'running_mean':
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == 'running_var':
            v = rng.uniform(0.5, 1.5, shape)
        elif is_bn and leaf == 'weight':
            v = (rng.uniform(0.5, 3.0, shape) if stress else rng.uniform(0.8, 1.2, shape)) * bn_gain
        elif leaf == 'bias':
            v = rng.normal(0.0, 0.1, shape)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
            if stress:
                v = v * (10.0 ** rng.uniform(-1.5, 0.5, (shape[0],) + (1,) * (len(shape) - 1)))
        else:
            v = rng.normal(0.0, 0.1, shape)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))
    return out


def shapes_of(state_dict):
    return {k: tuple(v.shape) for k, v in state_dict.items()}
