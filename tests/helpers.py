"""Shared helpers for the test-suite (loads golden fixtures and seeded weights)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_case(case):
    """-> (manifest dict, state_dict, x [B,T,F] tensor, reference embedding tensor, npz)."""
    from oracle import weights
    with open(os.path.join(GOLDEN, f'manifest_{case}.json')) as f:
        man = json.load(f)
    sd = weights.make_state_dict(man['shapes'], man['seed'], man.get('bn_gain', 1.0), stress=man.get('stress', False))
    z = np.load(os.path.join(GOLDEN, f'{case}.npz'))
    for k in z.files:  # calibrated BatchNorm statistics of the stress cases travel in the fixture
        if k.startswith('bn:'):
            sd[k[3:].replace('/', '.')] = torch.from_numpy(z[k])
    if man.get('hot_head_log2'):   # oracle/make_golden.py::hot_head_state_dict: FCM head maps 2^k times larger, the same network otherwise
        g = float(2 ** man['hot_head_log2'])
        for name in list(sd):
            stem, _, leaf = name.rpartition('.')
            if name.startswith('head.') and stem != 'head.bn2' and (stem + '.running_mean') in sd and leaf in ('weight', 'bias'):
                sd[name] = sd[name] * g
    return man, sd, torch.from_numpy(z['x']), torch.from_numpy(z['emb']), z


def cos_dist(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return (1.0 - torch.nn.functional.cosine_similarity(a, b, dim=-1))
