"""Shared helpers for the parity tests (inputs and weights are re-derived from seeds/names)."""
import functools
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@functools.lru_cache(maxsize=None)
def manifest():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


@functools.lru_cache(maxsize=None)
def weights(which):
    """name-keyed synthetic state_dict for 'network_g' / 'network_motion_estimator'."""
    from synergize_motion_appearance_amd.synth import synth_state_dict
    return synth_state_dict([(k, tuple(s)) for k, s in manifest()[which]])


@functools.lru_cache(maxsize=None)
def clip(n=8, seed=123):
    from synergize_motion_appearance_amd.synth import synth_clip
    return synth_clip(n, seed=seed)


def maxabs(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max())
