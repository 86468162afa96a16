"""Minimal stand-ins for `timm.models.layers.DropPath` / `trunc_normal_` (reference import at
swin_hp_transformer.py:14).  Golden vectors are generated with drop_path_rate = 0, where DropPath
is the identity, and with explicitly loaded state dicts, so the init RNG stream is irrelevant."""
import torch
from torch import nn


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)
