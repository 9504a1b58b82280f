"""Helpers with the names of DiffVC/model/utils.py:16-40 (decoder-side subset)."""
import torch

from ...model.utils import convert_pad_shape, fix_len_compatibility, sequence_mask  # noqa: F401


def mse_loss(x, y, mask, n_feats):
    """utils.py:16-18."""
    return torch.sum(((x - y) ** 2) * mask) / (torch.sum(mask) * n_feats)
