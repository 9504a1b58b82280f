"""Host-side helpers with the names and semantics of Grad-TTS/model/utils.py:6-44."""
import torch


def sequence_mask(length, max_length=None):
    """utils.py:6-10 -- [b] lengths -> [b, max_length] bool."""
    if max_length is None:
        max_length = length.max()
    steps = torch.arange(int(max_length), dtype=length.dtype, device=length.device)
    return steps[None, :] < length[:, None]


def fix_len_compatibility(length, num_downsamplings_in_unet=2):
    """utils.py:13-17 -- smallest multiple of 2**n that is >= length."""
    q = 1 << num_downsamplings_in_unet
    return ((int(length) + q - 1) // q) * q if length % q else length


def convert_pad_shape(pad_shape):
    """utils.py:20-23 -- [[a,b],[c,d],..] (outer dim first) -> flat F.pad list (last dim first)."""
    return [v for pair in reversed(pad_shape) for v in pair]


def generate_path(duration, mask):
    """utils.py:26-39 -- durations [b,t_x] + mask [b,t_x,t_y] -> 0/1 monotone alignment [b,t_x,t_y]."""
    b, t_x, t_y = mask.shape
    ends = torch.cumsum(duration, 1)
    upto = sequence_mask(ends.reshape(b * t_x), t_y).to(mask.dtype).view(b, t_x, t_y)
    prev = torch.nn.functional.pad(upto, (0, 0, 1, 0))[:, :-1]
    return (upto - prev) * mask


def duration_loss(logw, logw_, lengths):
    """utils.py:42-44."""
    return torch.sum((logw - logw_) ** 2) / torch.sum(lengths)
