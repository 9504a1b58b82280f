"""BaseModule -- same public surface as Grad-TTS/model/base.py:12-37 (`nparams`, `relocate_input`)."""
import torch


class BaseModule(torch.nn.Module):
    @property
    def nparams(self):
        """Number of trainable parameters (base.py:17-25)."""
        return int(sum(p.numel() for p in self.parameters() if p.requires_grad))

    def relocate_input(self, x: list):
        """Move the given tensors to the module's device (base.py:28-37)."""
        device = next(self.parameters()).device
        return [v.to(device) if isinstance(v, torch.Tensor) and v.device != device else v for v in x]
