"""`compressai.models.CompressionModel` import surface (reference compressai/models/google.py's base
class; MASIC.py:34 imports it and then shadows it with its own two-bottleneck variant, MASIC.py:40)."""
import torch.nn as nn

from compressai.entropy_models import EntropyBottleneck


class CompressionModel(nn.Module):
    def __init__(self, entropy_bottleneck_channels, init_weights=True):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)
        if init_weights:
            for m in self.modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                    nn.init.kaiming_normal_(m.weight)
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def forward(self, *args):
        raise NotImplementedError()

    def update(self, force=False):
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                m.update(force=force)
