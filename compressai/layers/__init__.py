from .gdn import GDN, GDN1
from .layers import MaskedConv2d, ResidualBlock, conv1x1, conv3x3

__all__ = ["GDN", "GDN1", "MaskedConv2d", "ResidualBlock", "conv3x3", "conv1x1"]
