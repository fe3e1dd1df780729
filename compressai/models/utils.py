"""Layer factories of the codec (reference compressai/models/utils.py:128-146) returning the
HIP-backed conv modules, plus `update_registered_buffers` (utils.py:90-125)."""
import torch

from masic_amd.nn import Conv2d, ConvTranspose2d


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    """Conv2d with padding = kernel_size // 2."""
    return Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    """ConvTranspose2d with padding = kernel_size // 2 and output_padding = stride - 1."""
    return ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                           output_padding=stride - 1, padding=kernel_size // 2)


def update_registered_buffers(module, module_name, buffer_names, state_dict, policy="resize_if_empty", dtype=torch.int):
    """Resize (or register) the dynamically sized CDF buffers of an entropy model so that
    `load_state_dict` accepts a checkpoint saved after `update()`."""
    if policy not in ("resize_if_empty", "resize", "register"):
        raise ValueError(f'Invalid policy "{policy}"')
    present = dict(module.named_buffers())
    for name in buffer_names:
        if name not in present:
            raise ValueError(f'Invalid buffer name "{name}"')
    for name in buffer_names:
        key = f"{module_name}.{name}"
        if policy == "register":
            if name in present and present[name] is not None and hasattr(module, name):
                raise RuntimeError(f'buffer "{name}" was already registered')
            module.register_buffer(name, torch.empty(state_dict[key].size(), dtype=dtype).fill_(0))
            continue
        if key not in state_dict:
            raise RuntimeError(f'buffer "{key}" was not found in the state_dict')
        buf = present[name]
        if policy == "resize" or buf.numel() == 0:
            buf.resize_(state_dict[key].size())
