"""Freshness of everything derived from a parameter: weight packs, GDN fragment images, entropy-bottleneck tables, captured graphs.

The key is the parameter's autograd version counter + storage pointer.  Two kinds of write leave the counter alone:
  * torch's fused (single-kernel) optimizers (checked on torch 2.10) -- covered for EVERY caller by the global optimizer-step hook
    below, registered when this module is imported (any `import compressai` / `import MASIC` does): after each `optimizer.step()` of a
    fused or capturable group the versions of its parameters are bumped, so a driver that builds `Adam(..., fused=True)` itself needs
    no cooperation (tests/test_gpu_driver_loop.py::test_fused_adam_stepped_by_the_caller_never_serves_stale_packs);
  * writes through `.data` (`p.data.copy_()`, `p.data *= m`): invisible to any version counter by torch's design.  The modules of this
    tree do not do that to their own parameters except MaskedConv2d, which re-packs itself.  A caller who edits weights through `.data`
    calls `invalidate_packs(model)`; `MASIC_PACK_VERIFY=1` / `set_pack_verify(True)` (a debugging mode: one device reduction + host
    read per key lookup) adds a content fingerprint to every key, so that a stale pack cannot be served at all.
"""
import os

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook

_PACK_VERIFY = os.environ.get("MASIC_PACK_VERIFY", "0") != "0"


def set_pack_verify(on):
    global _PACK_VERIFY
    _PACK_VERIFY = bool(on)


def _fingerprint(t):
    d = t.detach().reshape(-1)
    if d.numel() == 0:
        return 0
    v = d.view(torch.int32) if d.dtype == torch.float32 else d.to(torch.float32).view(torch.int32)
    idx = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64)
    return int(((v.to(torch.int64) * (idx % 8191 + 1)).sum()).item())


def stamp(t):
    """What a cache compares to decide that tensor `t` still holds the values a derived buffer was built from."""
    if _PACK_VERIFY:
        return (t._version, _fingerprint(t))
    return t._version


def weight_key(w):
    """Cache key of anything derived from parameter `w`."""
    return (stamp(w), w.data_ptr(), str(w.device))


def invalidate_packs(module):
    """After writing parameters of `module` through `.data` (or any other way that bypasses the version counters)."""
    ts = [t for _, t in module.named_parameters()] + [t for n, t in module.named_buffers() if n.endswith(".mask")]
    if ts:
        torch._C._increment_version(ts)          # (the list form: per tensor it iterates over the tensor's rows)


def _bump_versions_after_step(optimizer, args, kwargs):
    ps = [p for g in optimizer.param_groups if g.get("fused") or g.get("capturable") for p in g["params"]]
    if ps:
        torch._C._increment_version(ps)


_STEP_HOOK = register_optimizer_step_post_hook(_bump_versions_after_step)
